// Convolution engine, 3xf16 split form with BOTH operands DMA'd straight into LDS.
//
// Arithmetic is the error-compensated split of conv_f16x3.hip (x.w = xh.wh + xh.wl + xl.wh on
// v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32-class result).  The difference is where the
// split happens: activations live in HBM in the "split16" format (conv_common.h: per pixel, each
// group of 8 channels = [8 x f16 hi][8 x f16 lo], same bytes as fp32) written ONCE by the
// producing kernel's epilogue, so a K tile of a row is one 128-B run made of the exact 16-B MFMA
// operand chunks.  Both the A (activation) and B (weight) panels are then filled with
// global_load_lds_dwordx4 -- no VGPR staging, no conversions, no ds_write: per K tile a wave
// issues 2*(MR+NR) DMA loads, 4*(MR+NR) ds_read_b128 and 12*MR*NR MFMAs.
//   * the DMAs are buffer loads (buffer_load_dwordx4 ... lds): a 128-bit descriptor in SGPRs, ONE 32-bit VGPR offset per
//     16-row group and a scalar offset that walks the K tiles -- in steady state the address stream costs no vector
//     instruction and no 64-bit pointer registers (a tap change of a 3x3 layer re-derives the lane offsets, nothing else);
//   * the LDS image of a DMA is lane-linear (wave base + lane*16), so the XOR chunk swizzle that
//     keeps ds_read_b128 conflict-free is applied on the SOURCE side: lane (row=l>>2, slot=l&3)
//     fetches chunk slot ^ ((row>>2)&3) of its row;
//   * zero padding (image borders, M/N tails) = lanes whose offset lies outside the descriptor's range: the
//     hardware bounds check writes zeros to LDS without touching memory;
//   * NS-stage LDS ring; the one barrier per K tile is preceded by a hand-written `s_waitcnt vmcnt(n)`
//     that waits only for the OLDEST tile in flight (vector-memory results return in order), so NS-1
//     (or NS, see PB below) tiles of DMA stay outstanding across barriers -- the L2 -> LDS path
//     (~56 B/clk/CU) runs at throughput instead of one latency per K tile;
//   * the K loop is software-pipelined by hand (k_tile below): a K tile is two 16-wide slices; the
//     operand fragments of a slice are fetched from LDS while the MFMAs of the previous slice run, and
//     the DMA instructions of the tile being prefetched are pinned one per two MFMAs (sched_barrier)
//     instead of issued as a block.  All waves of a workgroup are phase-locked by the barrier, so any
//     block of non-MFMA work (DMA issue, LDS wait) would idle the matrix pipe on every SIMD at once;
//   * per-lane source cursors: the (tap, channel-tile) address of a lane's row is re-derived only when
//     the tap changes, otherwise advanced by one 64-bit add per K tile;
//   * epilogue: residual groups and bias are loaded before the accumulators are transposed through the
//     LDS, then bias / residual / ReLU / SPLIT16 re-split on 8 channels per lane, 16-byte stores.
#pragma once
#include "conv_common.h"
#include <type_traits>

#ifndef SRCNN_PB_MAX_NS
#define SRCNN_PB_MAX_NS 2          // ring depths up to this use the issue-behind-the-barrier DMA schedule (see the kernel)
#endif

namespace srcnn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int SROW = BK;   // halves per LDS row (64 B), chunks XOR-swizzled

// one buffer_load_dwordx4 ... lds: every lane moves 16 B from (descriptor base + its own 32-bit offset + a wave-uniform
// scalar offset) to (wave-uniform LDS base) + lane*16 (IMM, the instruction offset, is added to BOTH addresses: keep it 0); lanes whose offset is outside the descriptor's num_records
// write zeros.  Device-only builtin, hence the guard for the host pass.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#else
struct rsrc_t {};            // host pass: the kernel body is parsed, never run
#endif
constexpr int OOB = (int)0x80000000u;      // lane offset of a padded row: beyond any descriptor (num_records <= 2^31 - 1)

template <int IMM>
__device__ __forceinline__ void dma16b(rsrc_t rsrc, int voff, int soff, _Float16 *lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_wave_base, 16, voff, soff, IMM, 0);
#else
    (void)rsrc; (void)voff; (void)soff; (void)lds_wave_base;
#endif
}

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, size_t bytes)
{
    const unsigned n = bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
#else
    (void)n;
    return rsrc_t{};
#endif
}

// wait until at most N of this wave's vector-memory operations are outstanding and every LDS read has
// returned, then workgroup barrier.  Hand-written so that the compiler's fence (vmcnt(0)) is not used.
template <int N>
__device__ __forceinline__ void wait_vm_barrier()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// debug form of the above (only when the stamp hook is armed): how long the wave sat in the vmcnt wait and in the barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier_timed(unsigned long long &w_vm, unsigned long long &w_bar)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    w_vm += t1 - t0;
    w_bar += t2 - t1;
}

extern __shared__ __attribute__((aligned(1024))) _Float16 smem[];

// MR/NR: per-WAVE tile in 32x32 MFMA tiles; WM x 2 waves per workgroup -> workgroup tile (32*MR*WM) x (64*NR).
// WM = 2: 4 waves (256 threads).  WM = 4: 8 waves (512 threads).  NS: LDS ring stages (NS-1 K tiles in flight).
// HEAD: 0 = none; 1 (256x256 tile only) = the epilogue applies a 6-channel 1x1 head to the activated pixels with fp32 FMAs and
// stores that instead of y (srcnn_conv_desc.head_w); 2 = the MFMA form of a narrow head (<= 32 outputs, srcnn_conv_desc.head_wf):
// a second GEMM over the tile's columns on the matrix pipe, final or as per-(eye, N tile) partial sums.
// One output tile (mt, nt) of a convolution: everything the single-launch kernel does between the tile mapping and its return.
// The launch kernels below (one tile per workgroup: conv_f16s.hip; a CHAIN of convolutions over the same rows or a GROUP of
// independent convolutions per launch: conv_chain.hip) only decide which tiles a workgroup computes.
// CTX (a small policy object) says which tile this is and where the launch-dependent odds and ends live:
//   int mt, nt, m_rows, kt_begin, kt_end;      tile coordinates; rows whose INPUT is read (p.M, or less under a device-side row
//                                              limit); this workgroup's K tiles (a split-K slice, or all of them)
//   int thread() const;                        threadIdx.x
//   int split_idx() const;                     >= 0: split-K slice index (partial sums go to p.partial); -1: not split
//   bool stamping(const ConvArgs &) const;     debug stamps armed?   unsigned long long *stamp_slot(const ConvArgs &) const;
// Functions, not fields, so that the single-launch kernel keeps reading blockIdx / the kernel arguments at the point of use
// (a pointer kept live across the K loop costs two SGPRs the 64x128 tiles do not have: 129 VGPRs = one wave per SIMD fewer).
struct TileCtxPlain {            // chained / grouped launches: never split, never stamped
    int mt, nt, m_rows, kt_begin, kt_end;
    __device__ __forceinline__ int split_idx() const { return -1; }
    // the thread index, laundered through an empty asm: inside a loop over tiles the compiler otherwise hoists every
    // lane-derived constant of the tile code (some 50 VGPRs) out of the loop and keeps it live across the K loops
    __device__ __forceinline__ int thread() const
    {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    }
    template <typename A> __device__ __forceinline__ bool stamping(const A &) const { return false; }
    template <typename A> __device__ __forceinline__ unsigned long long *stamp_slot(const A &) const { return nullptr; }
};

}  // namespace srcnn
