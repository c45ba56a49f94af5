// Fused tail of a ResNet bottleneck (resnet.py:82-101): conv2 (3x3, C -> C) + BN + ReLU  ->  conv3 (1x1, C -> 4C) + BN
// + residual + ReLU in ONE launch; the C-channel intermediate never leaves the CU.
//
// Why: in the trunk the 1x1 expand convolution has K = C only (64..256): a workgroup of the stand-alone kernel spends
// half of its life in prologue / epilogue, the C-channel map is written to and read back from HBM in between, and
// layer3's 23 blocks are 69 launch-synchronous phases of <= 48 us each (profiles/timeline_r01_f16x3.txt).  Here one
// workgroup owns BM = 16384 / C output pixels across ALL channels:
//   phase 1  implicit-GEMM 3x3 conv, tile BM x C, K = 9C: both operands DMA'd into an LDS ring (global_load_lds), exactly
//            the arithmetic of conv_f16s.hip (3-term f16 split, fp32 accumulate, cross terms in their own accumulator);
//   hand-off the BM x C result (bias, ReLU, hi/lo re-split) is written into LDS in MFMA-operand layout (64 KB, over the
//            ring it no longer needs) -- it is the complete K extent of conv3 for these pixels;
//   phase 2  conv3 as 4 output-channel chunks of C: A operand resident in LDS, only the weights stream through the ring;
//            per chunk: + bias + residual (SPLIT16, 16-byte loads), ReLU, SPLIT16 re-split, 16-byte stores.
// The MFMAs are issued with the weight fragment as the first operand, so the accumulator of a lane holds ONE pixel and 16
// channels; a v_permlane32_swap between lanes l and l+32 turns that into two whole 8-channel groups per lane, which is the
// unit of the SPLIT16 format and of the MFMA operand rows: no LDS transpose in either epilogue.
// 8 wavefronts per workgroup (BM/32 x C/64), per-wave tile 32 pixels x 64 channels.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

namespace srcnn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float __attribute__((address_space(4))) cfloat_k;      // constant address space: uniform loads become s_load

struct BlockArgs {
    const void *x;                 // conv2 input, SPLIT16 (B, H, W, C)
    const void *w2_hi, *w2_lo;     // (C, 3, 3, C) f16 halves of w * 2^k2
    const float *bias2;
    const void *w3_hi, *w3_lo;     // (4C, C)
    const float *bias3;
    const void *res;               // block input, SPLIT16 (B, H, W, 4C)
    void *y;                       // block output, SPLIT16 (B, H, W, 4C)
    const void *zero_page;
    float os2, os3;                // 2^-k2, 2^-k3
    int H, W, M;                   // M = B * H * W
    int mtiles;
    int flags;                     // debug: bit 0 count the epilogue stores in the vmcnt budget, bit 1 skip the residual
                                   // loads, bit 2 skip the stores (timing experiments only; SRCNN_BLK_FLAGS)
    unsigned long long *stamp;     // debug: 8 x u64 per workgroup (100 MHz chip clock), normally nullptr
    unsigned *range_flag;          // SPLIT16 range guard (conv_common.h)
    int tag;
};

__device__ __forceinline__ void blk_dma16(const void *gsrc, _Float16 *lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#else
    (void)gsrc;
    (void)lds_wave_base;
#endif
}

// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform, one of a handful of
// values), every LDS access has returned, then workgroup barrier
__device__ __forceinline__ void blk_wait_barrier(int n)
{
#define SRCNN_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    switch (n) {
        SRCNN_W(0) SRCNN_W(1) SRCNN_W(2) SRCNN_W(3) SRCNN_W(4) SRCNN_W(5) SRCNN_W(6) SRCNN_W(7) SRCNN_W(8) SRCNN_W(9)
        SRCNN_W(10) SRCNN_W(11) SRCNN_W(12) SRCNN_W(13) SRCNN_W(14) SRCNN_W(15) SRCNN_W(16) SRCNN_W(17) SRCNN_W(18)
        SRCNN_W(19) SRCNN_W(20) SRCNN_W(21) SRCNN_W(22) SRCNN_W(23) SRCNN_W(24)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
#undef SRCNN_W
}

// lanes l and l + 32 exchange: afterwards (x, y) of lane l < 32 = (x_l, x_{l+32}) and of lane l >= 32 = (y_{l-32}, y_l)
__device__ __forceinline__ void swap32(float &x, float &y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
#endif
}

extern __shared__ __attribute__((aligned(1024))) _Float16 blk_smem[];

template <int CM>
struct BlockCfg {
    static constexpr int BM = 16384 / CM;                 // pixels per workgroup
    static constexpr int WNG = CM / 64, WMG = BM / 32;    // wave grid (N x M), WNG * WMG == 8
    static constexpr int KT2 = CM / 32;                   // K tiles of conv3 (= K tiles of the LDS-resident operand)
    static constexpr int APW = BM / 64, BPW = CM / 64;    // DMA pieces per wave per K tile (A, B)
    static constexpr int LPT1 = APW + BPW, LPT2 = BPW;
    static constexpr int STAGE1 = (BM + CM) * 64;         // halves: (hi + lo) x (BM + CM) rows x 32
    static constexpr int STAGE2 = CM * 64;                // halves: (hi + lo) x CM weight rows x 32
    static constexpr int T2 = BM * CM * 2;                // halves (64 KB)
    static constexpr int NS1 = 4;
    static constexpr int NS2 = (81920 - T2) / STAGE2 >= 4 ? 4 : 3;       // ring of phase 2 behind the resident operand
    static constexpr size_t LDS = 2 * (size_t)(NS1 * STAGE1 > T2 + NS2 * STAGE2 ? NS1 * STAGE1 : T2 + NS2 * STAGE2);
    static_assert(WNG * WMG == 8, "8 wavefronts");
    static_assert(LDS <= 163840, "LDS budget");
};

template <int CM, bool PB1, bool PB2>
__global__ __launch_bounds__(512, 2) void conv_block_kernel(const BlockArgs p)
{
    using C = BlockCfg<CM>;
    constexpr int BM = C::BM, WNG = C::WNG, KT2 = C::KT2, APW = C::APW, BPW = C::BPW;
    constexpr int AG = APW >= 2 ? APW / 2 : 1, BG = BPW >= 2 ? BPW / 2 : 1;      // 16-row DMA groups per wave
    constexpr int PANEL_A = BM * 32, PANEL_B = CM * 32;                          // halves
    constexpr int K2T = 9 * CM / 32;                                             // K tiles of conv2
    constexpr int NST = 8;                                                       // 16-byte stores per lane per chunk epilogue

    const int t = threadIdx.x;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (p.stamp) ts[0] = __builtin_amdgcn_s_memrealtime();
    const int count_stores = (p.flags & 1) ? NST : 0;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wm = wave / WNG, wn = wave % WNG;
    const int li = lane & 31, lg = lane >> 5;
    // XCD-aware tile map: workgroups of one XCD (blockIdx % 8) own a contiguous range of pixel tiles
    const int nblk = p.mtiles;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, slot_x = bid >> 3;
    const int mt = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot_x;
    const int m0 = mt * BM;

    // ------------------------------------------------------------------ DMA geometry
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);          // chunk swizzle applied on the source side
    const char *zero = reinterpret_cast<const char *>(p.zero_page) + (lane & 3) * 16;
    // A (activations): AG row groups of 16 pixels; hi and lo halves of a group share the cursor
    const int a_grp0 = APW >= 2 ? wave * AG : (wave >> 1);       // first row group of this wave
    const int a_half = APW >= 2 ? 0 : (wave & 1);                // APW == 1: this wave moves only one half
    int a_ih0[AG], a_iw0[AG];
    const char *a_base[AG], *a_cur[AG];
    int a_step[AG];
#pragma unroll
    for (int g = 0; g < AG; ++g) {
        const int m = m0 + (a_grp0 + g) * 16 + drow;
        if (m < p.M) {
            const int hw = p.H * p.W;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oh = rem / p.W;
            const int ow = rem - oh * p.W;
            a_ih0[g] = oh - 1;                                   // 3x3, stride 1, pad 1
            a_iw0[g] = ow - 1;
            a_base[g] = reinterpret_cast<const char *>(p.x) + ((size_t)((b * p.H + a_ih0[g]) * p.W + a_iw0[g]) * CM) * 4 + dchunk * 32;
        } else {
            a_ih0[g] = -(1 << 28);
            a_iw0[g] = 0;
            a_base[g] = zero;
        }
    }
    int ld_kh = 0, ld_kw = 0, ld_c0 = 0;
    auto retap = [&]() {
        const size_t tap_off = ((size_t)(ld_kh * p.W + ld_kw) * CM + ld_c0) * 4;
#pragma unroll
        for (int g = 0; g < AG; ++g) {
            const int ih = a_ih0[g] + ld_kh, iw = a_iw0[g] + ld_kw;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            a_cur[g] = ok ? a_base[g] + tap_off : zero;
            a_step[g] = ok ? 128 : 0;
        }
    };
    retap();
    // B (weights): BG row groups of 16 output channels
    const int b_grp0 = BPW >= 2 ? wave * BG : (wave >> 1);
    const int b_half = BPW >= 2 ? 0 : (wave & 1);
    const char *bh_cur[BG], *bl_cur[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) {
        const int n = (b_grp0 + g) * 16 + drow;
        const size_t off = ((size_t)n * (9 * CM) + dchunk * 8) * 2;
        bh_cur[g] = reinterpret_cast<const char *>(p.w2_hi) + off;
        bl_cur[g] = reinterpret_cast<const char *>(p.w2_lo) + off;
    }
    // one DMA instruction of the current conv2 K tile into ring stage `sb`; pc is a compile-time index after unrolling
    auto dma1 = [&](int pc, _Float16 *sb) {
        if (pc < APW) {
            const int g = APW >= 2 ? (pc >> 1) : 0;
            const int half = APW >= 2 ? (pc & 1) : a_half;
            blk_dma16(a_cur[g] + half * 16, sb + half * PANEL_A + (a_grp0 + g) * 16 * 32);
        } else {
            const int pb = pc - APW;
            const int g = BPW >= 2 ? (pb >> 1) : 0;
            const int half = BPW >= 2 ? (pb & 1) : b_half;
            blk_dma16(half ? bl_cur[g] : bh_cur[g], sb + 2 * PANEL_A + half * PANEL_B + (b_grp0 + g) * 16 * 32);
        }
    };
    auto advance1 = [&]() {
        ld_c0 += 32;
        if (ld_c0 == CM) {
            ld_c0 = 0;
            if (++ld_kw == 3) { ld_kw = 0; ++ld_kh; }
            retap();
        } else {
#pragma unroll
            for (int g = 0; g < AG; ++g) a_cur[g] += a_step[g];
        }
#pragma unroll
        for (int g = 0; g < BG; ++g) {
            bh_cur[g] += 64;
            bl_cur[g] += 64;
        }
    };

    // ------------------------------------------------------------------ fragments / accumulators
    struct Frag {
        half8 ah, al, bh[2], bl[2];
    };
    const int r_sw = (lg ^ ((li >> 2) & 3)) << 3;
    const int k_flip = (r_sw ^ 16) - r_sw;
    const int a_row = (wm * 32 + li) * 32 + r_sw;                                // within an activation panel
    const int b_row = (wn * 64 + li) * 32 + r_sw;                                // within a weight panel
    floatx16 acc[2], accx[2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; accx[j][e] = 0.f; }
    };
    zero_acc();
    // the 6 MFMAs of one 16-wide K slice (weights first: D[channel][pixel]); between(m) runs after the m-th
    auto mfma_slice = [&](const Frag &f, auto &&between) {
        int m = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al, accx[j], 0, 0, 0);
            between(m++);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah, accx[j], 0, 0, 0);
            between(m++);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah, acc[j], 0, 0, 0);
            between(m++);
        }
    };
    // fragment piece r (0..5) of slice kk: activations from `abase` (panel pair hi | lo at distance apanel), weights from
    // `bbase` (hi | lo at distance PANEL_B)
    auto read_piece = [&](Frag &f, const _Float16 *abase, int apanel, const _Float16 *bbase, int kk, int r) {
        const int kf = kk ? k_flip : 0;
        if (r == 0) f.ah = *reinterpret_cast<const half8 *>(abase + a_row + kf);
        else if (r == 1) f.al = *reinterpret_cast<const half8 *>(abase + apanel + a_row + kf);
        else {
            const int j = (r - 2) >> 1;
            if (r & 1) f.bl[j] = *reinterpret_cast<const half8 *>(bbase + PANEL_B + b_row + j * 32 * 32 + kf);
            else f.bh[j] = *reinterpret_cast<const half8 *>(bbase + b_row + j * 32 * 32 + kf);
        }
    };

    // One pass over `nk` K tiles with an NS-stage ring.  issue(pc, stage) = DMA piece pc of the tile being prefetched,
    // next() = advance the source cursors to the following tile, aptr(tile, stage) = (base, panel distance) of the activation
    // panels of a tile, bofs = offset of the weight panels inside a stage, after(tile) = hook behind a tile's last MFMA
    // (returns true when it issued NST stores).  Schedule (per K tile, all waves in lockstep through one barrier):
    //   phase A: MFMAs of slice 0, slice 1 fetched from LDS behind the first three;  wait for tile t + 1 only (counted
    //            vmcnt), barrier;
    //   phase B: MFMAs of slice 1, slice 0 of tile t + 1 fetched behind them.
    // The DMA pieces of the tile being prefetched are pinned one behind each MFMA of phase A (PB = false: tile t + NS - 1 into
    // the stage tile t - 1 left, NS - 1 tiles in flight) or of phase B (PB = true: tile t + NS into the stage tile t has just
    // left, NS tiles in flight -- what a shallow ring needs).
    auto run = [&](auto ns_c, auto lpt_c, auto pb_c, int nk, _Float16 *ring, int stage_halves, int bofs, auto &&issue, auto &&next,
                   auto &&aptr, auto &&after) {
        constexpr int NS = decltype(ns_c)::value, LPT = decltype(lpt_c)::value;
        constexpr bool PB = decltype(pb_c)::value;
        int issued = 0;                       // tiles issued so far
        int store_mark = -1;                  // last tile issued BEFORE the most recent epilogue stores (-1: none pending)
        int load_mark = -1;                   // ... before the most recent residual prefetch (8 loads per lane)
        const int pre = min(PB ? NS : NS - 1, nk);
        for (int i = 0; i < pre; ++i) {
#pragma unroll
            for (int pc = 0; pc < LPT; ++pc) issue(pc, ring + i * stage_halves);
            next();
            ++issued;
        }
        blk_wait_barrier((issued - 1) * LPT);
        Frag f0, f1;
        {
            const _Float16 *ab;
            int ap;
            aptr(0, ring, ab, ap);
#pragma unroll
            for (int r = 0; r < 6; ++r) read_piece(f0, ab, ap, ring + bofs, 0, r);
        }
        int cs = 0, ls = NS - 1;
        int tt = 0;
        auto tile = [&](auto dma_c, auto next_c) {
            constexpr bool DMA = decltype(dma_c)::value, NEXT = decltype(next_c)::value;
            const _Float16 *cbase = ring + cs * stage_halves;
            _Float16 *lbase = ring + (PB ? cs : ls) * stage_halves;
            const _Float16 *ab;
            int ap;
            aptr(tt, cbase, ab, ap);
            mfma_slice(f0, [&](int m) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (r / 2 == m) {
                        __builtin_amdgcn_sched_barrier(0);
                        read_piece(f1, ab, ap, cbase + bofs, 1, r);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if (DMA && !PB) {
#pragma unroll
                    for (int pc = 0; pc < LPT; ++pc)
                        if (1 + pc * 5 / LPT == m) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(pc, lbase);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            });
            if (DMA && !PB) {
                next();
                ++issued;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): slice 1 is in registers
            // tile tt + 1 must have landed: everything issued after it may stay in flight
            int allow = 0;
            if (NEXT) {
                allow = (issued - 1 - (tt + 1)) * LPT;
                if (tt + 1 <= store_mark) allow += count_stores;  // the epilogue stores were issued after that tile
                if (tt + 1 <= load_mark) allow += 8;              // so was the residual prefetch
            }
            blk_wait_barrier(allow);
            ls = (ls + 1 == NS) ? 0 : ls + 1;
            cs = (cs + 1 == NS) ? 0 : cs + 1;
            const _Float16 *nbase = ring + cs * stage_halves;
            const _Float16 *nab = nullptr;
            int nap = 0;
            if (NEXT) aptr(tt + 1, nbase, nab, nap);
            __builtin_amdgcn_sched_barrier(0);
            mfma_slice(f1, [&](int m) {
                if (NEXT) {
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        if (r / 2 == m) {
                            __builtin_amdgcn_sched_barrier(0);
                            read_piece(f0, nab, nap, nbase + bofs, 0, r);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                if (DMA && PB) {
#pragma unroll
                    for (int pc = 0; pc < LPT; ++pc)
                        if (1 + pc * 5 / LPT == m) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(pc, lbase);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            });
            if (DMA && PB) {
                next();
                ++issued;
            }
            const int did = after(tt, issued);    // bit 0: NST stores issued, bit 1: 8 residual loads issued
            if (did & 1) store_mark = issued - 1;
            if (did & 2) load_mark = issued - 1;
            ++tt;
        };
        while (issued < nk) tile(std::true_type{}, std::true_type{});          // steady state: a tile is prefetched every tile
        while (tt + 1 < nk) tile(std::false_type{}, std::true_type{});         // drain
        if (tt < nk) tile(std::false_type{}, std::false_type{});               // last tile: nothing to fetch for
    };

    // ------------------------------------------------------------------ epilogue helpers
    // accumulators of n-subtile j -> scaled, biased (own 16 channels), then lanes l / l+32 exchange so that this lane holds
    // the two whole 8-channel groups g = 2 * pp + lg (pp = 0, 1) of its pixel: v[pp][0..7]
    auto finish_tile = [&](int j, float os, const float *bias, int ch0, float v[2][8]) {
        // bias through the scalar cache (ch0 is wave-uniform): s_load results count on lgkmcnt, so the epilogue neither waits
        // for nor drains the vector-memory queue the DMA ring lives in
        const cfloat_k *cb = reinterpret_cast<const cfloat_k *>(reinterpret_cast<unsigned long long>(bias)) + ch0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b0 = cb[8 * q + r], b1 = cb[8 * q + 4 + r];
                acc[j][4 * q + r] = (acc[j][4 * q + r] + accx[j][4 * q + r]) * os + (lg ? b1 : b0);
            }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[j][4 * (2 * pp) + r], y = acc[j][4 * (2 * pp + 1) + r];
                swap32(x, y);
                v[pp][r] = x;
                v[pp][4 + r] = y;
            }
    };

    // ================================================================== phase 1: conv2 (3x3)
    {
        auto issue = [&](int pc, _Float16 *sb) { dma1(pc, sb); };
        auto next = [&]() { advance1(); };
        auto aptr = [&](int, const _Float16 *stage, const _Float16 *&ab, int &ap) { ab = stage; ap = PANEL_A; };
        auto after = [&](int, int) { return 0; };
        if (p.stamp) ts[1] = __builtin_amdgcn_s_memrealtime();
        run(std::integral_constant<int, C::NS1>{}, std::integral_constant<int, C::LPT1>{}, std::integral_constant<bool, PB1>{}, K2T,
            blk_smem, C::STAGE1, 2 * PANEL_A, issue, next, aptr, after);
        if (p.stamp) ts[2] = __builtin_amdgcn_s_memrealtime();
    }
    // every wave is past the last barrier: nobody reads the ring any more.  Start the weight stream of conv3 (ring behind the
    // resident operand), then turn the accumulators into that operand.
    _Float16 *ring2 = blk_smem + C::T2;
    const char *w3h[BG], *w3l[BG];
    int kt2 = 0, nc2 = 0;
    auto w3_rebase = [&]() {
#pragma unroll
        for (int g = 0; g < BG; ++g) {
            const int n = nc2 * CM + (b_grp0 + g) * 16 + drow;
            const size_t off = ((size_t)n * CM + dchunk * 8) * 2;
            w3h[g] = reinterpret_cast<const char *>(p.w3_hi) + off;
            w3l[g] = reinterpret_cast<const char *>(p.w3_lo) + off;
        }
    };
    w3_rebase();
    auto issue2 = [&](int pb, _Float16 *sb) {
        const int g = BPW >= 2 ? (pb >> 1) : 0;
        const int half = BPW >= 2 ? (pb & 1) : b_half;
        blk_dma16(half ? w3l[g] : w3h[g], sb + half * PANEL_B + (b_grp0 + g) * 16 * 32);
    };
    auto next2 = [&]() {
        if (++kt2 == KT2) {
            kt2 = 0;
            ++nc2;
            w3_rebase();
        } else {
#pragma unroll
            for (int g = 0; g < BG; ++g) {
                w3h[g] += 64;
                w3l[g] += 64;
            }
        }
    };
    {
        // hand-off: T2[kt][hi|lo][pixel][32] with the operand chunk swizzle; this wave's 32 x 64 result = K tiles 2*wn, 2*wn+1
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[2][8];
            finish_tile(j, p.os2, p.bias2, wn * 64 + j * 32, v);
            _Float16 *panel = blk_smem + (size_t)(wn * 2 + j) * 2 * PANEL_A + (wm * 32 + li) * 32;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = fmaxf(v[pp][e], 0.f);                     // ReLU of conv2 (resnet.py:89)
                    if (!(x <= 65504.f)) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                    hi[e] = (_Float16)x;
                    lo[e] = (_Float16)(x - (float)hi[e]);
                }
                const int slot = ((2 * pp + lg) ^ ((li >> 2) & 3)) << 3;
                *reinterpret_cast<half8 *>(panel + slot) = hi;
                *reinterpret_cast<half8 *>(panel + PANEL_A + slot) = lo;
            }
        }
        zero_acc();
        if (p.stamp) ts[3] = __builtin_amdgcn_s_memrealtime();
    }
    // ================================================================== phase 2: conv3 (1x1) + residual + ReLU
    {
        const int pix = m0 + wm * 32 + li;
        const bool pix_ok = pix < p.M;
        const size_t row_off = (size_t)pix * (4 * CM) * 4;
        auto aptr = [&](int tile, const _Float16 *, const _Float16 *&ab, int &ap) {
            ab = blk_smem + (size_t)(tile % KT2) * 2 * PANEL_A;
            ap = PANEL_A;
        };
        // residual groups of a chunk (4 x 32 B per lane) are requested PF K tiles before the chunk's last MFMA, so that their
        // latency (L2 / MALL: ~2 us, exposed 4x per workgroup otherwise) runs behind the matrix pipe
        constexpr int PF = KT2 >= 4 ? 3 : 1;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 rh[2][2], rl[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                rh[j][pp] = u32x4{0, 0, 0, 0};
                rl[j][pp] = u32x4{0, 0, 0, 0};
            }
        auto after = [&](int tile, int) {
            int did = 0;
            if (tile % KT2 == KT2 - 1 - PF && !(p.flags & 2)) did = 2;
            if (did) {
                const int chp = (tile / KT2) * CM + wn * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        // (a pixel past M reads row 0: every wave issues the same 8 loads, which is what the vmcnt budget assumes)
                        const char *q = reinterpret_cast<const char *>(p.res) + (pix_ok ? row_off : 0) + (size_t)(chp + j * 32 + (2 * pp + lg) * 8) * 4;
                        rh[j][pp] = *reinterpret_cast<const u32x4 *>(q);
                        rl[j][pp] = *reinterpret_cast<const u32x4 *>(q + 16);
                    }
            }
            if (tile % KT2 != KT2 - 1) return did;
            const int nc = tile / KT2;
            const unsigned long long te0 = p.stamp ? __builtin_amdgcn_s_memrealtime() : 0;
            const int chb = nc * CM + wn * 64;                     // first channel of this wave in the chunk
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[2][8];
                finish_tile(j, p.os3, p.bias3, chb + j * 32, v);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const half8 h = __builtin_bit_cast(half8, rh[j][pp]), l = __builtin_bit_cast(half8, rl[j][pp]);
                    half8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = v[pp][e] + ((float)h[e] + (float)l[e]);
                        x = fmaxf(x, 0.f);
                        if (!(x <= 65504.f)) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                        hi[e] = (_Float16)x;
                        lo[e] = (_Float16)(x - (float)hi[e]);
                    }
                    if (pix_ok && !(p.flags & 4)) {
                        char *q = reinterpret_cast<char *>(p.y) + row_off + (size_t)(chb + j * 32 + (2 * pp + lg) * 8) * 4;
                        *reinterpret_cast<half8 *>(q) = hi;
                        *reinterpret_cast<half8 *>(q + 16) = lo;
                    }
                }
            }
            zero_acc();
            if (p.stamp) ts[5] += __builtin_amdgcn_s_memrealtime() - te0;
            return did | 1;
        };
        run(std::integral_constant<int, C::NS2>{}, std::integral_constant<int, C::LPT2>{}, std::integral_constant<bool, PB2>{}, 4 * KT2,
            ring2, C::STAGE2, 0, issue2, next2, aptr, after);
    }
    if (p.stamp && t == 0) {
        ts[4] = __builtin_amdgcn_s_memrealtime();
        unsigned long long *o = p.stamp + 8 * (size_t)blockIdx.x;
        for (int i = 0; i < 6; ++i) o[i] = ts[i];
        o[6] = 1;
    }
}

template <int CM, bool PB1, bool PB2>
static void launch_block_v(const BlockArgs &a, hipStream_t st)
{
    static bool configured = false;
    auto *k = conv_block_kernel<CM, PB1, PB2>;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BlockCfg<CM>::LDS);
        configured = true;
    }
    SRCNN_LAUNCH(k, dim3(a.mtiles), dim3(512), BlockCfg<CM>::LDS, st, a);
}

template <int CM>
static void launch_block(const BlockArgs &a, int variant, hipStream_t st)
{
    switch (variant & 3) {
    case 0: launch_block_v<CM, false, false>(a, st); break;
    case 1: launch_block_v<CM, true, false>(a, st); break;
    case 2: launch_block_v<CM, false, true>(a, st); break;
    default: launch_block_v<CM, true, true>(a, st); break;
    }
}

}  // namespace srcnn

extern "C" {

int srcnn_conv_block(const srcnn_block_desc *d, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(d && d->x && d->w2_hi && d->w2_lo && d->bias2 && d->w3_hi && d->w3_lo && d->bias3 && d->residual && d->y,
                  "null pointer");
    SRCNN_REQUIRE(d->C == 64 || d->C == 128 || d->C == 256, "C must be 64, 128 or 256 (layer1..3 of the ResNet trunk)");
    SRCNN_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "bad shape");
    const long long M = (long long)d->B * d->H * d->W;
    SRCNN_REQUIRE(M < (1LL << 29), "tensor too large");
    BlockArgs a;
    a.x = d->x; a.w2_hi = d->w2_hi; a.w2_lo = d->w2_lo; a.bias2 = d->bias2;
    a.w3_hi = d->w3_hi; a.w3_lo = d->w3_lo; a.bias3 = d->bias3;
    a.res = d->residual; a.y = d->y;
    a.zero_page = zero_page();
    SRCNN_REQUIRE(a.zero_page != nullptr, "zero page allocation failed");
    a.os2 = d->w2_inv_scale; a.os3 = d->w3_inv_scale;
    a.H = d->H; a.W = d->W; a.M = (int)M;
    a.mtiles = cdiv(a.M, 16384 / d->C);
    hipStream_t st = as_stream(stream);
    const bool prof = prof_enabled();
    if (prof) prof_begin(st);
    // timing-experiment knobs of profiles/fused_block_r02.txt (flags bit 1 / 2 produce INVALID results): compiled in only with
    // -DSRCNN_BLOCK_EXPERIMENTS, never in the shipped library.  variant: bit 0 PB schedule in phase 1, bit 1 in phase 2.
#ifdef SRCNN_BLOCK_EXPERIMENTS
    static const int env_flags = getenv("SRCNN_BLK_FLAGS") ? atoi(getenv("SRCNN_BLK_FLAGS")) : 0;
    static const int env_variant = getenv("SRCNN_BLK_VARIANT") ? atoi(getenv("SRCNN_BLK_VARIANT")) : 2;
#else
    const int env_flags = 0, env_variant = 2;
#endif
    a.flags = env_flags;
    a.stamp = debug_stamp_buffer();
    a.range_flag = range_flag_word();
    a.tag = d->layer_tag > 0 ? d->layer_tag : 0;
    SRCNN_REQUIRE(a.range_flag != nullptr, "range flag allocation failed");
    if (d->C == 64) launch_block<64>(a, env_variant, st);
    else if (d->C == 128) launch_block<128>(a, env_variant, st);
    else launch_block<256>(a, env_variant, st);
    if (prof) prof_end(st, 2.0 * (double)a.M * (double)d->C * (9.0 * d->C + 4.0 * d->C));
    return check_launch("srcnn_conv_block");
}

}  // extern "C"
