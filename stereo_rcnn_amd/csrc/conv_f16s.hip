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
template <int MR, int NR, bool OUT_SPLIT, int WM, int NS, int HEAD = 0>
__global__ __launch_bounds__(128 * WM, (MR * NR >= 16 ? 1 : 2)) void conv_f16s_kernel(const ConvArgs p)
{
    constexpr bool HEAD6 = HEAD == 1;
    static_assert(!HEAD6 || (MR == 2 && NR == 4 && WM == 4 && NS == 2 && !OUT_SPLIT), "the fused head lives in the 256x256 tile");
    static_assert(HEAD != 2 || (!OUT_SPLIT && WM == 4 && NS == 2 && ((MR == 2 && NR == 4) || (MR == 1 && NR == 2))),
                  "the MFMA-form head exists for the 256x256 and the 128x128 8-wave tiles");
    constexpr int NWAVES = 2 * WM, NTHREADS = 64 * NWAVES;
    constexpr int BM = 32 * MR * WM, BN = 64 * NR;
    constexpr int AG = BM / (16 * NWAVES), BG = BN / (16 * NWAVES);   // 16-row DMA groups per wave (A, B)
    static_assert(AG >= 1 && BG >= 1 && BM % (16 * NWAVES) == 0 && BN % (16 * NWAVES) == 0, "tile / wave count mismatch");
    constexpr int PANEL_A = BM * SROW, PANEL_B = BN * SROW;   // halves
    constexpr int STAGE = 2 * PANEL_A + 2 * PANEL_B;
    constexpr int LPT = 2 * (AG + BG);                        // DMA instructions per wave per K tile
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * LPT < 64, "ring depth");

    const int t = threadIdx.x;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;     // debug stamps (only when p.stamp is set)
    unsigned long long rt0 = 0;
    if (p.stamp) {
        st0 = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz, common to the whole chip
    }
    int mtiles = p.mtiles, nblk = p.mtiles * p.ntiles;
    const int bid = blockIdx.x;
    int m_rows = p.M;                                         // rows whose INPUT is read: beyond, the A operand is zeros
    if (p.m_limit) {
        // device-side row limit: tiles beyond it are not needed, and rows beyond it inside a computed tile never read their
        // (possibly stale, possibly other-format) input.  The XCD mapping below is taken over the tiles that DO run: over
        // the whole grid, the limited rows (the first logical tiles) would all be XCD 0's share and run on 32 CUs.
        const int lim = max(__builtin_amdgcn_readfirstlane(*p.m_limit), 0) * p.m_limit_mul;
        m_rows = min(m_rows, lim);
        mtiles = (m_rows + BM - 1) / BM;
        nblk = mtiles * p.ntiles;
        if (bid >= nblk) return;
    }
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    // consecutive logical tiles (one XCD's share) walk N inside an M tile -- they share the activation rows -- or, for the
    // fully connected shapes, M inside an N tile: they share the weight slab
    const int mt = p.m_fast ? logical % mtiles : logical / p.ntiles;
    const int nt = p.m_fast ? logical / mtiles : logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_begin = blockIdx.y * p.kt_per_split;
    const int kt_end = min(p.nkt, kt_begin + p.kt_per_split);

    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    // ---- DMA geometry: wave w fills rows [w*16*AG, +16*AG) of the A panels and [w*16*BG, +16*BG) of the B panels,
    //      16 rows x 4 chunks per instruction.  Addressing = buffer descriptors: A relative to the first input row this tile
    //      touches (so any tensor size works: a tile spans a few rows, possibly across an image boundary -- NHWC batches are
    //      contiguous --, and its 32-bit lane offsets stay small), B relative to the weight arrays.
    const int drow = lane >> 2;                               // row inside the 16-row group
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);         // source chunk (swizzle on the source side)
    const int ohw = p.OH * p.OW;
    const int b0 = m0 / ohw;                                   // first image of this tile (workgroup-uniform)
    const int row0 = max(((m0 - b0 * ohw) / p.OW) * p.stride - p.pad, 0);   // its first input row that a valid tap can touch
    const size_t base_bytes = (((size_t)b0 * p.H + row0) * p.W) * p.xcs * 4;
    rsrc_t rx = make_rsrc(reinterpret_cast<const char *>(p.x) + base_bytes, (size_t)p.nimg * p.H * p.W * p.xcs * 4 - base_bytes);
    const rsrc_t rwh = make_rsrc(p.w, (size_t)p.Cout * p.K * 2), rwl = make_rsrc(p.w_lo, (size_t)p.Cout * p.K * 2);
    // per-lane state of A group g: byte offset of the row's window origin relative to (image b0, row row0) plus the lane's
    // chunk (negative inside the padding: never used there), and the taps that fall inside the image as two bit fields --
    // bit kh: row ih0 + kh exists, bit 8 + kw: column iw0 + kw exists (KH, KW <= 7) -- so that a tap change costs a shift,
    // an and, an add and a select per group.  Rows past M have no valid tap.
    int a_org[AG], a_vm[AG];
#pragma unroll
    for (int g = 0; g < AG; ++g) {
        const int m = m0 + wave * 16 * AG + g * 16 + drow;
        a_org[g] = 0;
        a_vm[g] = 0;
        if (m < m_rows) {
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            a_org[g] = (((b - b0) * p.H + ih0 - row0) * p.W + iw0) * (p.xcs * 4) + dchunk * 32;
            int vm = 0;
            for (int k = 0; k < p.KH; ++k) vm |= ((unsigned)(ih0 + k) < (unsigned)p.H ? 1 : 0) << k;
            for (int k = 0; k < p.KW; ++k) vm |= ((unsigned)(iw0 + k) < (unsigned)p.W ? 1 : 0) << (8 + k);
            a_vm[g] = vm;
        }
    }
    // ---- second input (p.x2: the projection shortcut, K-concatenated behind the Cin channels of a 1x1 conv): its own
    //      descriptor and one lane offset per group -- pixel (oh * stride2, ow * stride2) of the row's image, or OOB past M.
    //      advance() switches the A stream over when the first input's channels are through.
    constexpr bool DUAL_OK = !(MR == 2 && NR == 4 && WM == 4 && NS == 2);     // not in the 256x256 tile (register budget)
    rsrc_t rx2 = rx;
    int a_off2[AG];
    int cin_cur = p.Cin;                                      // channels of the input the A stream is on (scalar)
    bool on_x2 = false;
    if constexpr (DUAL_OK) {
        if (p.x2) {
            const int row0_2 = ((m0 - b0 * ohw) / p.OW) * p.stride2;
            const size_t base2 = (((size_t)b0 * p.H2 + row0_2) * p.W2) * p.xcs2 * 4;
            rx2 = make_rsrc(reinterpret_cast<const char *>(p.x2) + base2, (size_t)p.nimg * p.H2 * p.W2 * p.xcs2 * 4 - base2);
#pragma unroll
            for (int g = 0; g < AG; ++g) {
                const int m = m0 + wave * 16 * AG + g * 16 + drow;
                a_off2[g] = OOB;
                if (m < m_rows) {
                    const int b = m / ohw;
                    const int rem = m - b * ohw;
                    const int oh = rem / p.OW;
                    const int ow = rem - oh * p.OW;
                    a_off2[g] = (((b - b0) * p.H2 + oh * p.stride2 - row0_2) * p.W2 + ow * p.stride2) * (p.xcs2 * 4) + dchunk * 32;
                }
            }
        }
    }
    // ---- DMA source cursors.  A: lane offset of the current tap's (row, chunk) run, or OOB when the tap falls outside
    //      the image / the row is past M; re-derived when the tap changes.  The channel tile inside the tap and B's K
    //      position are scalar offsets.
    int a_voff[AG], b_voff[BG];
    // K order.  p.tap_inner = 0: (tap, channel tile) as K runs in memory -- a tap's data is touched Cin/32 K tiles apart.
    // p.tap_inner = 1 (3x3 layers): (channel tile, tap) -- consecutive K tiles read the SAME input pixels shifted by one
    // tap, i.e. the cache lines the workgroup fetched one K tile earlier: the L2 serves the 3 taps of a row instead of the
    // fabric (profiles/k_order_r03.txt); the weights' K offset then jumps by Cin per K tile, on the scalar unit.
    int ld_kh, ld_kw, ld_c0, soff_b;
    {
        const int ntaps = p.KH * p.KW;
        const int tap = p.tap_inner ? kt_begin % ntaps : kt_begin / p.ctiles;
        const int ct = p.tap_inner ? kt_begin / ntaps : kt_begin - tap * p.ctiles;
        const int kh = tap / p.KW;
        ld_c0 = __builtin_amdgcn_readfirstlane(ct * BK);                            // keep the tap state scalar
        ld_kh = __builtin_amdgcn_readfirstlane(kh);
        ld_kw = __builtin_amdgcn_readfirstlane(tap - kh * p.KW);
        soff_b = __builtin_amdgcn_readfirstlane((tap * p.Cin + ct * BK) * 2);
    }
    auto retap = [&]() __attribute__((always_inline)) {
        const int tap_off = (ld_kh * p.W + ld_kw) * (p.xcs * 4);                      // wave-uniform
#pragma unroll
        for (int g = 0; g < AG; ++g) {
            const bool ok = ((a_vm[g] >> ld_kh) & (a_vm[g] >> (8 + ld_kw)) & 1) != 0;
            a_voff[g] = ok ? a_org[g] + tap_off : OOB;
        }
    };
    retap();
#pragma unroll
    for (int g = 0; g < BG; ++g) {
        const int n = n0 + wave * 16 * BG + g * 16 + drow;
        b_voff[g] = n < p.Cout ? (n * p.K + dchunk * 8) * 2 : OOB;
    }
    // one DMA instruction of the current K tile into ring stage at `stage_base` (halves); pc is a compile-time index
    // after unrolling: pieces 0..2*AG-1 = A (hi, lo per 16-row group), then B.
    auto dma_piece = [&](int pc, _Float16 *stage_base) __attribute__((always_inline)) {
        _Float16 *sa_hi = stage_base + (wave * 16 * AG) * SROW;
        _Float16 *sb_hi = stage_base + 2 * PANEL_A + (wave * 16 * BG) * SROW;
        const int soff_a = ld_c0 * 4;
        if (pc < 2 * AG) {
            const int g = pc >> 1;
            // (the lo half sits 16 B behind the hi half: on the scalar offset -- the instruction's immediate offset would
            //  move the LDS destination as well)
            if (pc & 1) dma16b<0>(rx, a_voff[g], soff_a + 16, sa_hi + PANEL_A + g * 16 * SROW);
            else dma16b<0>(rx, a_voff[g], soff_a, sa_hi + g * 16 * SROW);
        } else {
            const int g = (pc - 2 * AG) >> 1;
            if (pc & 1) dma16b<0>(rwl, b_voff[g], soff_b, sb_hi + PANEL_B + g * 16 * SROW);
            else dma16b<0>(rwh, b_voff[g], soff_b, sb_hi + g * 16 * SROW);
        }
    };
    // move the cursors to the next K tile
    auto advance = [&]() __attribute__((always_inline)) {
        if (p.tap_inner) {                                    // next tap of the same channel tile; after the last, next tile
            soff_b += p.Cin * 2;
            if (++ld_kw == p.KW) {
                ld_kw = 0;
                if (++ld_kh == p.KH) {
                    ld_kh = 0;
                    ld_c0 += BK;
                    soff_b += (BK - p.KH * p.KW * p.Cin) * 2;
                }
            }
            retap();
            return;
        }
        ld_c0 += BK;
        soff_b += BK * 2;
        if constexpr (DUAL_OK) {
            if (p.x2 && !on_x2 && ld_c0 == cin_cur) {         // first input through: the A stream moves to the second one
                on_x2 = true;
                cin_cur = p.Cin2;
                ld_c0 = 0;
                rx = rx2;
#pragma unroll
                for (int g = 0; g < AG; ++g) a_voff[g] = a_off2[g];
                return;
            }
        }
        if (ld_c0 == cin_cur) {
            ld_c0 = 0;
            if (++ld_kw == p.KW) { ld_kw = 0; ++ld_kh; }
            retap();
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lg = lane >> 5;
    const int r_sw = (lg ^ ((li >> 2) & 3)) << 3;
    constexpr bool XACC = (MR * NR <= 2);
    constexpr int NX = XACC ? (MR * NR == 1 ? 2 : 1) : 0;
    floatx16 acc[MR][NR];
    floatx16 accx[NX > 0 ? NX : 1][MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
#pragma unroll
                for (int x = 0; x < (NX > 0 ? NX : 1); ++x) accx[x][i][j][e] = 0.f;
            }

    // MFMA operand fragments of one 16-wide K slice
    struct Frag {
        half8 ah[MR], al[MR], bh[NR], bl[NR];
    };
    const int a_row = (wm * 32 * MR + li) * SROW + r_sw, b_row = 2 * PANEL_A + (wn * 32 * NR + li) * SROW + r_sw;
    const int k_flip = (r_sw ^ 16) - r_sw;                     // second 16-wide slice of the swizzled row
    constexpr int NRD = 2 * (MR + NR);                        // ds_read_b128 per slice
    auto read_piece = [&](Frag &f, const _Float16 *stage_base, int kk, int r) __attribute__((always_inline)) {
        const _Float16 *sah = stage_base + a_row + (kk ? k_flip : 0);
        const _Float16 *sbh = stage_base + b_row + (kk ? k_flip : 0);
        if (r < 2 * MR) {
            const int i = r >> 1;
            if (r & 1) f.al[i] = *reinterpret_cast<const half8 *>(sah + PANEL_A + i * 32 * SROW);
            else f.ah[i] = *reinterpret_cast<const half8 *>(sah + i * 32 * SROW);
        } else {
            const int j = (r - 2 * MR) >> 1;
            if (r & 1) f.bl[j] = *reinterpret_cast<const half8 *>(sbh + PANEL_B + j * 32 * SROW);
            else f.bh[j] = *reinterpret_cast<const half8 *>(sbh + j * 32 * SROW);
        }
    };
    auto read_frag = [&](Frag &f, const _Float16 *stage_base, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NRD; ++r) read_piece(f, stage_base, kk, r);
    };
    // fetch of a slice spread behind the first MFMAs of the other slice, two reads per MFMA: the matrix pipe is fed
    // before the LDS queue (all waves fetch at the same moment, right after the barrier) has drained
    auto read_slot = [&](Frag &f, const _Float16 *stage_base, int kk, int m) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NRD; ++r)
            if (r / 2 == m) {
                __builtin_amdgcn_sched_barrier(0);
                read_piece(f, stage_base, kk, r);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // the 3*MR*NR MFMAs of one slice; `between(m)` runs after the m-th (DMA pieces are slotted in there so that
    // their issue cost hides under the matrix pipe instead of forming a block in which no wave of the SIMD computes)
    constexpr int NM = 3 * MR * NR;
    auto mfma_slice = [&](const Frag &f, auto &&between) __attribute__((always_inline)) {
        int m = 0;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                floatx16 &d = NX > 0 ? accx[0][i][j] : acc[i][j];
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], d, 0, 0, 0);
                between(m++);
            }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                floatx16 &d = NX > 1 ? accx[NX > 1 ? 1 : 0][i][j] : (NX > 0 ? accx[0][i][j] : acc[i][j]);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], d, 0, 0, 0);
                between(m++);
            }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
                between(m++);
            }
    };

    // ---- K loop.  Two forms.
    // WIDE (256x256 workgroup tile, 8 waves of 64x128, 2-stage ring): the wave's 128 accumulators leave no room for two
    // whole operand-fragment sets, so the B fragments (the wide side) live in ONE buffer that is refreshed in place, a pair
    // of 32-column blocks at a time, as soon as the MFMAs that read it have issued; only the narrow A side is double-buffered.
    // A K tile is four half-steps of 12 MFMAs (slice s, column pair p):  H0 (0,0)  H1 (0,1)  H2 (1,0)  [P]  H3 (1,1).
    //   H0 fetches B pair 1 of slice 0 and A of slice 1; H1 B pair 0 of slice 1; H2 B pair 1 of slice 1 -- the last read of
    //   this tile's stage; P = wait for tile kt+1's DMA + barrier (everybody is done with this stage, tile kt+1 is visible);
    //   H3 fetches B pair 0 and A of slice 0 of tile kt+1.  The DMA of tile kt+2 into the stage just freed is issued 4 pieces
    //   behind the reads of H3 and 4 behind those of the next H0, so it has H1 + H2 of every wave to land.
    // One barrier per 48 MFMAs of a wave; 12 fragment reads per 24 MFMAs (16 with the 64x64 per-wave tile of 256x128).
    constexpr bool WIDE = (MR == 2 && NR == 4 && WM == 4 && NS == 2);
    unsigned long long w_vm = 0, w_bar = 0;                       // debug: time in the steady-state vmcnt waits / barriers
    const int nk = kt_end - kt_begin;
    if (p.stamp) st1 = __builtin_readcyclecounter();
    if constexpr (WIDE) {
        static_assert(!WIDE || LPT == 8, "DMA pieces per K tile");
        half8 fa_h[2][MR], fa_l[2][MR], fb_h[NR], fb_l[NR];
        const int a_row = (wm * 32 * MR + li) * SROW + r_sw, b_row = 2 * PANEL_A + (wn * 32 * NR + li) * SROW + r_sw;
        const int k_flip = (r_sw ^ 16) - r_sw;
        // r = 0..2*MR-1: (row block r>>1, hi / lo) of slice kk into A buffer s
        auto rd_a = [&](int sbuf, const _Float16 *stage_base, int kk, int r) __attribute__((always_inline)) {
            const _Float16 *sah = stage_base + a_row + (kk ? k_flip : 0) + (r >> 1) * 32 * SROW;
            if (r & 1) fa_l[sbuf][r >> 1] = *reinterpret_cast<const half8 *>(sah + PANEL_A);
            else fa_h[sbuf][r >> 1] = *reinterpret_cast<const half8 *>(sah);
        };
        // r = 0..3: (column block 2*pr + (r>>1), hi / lo) of slice kk
        auto rd_b = [&](const _Float16 *stage_base, int kk, int pr, int r) __attribute__((always_inline)) {
            const int j = 2 * pr + (r >> 1);
            const _Float16 *sbh = stage_base + b_row + (kk ? k_flip : 0) + j * 32 * SROW;
            if (r & 1) fb_l[j] = *reinterpret_cast<const half8 *>(sbh + PANEL_B);
            else fb_h[j] = *reinterpret_cast<const half8 *>(sbh);
        };
        // 12 MFMAs: rows 0..1 x columns (2*pr, 2*pr+1) x the three products; four accumulators in rotation
        auto half_step = [&](int sbuf, int pr, auto &&between) __attribute__((always_inline)) {
            int m = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * pr + jj;
                        const half8 &a = t == 0 ? fa_l[sbuf][i] : fa_h[sbuf][i];
                        const half8 &b = t == 1 ? fb_l[j] : fb_h[j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
                        between(m++);
                    }
        };
        auto pinned = [&](auto &&fn) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            fn();
            __builtin_amdgcn_sched_barrier(0);
        };
        // reads of a half-step: the 4 B reads behind MFMAs 0 and 1, the 4 A reads (if any) behind MFMAs 2 and 3
        auto reads_b = [&](const _Float16 *base, int kk, int pr, int m) __attribute__((always_inline)) {
            if (m < 2) pinned([&]() __attribute__((always_inline)) { rd_b(base, kk, pr, 2 * m); rd_b(base, kk, pr, 2 * m + 1); });
        };
        auto reads_a = [&](int sbuf, const _Float16 *base, int kk, int m) __attribute__((always_inline)) {
            if (m == 2 || m == 3) pinned([&]() __attribute__((always_inline)) { rd_a(sbuf, base, kk, 2 * (m - 2)); rd_a(sbuf, base, kk, 2 * (m - 2) + 1); });
        };
        // DMA pieces first..first+3 of the tile at the cursors, behind MFMAs 5, 7, 9, 11
        auto dma4 = [&](int first, _Float16 *lbase, int m) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m == 5 + 2 * q) pinned([&]() __attribute__((always_inline)) { dma_piece(first + q, lbase); });
        };
        // prologue: tile 0 whole, the first half of tile 1; tile 0's first fragments
        {
#pragma unroll
            for (int pc = 0; pc < LPT; ++pc) dma_piece(pc, smem);
            advance();
            if (nk > 1) {
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) dma_piece(pc, smem + STAGE);
                wait_vm_barrier<4>();
            } else {
                wait_vm_barrier<0>();
            }
        }
        if (p.stamp) st2 = __builtin_readcyclecounter();
#pragma unroll
        for (int r = 0; r < 4; ++r) rd_b(smem, 0, 0, r);
#pragma unroll
        for (int r = 0; r < 2 * MR; ++r) rd_a(0, smem, 0, r);
        if (wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);
        int cs = 0;
        // D1: tile kt+1 exists (its pieces 4..7 go out in H0, its first fragments are fetched in H3); D2: tile kt+2 exists
        auto k_tile = [&](auto d1, auto d2) __attribute__((always_inline)) {
            constexpr bool D1 = decltype(d1)::value, D2 = decltype(d2)::value;
            const _Float16 *cbase = smem + cs * STAGE;
            _Float16 *obase = smem + (cs ^ 1) * STAGE;
            half_step(0, 0, [&](int m) __attribute__((always_inline)) {
                reads_b(cbase, 0, 1, m);
                reads_a(1, cbase, 1, m);
                if (D1) dma4(4, obase, m);
            });
            if (D1) advance();
            half_step(0, 1, [&](int m) __attribute__((always_inline)) { reads_b(cbase, 1, 0, m); });
            half_step(1, 0, [&](int m) __attribute__((always_inline)) { reads_b(cbase, 1, 1, m); });
            __builtin_amdgcn_sched_barrier(0);                // H2's MFMAs stay above the barrier: the wait comes as late as it can
            __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the compiler's own count knows the reads are in
            wait_vm_barrier<0>();                             // tile kt+1 has landed everywhere; this stage is free
            __builtin_amdgcn_sched_barrier(0);
            half_step(1, 1, [&](int m) __attribute__((always_inline)) {
                if (D1) {
                    reads_b(obase, 0, 0, m);
                    reads_a(0, obase, 0, m);
                }
                if (D2) dma4(0, smem + cs * STAGE, m);
            });
            cs ^= 1;
        };
        int kt = kt_begin;
        for (; kt + 2 < kt_end; ++kt) k_tile(std::true_type{}, std::true_type{});
        if (kt + 1 < kt_end) {
            k_tile(std::true_type{}, std::false_type{});
            ++kt;
        }
        k_tile(std::false_type{}, std::false_type{});
    } else {
        // ---- prologue: fill the ring, wait for the first tile, fetch its first slice.
        // Two issue schedules for the DMA of a K tile (PB, per ring depth):
        //   PB = false: tile kt+NS-1 is issued during phase A of tile kt (between slice 0's MFMAs) -> NS-1 tiles in flight;
        //   PB = true : tile kt+NS is issued during phase B of tile kt, right behind the barrier that frees the stage tile kt
        //               occupied -> NS tiles in flight, half a K tile more latency cover from the same LDS.  The shallow
        //               rings need it (a 2-stage ring otherwise waits for every tile: profiles/stamp_conv_r01.txt).
        constexpr bool PB = (NS <= SRCNN_PB_MAX_NS);
        constexpr int PRE = PB ? NS : NS - 1;                     // tiles issued before the first MFMA
        {
            const int pre = min(PRE, nk);
            for (int i = 0; i < pre; ++i) {
    #pragma unroll
                for (int pc = 0; pc < LPT; ++pc) dma_piece(pc, smem + i * STAGE);
                advance();
            }
            if (PRE >= 4 && pre == 4) wait_vm_barrier<3 * LPT>();
            else if (PRE >= 3 && pre == 3) wait_vm_barrier<2 * LPT>();
            else if (PRE >= 2 && pre == 2) wait_vm_barrier<LPT>();
            else wait_vm_barrier<0>();
        }
        if (p.stamp) st2 = __builtin_readcyclecounter();
        Frag f0, f1;
        read_frag(f0, smem, 0);
        int cs = 0, ls = NS - 1;                                  // compute stage / load stage (PB = false) of the ring
        // One K tile.  Entry: slice 0 of tile kt is in f0.  Phase A: run slice 0's MFMAs, fetching slice 1 behind the first
        // of them (and, PB = false, with the DMA pieces of tile kt+NS-1 slotted between them).  Then wait until tile kt+1
        // (only) has landed, barrier (every wave has finished reading this stage, everybody's part of tile kt+1 is visible).
        // Phase B: run slice 1's MFMAs, fetching slice 0 of tile kt+1 behind the first of them (and, PB = true, with the DMA
        // pieces of tile kt+NS going into the stage just freed).  LDS latency and DMA issue never stall a wave's MFMAs.
        auto k_tile = [&](auto with_dma, int n_after, bool has_next) __attribute__((always_inline)) {
            constexpr bool DMA = decltype(with_dma)::value;
            const _Float16 *cbase = smem + cs * STAGE;
            _Float16 *lbase = smem + (PB ? cs : ls) * STAGE;
            mfma_slice(f0, [&](int m) __attribute__((always_inline)) {
                read_slot(f1, cbase, 1, m);                      // slice 1: not needed before phase B
                if (DMA && !PB) {
    #pragma unroll
                    for (int pc = 0; pc < LPT; ++pc)
                        if (1 + pc * (NM - 1) / LPT == m) {
                            __builtin_amdgcn_sched_barrier(0);    // pin the piece to its slot
                            dma_piece(pc, lbase);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            });
            if (DMA && !PB) advance();
            // slice 1 has landed in registers -- said with the builtin so that the compiler's own wait-count tracking knows
            // it (it cannot see into the asm below) and puts no lgkmcnt wait between the next fetch and slice 1's MFMAs
            __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0), vmcnt / expcnt untouched
            // tiles issued after tile kt+1 at this point: kt+2 .. min(kt+NS-1, last) in either schedule
            if (n_after == NS - 2 && p.stamp) wait_vm_barrier_timed<(NS - 2) * LPT>(w_vm, w_bar);
            else if (n_after == NS - 2) wait_vm_barrier<(NS - 2) * LPT>();
            else if (NS >= 4 && n_after == 1) wait_vm_barrier<LPT>();
            else wait_vm_barrier<0>();
            ls = (ls + 1 == NS) ? 0 : ls + 1;
            cs = (cs + 1 == NS) ? 0 : cs + 1;
            const _Float16 *nbase = smem + cs * STAGE;
            __builtin_amdgcn_sched_barrier(0);
            mfma_slice(f1, [&](int m) __attribute__((always_inline)) {
                if (has_next) read_slot(f0, nbase, 0, m);        // slice 0 of the next tile: needed at the next phase A
                if (DMA && PB) {
    #pragma unroll
                    for (int pc = 0; pc < LPT; ++pc)
                        if (1 + pc * (NM - 1) / LPT == m) {
                            __builtin_amdgcn_sched_barrier(0);
                            dma_piece(pc, lbase);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            });
            if (DMA && PB) advance();
        };
        // 8-wave workgroups: the second-dispatched half loses every VALU arbitration to its older SIMD sibling; one static
        // priority raise for that half (no per-phase flips) evens the pair out (MI355X_MICROARCH.md, two waves per SIMD)
        if (WM == 4 && wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);
        int kt = kt_begin;
        for (; kt + PRE < kt_end; ++kt) k_tile(std::true_type{}, NS - 2, true);
        for (; kt < kt_end; ++kt) k_tile(std::false_type{}, min(NS - 2, kt_end - 2 - kt), kt + 1 < kt_end);
    }
    if (p.stamp) st3 = __builtin_readcyclecounter();
    if (NX > 0) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float xs = accx[0][i][j][e];
                    if (NX > 1) xs += accx[NX > 1 ? 1 : 0][i][j][e];
                    acc[i][j][e] += xs;
                }
    }

    // ---- epilogue
    const bool split = gridDim.y > 1;
    const float os = p.out_scale;
    // Fast path (every layer of the network except the 6-channel keypoint classifier and the deconv
    // scatter): the accumulator tile is transposed through the now idle operand LDS so that each lane
    // owns 8 consecutive channels of one pixel -> bias / residual / ReLU / SPLIT16 re-split on 8 values,
    // residual read and result written with 16-byte accesses (2 per group instead of 16 two-byte ones).
    const int cq = p.mode == 1 ? (p.Cout >> 2) : p.Cout;     // channels per output pixel (deconv: Cout = 4 taps x cq)
    if ((cq & 7) == 0 && (p.ycs & 7) == 0 && (p.yco & 7) == 0 && (!p.res || (p.rcs & 7) == 0)) {
        // The accumulator tile goes through the operand LDS in PASSES row blocks of RPP rows (1 pass for every tile up to
        // 256x128; the 256x256 tile of the 4-wave / 512-register configuration does not fit at once and takes 2).
        constexpr int PASSES = (BM * BN * 4 + NS * STAGE * 2 - 1) / (NS * STAGE * 2);
        constexpr int RPP = BM / PASSES;
        static_assert((BM % PASSES == 0 && RPP % (32 * MR) == 0) || PASSES == 1, "a pass is a whole number of per-wave row blocks");
        float *tile = reinterpret_cast<float *>(smem);       // [RPP][BN] floats <= the operand ring
        // thread -> (8-channel group g, rows r0 + it * RSTEP): the group is the same in every iteration, so bias and column
        // tests are loop invariants, and the NG residual groups of the thread are independent 32-byte loads that are all
        // put in flight BEFORE the accumulators go through the LDS (one memory latency per workgroup instead of one per
        // iteration: the conv3 + residual layers of the trunk are epilogue-bound, profiles/stamp_conv_r01.txt)
        constexpr int GROUPS = BN / 8, NG = RPP * GROUPS / NTHREADS, RSTEP = NTHREADS / GROUPS;
        static_assert(RPP * GROUPS % NTHREADS == 0 && NTHREADS % GROUPS == 0, "epilogue mapping");
        const int g = t % GROUPS, r0 = t / GROUPS;
        const int col = n0 + g * 8;
        const bool col_ok = col < p.Cout;
        // deconv (mode 1): column = (tap ij, channel co); the 8-channel group never straddles a tap
        const int ij = p.mode == 1 ? col / cq : 0;
        const int co = col - ij * cq;
        const bool use_res = p.res && !split && col_ok;
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        if (p.bias && !split && col_ok) {
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + co);
            const float4 b1 = *reinterpret_cast<const float4 *>(p.bias + co + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
            bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
        int head_slice = -1;                                     // HEAD 2: the eye whose W2 slice + bias sit in LDS (workgroup-uniform)
        // one pass as a function of the COMPILE-TIME pass index: a run-time `ps` loop that the optimizer declines to unroll
        // (it did, for the two-pass tiles) would index the accumulators dynamically and push all of them into scratch
        auto one_pass = [&](auto ps_c) __attribute__((always_inline)) {
            constexpr int ps = decltype(ps_c)::value;
            const int mp = m0 + ps * RPP;                    // first output row of this pass
            // residual groups put in flight BEFORE the accumulators go through the LDS: all NG of them, except for the
            // 256x256 tile, whose 128 live accumulators leave room for 4 (the rest are loaded where they are used; a spill
            // would cost the same trip through memory twice)
            constexpr int PF = (MR * NR >= 8 && NG > 4) ? 4 : NG;
            uint4 res_a[PF], res_b[PF];
            auto load_res = [&](int it, uint4 &ra, uint4 &rb) __attribute__((always_inline)) {
                const int row = mp + r0 + it * RSTEP;
                ra = make_uint4(0, 0, 0, 0);
                rb = make_uint4(0, 0, 0, 0);
                if (use_res && row < p.M) {
                    const char *q = reinterpret_cast<const char *>(p.res) + ((size_t)row * p.rcs + (size_t)col) * 4;
                    ra = *reinterpret_cast<const uint4 *>(q);
                    rb = *reinterpret_cast<const uint4 *>(q + 16);
                }
            };
#pragma unroll
            for (int it = 0; it < PF; ++it) load_res(it, res_a[it], res_b[it]);
            if (ps > 0) __syncthreads();                     // the previous pass has been read out of the tile
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int rb = (wm * MR + i) * 32 - ps * RPP;     // this 32-row block inside the pass (wave-uniform)
                if (PASSES == 1 || (rb >= 0 && rb < RPP)) {
#pragma unroll
                    for (int j = 0; j < NR; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int r = rb + (e & 3) + 8 * (e >> 2) + 4 * lg;
                            const int c = (wn * NR + j) * 32 + li;
                            // HEAD 2 reads the tile back one ROW per lane (MFMA A fragments): the 8-float groups of a row are
                            // XOR-swizzled by the row so that 32 rows do not meet in one bank
                            tile[r * BN + (HEAD == 2 ? (c ^ ((r & 7) << 3)) : c)] = acc[i][j][e] * os;
                        }
                }
            }
            __syncthreads();
            if (p.stamp && ps == 0) st4 = __builtin_readcyclecounter();
            if constexpr (HEAD == 2) {
                // ---- narrow head as a second GEMM on the matrix pipe:  out[row][n] = sum_c act(tile[row][c] + bias[c]) * W2[n][c]
                // over the BN columns of this tile.  A 32-row block per wave: lane (row li, k group lg) reads 8 consecutive tile
                // columns per 16-wide K step, adds bias, ReLUs and splits into the hi / lo f16 fragments IN REGISTERS; the W2
                // fragments of the tile's columns sit in LDS behind the tile in fragment order (one 16-byte read per lane, no
                // conflicts).  3 products per step into one 32x32 accumulator whose lanes li < head_n hold 16 rows of output n = li.
                // Mode 2 (stereo pair launch): rows of the second half of the batch are the other eye and meet another slice of W2 --
                // a tile that straddles the halves runs the K steps once per eye with the other eye's rows zeroed.
                constexpr int NBLK = RPP / 32, KS = BN / 16;
                static_assert(NBLK <= NWAVES, "one 32-row block per wave");
                float *hb = reinterpret_cast<float *>(smem + NS * STAGE);                 // [BN] bias of the tile's columns
                half8 *hw = reinterpret_cast<half8 *>(hb + BN);                            // [KS][hi, lo][lg][head_rows] fragments
                const int n2p = p.head_rows;
                const int colbase = p.mode == 1 ? n0 - (n0 / cq) * cq : n0;               // first column's channel within the pixel
                const int half_rows = (p.nimg >> 1) * p.OH * p.OW;
                const int m_hi = min(mp + RPP, p.M);
                if (mp < m_hi) {                                                            // (workgroup-uniform)
                    const int pe_lo = (p.mode == 2 && mp >= half_rows) ? 1 : 0;
                    const int pe_hi = (p.mode == 2 && m_hi - 1 >= half_rows) ? 1 : 0;
                    floatx16 acc2;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
                    float gmax = 0.f;
                    bool bad = false;
                    const int row_l = wave * 32 + li;                                       // this lane's A row inside the pass
                    const int row_g = mp + row_l;
                    for (int eye = pe_lo; eye <= pe_hi; ++eye) {
                        if (head_slice != eye) {                                            // (uniform) stage bias + W2 slice of this eye
                            if (head_slice >= 0) __syncthreads();                           // readers of the previous slice are through
                            for (int i = t; i < BN; i += NTHREADS)
                                hb[i] = (p.bias && n0 + i < p.Cout) ? p.bias[colbase + i] : 0.f;
                            const int s0 = ((p.mode == 2 ? eye * p.Cout : 0) + colbase) >> 4;
                            const uint4 *src = reinterpret_cast<const uint4 *>(p.head_wf) + (size_t)s0 * n2p * 4;
                            uint4 *dst = reinterpret_cast<uint4 *>(hw);
                            for (int i = t; i < KS * n2p * 4; i += NTHREADS) dst[i] = src[i];
                            __syncthreads();
                            head_slice = eye;
                        }
                        if (wave < NBLK) {
                            const bool mine = row_g < p.M && (p.mode != 2 || ((row_g >= half_rows) ? 1 : 0) == eye);
                            const float *trow = tile + row_l * BN;
                            const int sw = row_l & 7;
#pragma unroll 2
                            for (int s = 0; s < KS; ++s) {
                                const int grp = (2 * s + lg) ^ sw;
                                const float4 a0 = *reinterpret_cast<const float4 *>(trow + grp * 8);
                                const float4 a1 = *reinterpret_cast<const float4 *>(trow + grp * 8 + 4);
                                const float4 b0 = *reinterpret_cast<const float4 *>(hb + 16 * s + 8 * lg);
                                const float4 b1 = *reinterpret_cast<const float4 *>(hb + 16 * s + 8 * lg + 4);
                                float v[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
                                half8 ah, al;
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    bad = bad || (v[i] != v[i]);                           // before the ReLU launders a NaN
                                    if (p.relu) v[i] = fmaxf(v[i], 0.f);
                                    if (!mine) v[i] = 0.f;
                                    gmax = fmaxf(gmax, fabsf(v[i]));
                                    ah[i] = (_Float16)v[i];
                                    al[i] = (_Float16)(v[i] - (float)ah[i]);
                                }
                                half8 bh, bl;
#pragma unroll
                                for (int i = 0; i < 8; ++i) { bh[i] = (_Float16)0.f; bl[i] = (_Float16)0.f; }
                                if (li < n2p) {
                                    bh = hw[((s * 2 + 0) * 2 + lg) * n2p + li];
                                    bl = hw[((s * 2 + 1) * 2 + lg) * n2p + li];
                                }
                                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
                                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2, 0, 0, 0);
                            }
                        }
                    }
                    // the activations never reach memory, so the range guard of the SPLIT16 store is applied here (hi = f16(v))
                    if (wave < NBLK && (bad || !(gmax <= 65504.f))) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                    if (wave < NBLK && li < p.head_n) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int row = mp + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
                            if (row >= p.M) continue;
                            if (p.head_parts == 0) {
                                size_t opix = (size_t)row;
                                if (p.mode == 1) {                   // 2x2 stride-2 scatter: pixel (2*oh + i, 2*ow + j); the tile is one tap
                                    const int ij2 = n0 / cq;
                                    const int ohw2 = p.OH * p.OW;
                                    const int bb = row / ohw2, rem = row - bb * ohw2;
                                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                                    opix = ((size_t)bb * 2 * p.OH + 2 * oh + (ij2 >> 1)) * (2 * p.OW) + 2 * ow + (ij2 & 1);
                                }
                                p.head_y[opix * p.head_n + li] = fmaf(acc2[e], p.head_scale, p.head_b[li]);
                            } else {
                                const int eye = (p.mode == 2 && row >= half_rows) ? 1 : 0;
                                const size_t px = (size_t)(row - eye * half_rows);
                                p.head_y[(size_t)(eye * p.ntiles + nt) * p.head_plane + px * p.head_n + li] = acc2[e] * p.head_scale;
                            }
                        }
                    }
                }
                return;
            }
            if constexpr (HEAD6) {
                // Fused 6-channel head (srcnn_conv_desc.head_w) instead of the y store.  GROUPS == 32 lanes hold the 256 channels
                // of a pixel (lanes 0-31 / 32-63 of a wave: two pixels): 8 channels per lane in order, then a DPP scan over the
                // half-wave -- a fixed summation order.  Three filters per sweep over the pass's rows: their 24 weights are
                // fetched AFTER the pass's accumulators have left the registers (128 accumulators + 48 weights + the unrolled
                // row loop do not fit the 256 registers of a wave); the second sweep re-reads the tile from LDS.
                static_assert(!HEAD6 || GROUPS == 32, "one half-wave per pixel");
#pragma unroll
                for (int kk = 0; kk < 6; kk += 3) {
                    float hw[3][8], hb[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 w0 = *reinterpret_cast<const float4 *>(p.head_w + (kk + k) * cq + co);
                        const float4 w1 = *reinterpret_cast<const float4 *>(p.head_w + (kk + k) * cq + co + 4);
                        hw[k][0] = w0.x; hw[k][1] = w0.y; hw[k][2] = w0.z; hw[k][3] = w0.w;
                        hw[k][4] = w1.x; hw[k][5] = w1.y; hw[k][6] = w1.z; hw[k][7] = w1.w;
                        hb[k] = p.head_b[kk + k];
                    }
#pragma unroll
                    for (int it = 0; it < NG; ++it) {
                        const int r = r0 + it * RSTEP;
                        const int row = mp + r;
                        if (row < p.M) {                          // uniform per half-wave (one pixel)
                            const float4 a = *reinterpret_cast<const float4 *>(tile + r * BN + g * 8);
                            const float4 b = *reinterpret_cast<const float4 *>(tile + r * BN + g * 8 + 4);
                            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if (p.bias) v[e] += bias8[e];
                                if (p.relu) v[e] = fmaxf(v[e], 0.f);
                            }
                            float hs[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float acc1 = v[0] * hw[k][0];
#pragma unroll
                                for (int e = 1; e < 8; ++e) acc1 = fmaf(v[e], hw[k][e], acc1);
                                hs[k] = acc1;
                            }
                            // sum over the 32 lanes of the pixel on the VALU (DPP), not through the LDS crossbar: inclusive scan
                            // inside each 16-lane row (row_shr 1, 2, 4, 8; lanes shifted in from outside read 0), then lane 15 of
                            // rows 0 / 2 added into rows 1 / 3 (row_bcast:15): lane 31 of each half-wave holds the pixel's sum
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float x = hs[k];
                                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
                                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
                                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
                                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
                                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, true));
                                hs[k] = x;
                            }
                            if (g == 31) {
                                size_t opix = (size_t)row;
                                if (p.mode == 1) {                   // 2x2 stride-2 scatter: pixel (2*oh + i, 2*ow + j)
                                    const int ohw2 = p.OH * p.OW;
                                    const int bb = row / ohw2, rem = row - bb * ohw2;
                                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                                    opix = ((size_t)bb * 2 * p.OH + 2 * oh + (ij >> 1)) * (2 * p.OW) + 2 * ow + (ij & 1);
                                }
                                float *dst = p.head_y + opix * 6 + kk;
#pragma unroll
                                for (int k = 0; k < 3; ++k) dst[k] = fmaf(hs[k], p.head_scale, hb[k]);
                            }
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const int r = r0 + it * RSTEP;
                const int row = mp + r;
                if (row < p.M && col_ok) {
                    float8 v;
                    const float4 a = *reinterpret_cast<const float4 *>(tile + r * BN + g * 8);
                    const float4 b = *reinterpret_cast<const float4 *>(tile + r * BN + g * 8 + 4);
                    v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w;
                    v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
                    if (split) {
                        float *dst = p.partial + ((size_t)blockIdx.y * p.M + row) * p.Cout + col;
                        *reinterpret_cast<float4 *>(dst) = a;
                        *reinterpret_cast<float4 *>(dst + 4) = b;
                    } else {
                        if (p.bias) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v.v[e] += bias8[e];
                        }
                        if (p.res) {
                            uint4 ra, rb;
                            if (it < PF) { ra = res_a[it < PF ? it : 0]; rb = res_b[it < PF ? it : 0]; }
                            else load_res(it, ra, rb);
                            if (p.res_fmt == 0) {
                                v.v[0] += __uint_as_float(ra.x); v.v[1] += __uint_as_float(ra.y);
                                v.v[2] += __uint_as_float(ra.z); v.v[3] += __uint_as_float(ra.w);
                                v.v[4] += __uint_as_float(rb.x); v.v[5] += __uint_as_float(rb.y);
                                v.v[6] += __uint_as_float(rb.z); v.v[7] += __uint_as_float(rb.w);
                            } else {                                  // SPLIT16 group: [8 x f16 hi][8 x f16 lo]
                                const half8 hi = __builtin_bit_cast(half8, ra), lo = __builtin_bit_cast(half8, rb);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v.v[e] += (float)hi[e] + (float)lo[e];
                            }
                        }
                        bool nan_pre = false;                        // fmaxf(NaN, 0) = 0: look before the ReLU launders an inf - inf
                        if (OUT_SPLIT) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) nan_pre = nan_pre || (v.v[e] != v.v[e]);
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(v.v[e], 0.f);
                        }
                        if (OUT_SPLIT && nan_pre) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                        if constexpr (MR * NR < 8) {
                            if (p.up_top) {
                                // FPN top-down addition (stereo_rcnn.py:91-108): the coarser level, bilinear with align_corners, added
                                // to this lateral.  upsample_add_kernel's arithmetic (pool_resize.hip) operation by operation -- this
                                // file is built without contraction too --, on the value the two-launch form would have stored as
                                // float32: the sum is bit-identical to srcnn_conv2d + srcnn_upsample_add.
                                const int TH = p.up_TH, TW = p.up_TW;
                                const float rh = p.OH > 1 ? (float)(TH - 1) / (float)(p.OH - 1) : 0.f;
                                const float rw = p.OW > 1 ? (float)(TW - 1) / (float)(p.OW - 1) : 0.f;
                                const int ohw = p.OH * p.OW;
                                const int bi = row / ohw, rem = row - bi * ohw;
                                const int h = rem / p.OW, w = rem - h * p.OW;
                                const float h1r = rh * (float)h;
                                const int h1 = (int)h1r;
                                const int h1p = (h1 < TH - 1) ? 1 : 0;
                                const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
                                const size_t trow0 = ((size_t)bi * TH + h1) * TW, trow1 = trow0 + (size_t)h1p * TW;
                                const float w1r = rw * (float)w;
                                const int w1 = (int)w1r;
                                const int w1p = (w1 < TW - 1) ? 1 : 0;
                                const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
                                const int gg = col >> 3;
                                const float8 ta = act_load8(p.up_top, p.up_fmt, trow0 + w1, p.Cout, gg);
                                const float8 tb = act_load8(p.up_top, p.up_fmt, trow0 + w1 + w1p, p.Cout, gg);
                                const float8 tc = act_load8(p.up_top, p.up_fmt, trow1 + w1, p.Cout, gg);
                                const float8 td = act_load8(p.up_top, p.up_fmt, trow1 + w1 + w1p, p.Cout, gg);
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    v.v[e] = (h0l * (w0l * ta.v[e] + w1l * tb.v[e]) + h1l * (w0l * tc.v[e] + w1l * td.v[e])) + v.v[e];
                            }
                        }
                        size_t opix = (size_t)row;
                        if (p.mode == 1) {                           // 2x2 stride-2 scatter: pixel (2*oh + i, 2*ow + j)
                            const int ohw = p.OH * p.OW;
                            const int bb = row / ohw, rem = row - bb * ohw;
                            const int oh = rem / p.OW, ow = rem - oh * p.OW;
                            opix = ((size_t)bb * 2 * p.OH + 2 * oh + (ij >> 1)) * (2 * p.OW) + 2 * ow + (ij & 1);
                        }
                        int eye_off = 0;
                        if (p.mode == 2) {                           // rows of the second half of the batch: same pixel, next Cout channels
                            const int half_rows = (p.nimg >> 1) * p.OH * p.OW;
                            if (row >= half_rows) {
                                opix = (size_t)(row - half_rows);
                                eye_off = p.Cout;
                            }
                        }
                        if (OUT_SPLIT) split16_guard(v, p.range_flag, p.tag);
                        if (p.nt_out) act_store8<true>(p.y, OUT_SPLIT ? 1 : 0, opix, p.ycs, (p.yco + eye_off + co) >> 3, v);
                        else act_store8(p.y, OUT_SPLIT ? 1 : 0, opix, p.ycs, (p.yco + eye_off + co) >> 3, v);
                    }
                }
            }
        };
        one_pass(std::integral_constant<int, 0>{});
        if constexpr (PASSES > 1) one_pass(std::integral_constant<int, 1>{});
        static_assert(PASSES <= 2, "epilogue passes");
        if (p.stamp && t == 0) {
            unsigned long long *o = p.stamp + 16 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[8] = rt0;
            o[10] = w_vm;
            o[11] = w_bar;
            o[9] = __builtin_amdgcn_s_memrealtime();
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4;
            o[5] = __builtin_readcyclecounter();
            o[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            o[7] = 1ULL | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 8);   // XCC_ID
        }
        return;
    }
    // General path (the 6-channel keypoint classifier, channel counts / offsets that are not multiples of 8).  The 256x256
    // tile is only ever planned for layers that take the fast path (conv_f16s_plan_ok): its 128 accumulators x this body
    // would not be unrolled, and accumulators indexed by a run-time loop live in scratch.
    if constexpr (MR * NR < 8) {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + li;
            if (col >= p.Cout) continue;
            const float bv = (!split && p.bias) ? p.bias[p.mode == 1 ? col % (p.Cout >> 2) : col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * MR + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
                if (row >= p.M) continue;
                float v = acc[i][j][e] * os;
                if (split) {
                    p.partial[((size_t)blockIdx.y * p.M + row) * p.Cout + col] = v;
                    continue;
                }
                v += bv;
                if (p.mode != 1) {
                    if (p.res) v += act_load(p.res, p.res_fmt, (size_t)row, p.rcs, col);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (OUT_SPLIT && !(fabsf(v) <= 65504.f)) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                    const int half_rows = (p.nimg >> 1) * p.OH * p.OW;
                    const bool second = p.mode == 2 && row >= half_rows;
                    act_store(p.y, OUT_SPLIT ? 1 : 0, (size_t)(second ? row - half_rows : row), p.ycs, p.yco + col + (second ? p.Cout : 0), v);
                } else {
                    const int cq = p.Cout >> 2;
                    const int ij = col / cq, co = col - ij * cq;
                    const int ohw = p.OH * p.OW;
                    const int b = row / ohw, rem = row - b * ohw;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    const size_t opix = ((size_t)b * 2 * p.OH + 2 * oh + (ij >> 1)) * (2 * p.OW) + 2 * ow + (ij & 1);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (OUT_SPLIT && !(fabsf(v) <= 65504.f)) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                    act_store(p.y, OUT_SPLIT ? 1 : 0, opix, p.ycs, p.yco + co, v);
                }
            }
        }
    }
    }
}

template <int MR, int NR, int WM, int NS>
static void launch(const ConvArgs &a, int splits, hipStream_t st)
{
    constexpr size_t lds = (size_t)NS * 128 * (32 * MR * WM + 64 * NR);   // NS stages of (hi+lo) x (BM+BN) rows x 64 B
    static bool configured = false;
    auto *k1 = conv_f16s_kernel<MR, NR, true, WM, NS>;
    auto *k0 = conv_f16s_kernel<MR, NR, false, WM, NS>;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    if constexpr (WM == 4 && NS == 2 && ((MR == 2 && NR == 4) || (MR == 1 && NR == 2))) {
        if (a.head_wf) {                       // MFMA-form narrow head: bias + W2 fragments of the tile's columns behind the tile
            auto *kh = conv_f16s_kernel<MR, NR, false, WM, NS, 2>;
            constexpr size_t BN_ = 64 * NR;
            const size_t lds2 = lds + BN_ * 4 + (BN_ / 16) * (size_t)a.head_rows * 64;
            static bool head2_configured = false;
            if (!head2_configured) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kh), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(lds + BN_ * 4 + (BN_ / 16) * 32 * 64 > 163840 ? 163840 : lds + BN_ * 4 + (BN_ / 16) * 32 * 64));
                head2_configured = true;
            }
            SRCNN_LAUNCH(kh, dim3(a.mtiles * a.ntiles, splits), dim3(128 * WM), lds2, st, a);
            return;
        }
    }
    if constexpr (MR == 2 && NR == 4 && WM == 4 && NS == 2) {
        if (a.head_w) {                        // fused 6-channel head instead of the y store
            auto *kh = conv_f16s_kernel<MR, NR, false, WM, NS, 1>;
            static bool head_configured = false;
            if (!head_configured) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                head_configured = true;
            }
            SRCNN_LAUNCH(kh, dim3(a.mtiles * a.ntiles, splits), dim3(128 * WM), lds, st, a);
            return;
        }
    }
    if (a.y_fmt == 1) SRCNN_LAUNCH(k1, dim3(a.mtiles * a.ntiles, splits), dim3(128 * WM), lds, st, a);
    else SRCNN_LAUNCH(k0, dim3(a.mtiles * a.ntiles, splits), dim3(128 * WM), lds, st, a);
}

// Plan -> instantiation.  Workgroup tile (64*mr) x (64*nr); waves 4 or 8; stages = LDS ring depth.
bool conv_f16s_plan_ok(const Plan &pl, const ConvArgs &a)
{
    const int t = pl.mr * 100 + pl.nr * 10 + (pl.waves == 8 ? 8 : 4);
    switch (t) {
    case 114: return pl.stages == 2 || pl.stages == 4;
    case 214: case 124: return pl.stages == 2 || pl.stages == 3;
    case 224: return pl.stages == 2;
    case 228: return pl.stages == 2 || pl.stages == 4;
    case 428: return pl.stages == 3;
    case 448: {                               // 256x256 on 8 waves of 64x128 (two waves per SIMD, 256 registers each):
        const int cq = a.mode == 1 ? (a.Cout >> 2) : a.Cout;      // vector epilogue only (see the kernel's general path)
        return pl.stages == 2 && !a.x2 && !a.up_top && (cq & 7) == 0 && (a.ycs & 7) == 0 && (a.yco & 7) == 0 && (!a.res || (a.rcs & 7) == 0);
    }
    default: return false;
    }
}

void launch_conv_f16s(const ConvArgs &a, const Plan &pl, hipStream_t st)
{
    const int t = pl.mr * 1000 + pl.nr * 100 + (pl.waves == 8 ? 80 : 40) + pl.stages;
    switch (t) {
    case 1142: launch<1, 1, 2, 2>(a, pl.splits, st); break;
    case 1144: launch<1, 1, 2, 4>(a, pl.splits, st); break;
    case 2142: launch<2, 1, 2, 2>(a, pl.splits, st); break;
    case 2143: launch<2, 1, 2, 3>(a, pl.splits, st); break;
    case 1242: launch<1, 2, 2, 2>(a, pl.splits, st); break;
    case 1243: launch<1, 2, 2, 3>(a, pl.splits, st); break;
    case 2242: launch<2, 2, 2, 2>(a, pl.splits, st); break;
    case 2282: launch<1, 2, 4, 2>(a, pl.splits, st); break;   // 128x128 on 8 waves of 32x64
    case 2284: launch<1, 2, 4, 4>(a, pl.splits, st); break;
    case 4283: launch<2, 2, 4, 3>(a, pl.splits, st); break;   // 256x128 on 8 waves of 64x64
    case 4482: launch<2, 4, 4, 2>(a, pl.splits, st); break;   // 256x256 on 8 waves of 64x128
    default: launch<1, 1, 2, 2>(a, pl.splits, st); break;     // unreachable: plan_for() validates with conv_f16s_plan_ok
    }
}

}  // namespace srcnn
