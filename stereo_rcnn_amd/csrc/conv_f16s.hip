// Convolution engine, 3xf16 split form with BOTH operands DMA'd straight into LDS: the single-launch kernel (one output tile
// per workgroup; grid.y = split-K slices) and its plan -> instantiation table.  The tile itself -- DMA ring, hand-pipelined K
// loop, epilogues -- lives in conv_f16s_tile.h, shared with the chained / grouped launches of conv_chain.hip.
#include "conv_f16s_tile.h"

namespace srcnn {

struct TileCtxLaunch {           // one tile per workgroup; grid.y = split-K slices
    int mt, nt, m_rows, kt_begin, kt_end;
    __device__ __forceinline__ int thread() const { return threadIdx.x; }
    __device__ __forceinline__ int split_idx() const { return gridDim.y > 1 ? (int)blockIdx.y : -1; }
    __device__ __forceinline__ bool stamping(const ConvArgs &p) const { return p.stamp != nullptr; }
    __device__ __forceinline__ unsigned long long *stamp_slot(const ConvArgs &p) const
    {
        return p.stamp + 16 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
    }
};

template <int MR, int NR, bool OUT_SPLIT, int WM, int NS, int HEAD = 0>
__global__ __launch_bounds__(128 * WM, (MR * NR >= 16 ? 1 : 2)) void conv_f16s_kernel(const ConvArgs p)
{
    constexpr int BM = 32 * MR * WM;
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
    TileCtxLaunch c;
    c.mt = p.m_fast ? logical % mtiles : logical / p.ntiles;
    c.nt = p.m_fast ? logical / mtiles : logical - c.mt * p.ntiles;
    c.m_rows = m_rows;
    c.kt_begin = blockIdx.y * p.kt_per_split;
    c.kt_end = min(p.nkt, c.kt_begin + p.kt_per_split);
    {
        const TileCtxLaunch ctx = c;
#include "conv_f16s_body.inc"
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
    case 424: case 244:                       // 256x128 / 128x256 on 4 waves of 128x64 / 64x128 (explicit plans and the throughput tuner only)
        if (pl.stages != 2 && pl.stages != 3) return false;
        [[fallthrough]];
    case 444:                                 // 256x256 on 4 waves of 128x128 (one wave per SIMD, 256 accumulator registers): a
                                              // third fewer LDS fragment reads per MFMA than the 8-wave form (16 per 48 instead of 12 per 24)
    case 448: {                               // 256x256 on 8 waves of 64x128 (two waves per SIMD, 256 registers each):
        const int cq = a.mode == 1 ? (a.Cout >> 2) : a.Cout;      // vector epilogue only (see the kernel's general path)
        if (t != 448 && (a.head_w || a.head_wf)) return false;   // the fused heads live in the 8-wave form
        return (pl.stages == 2 || t == 424 || t == 244) && !a.x2 && !a.up_top && (cq & 7) == 0 && (a.ycs & 7) == 0 && (a.yco & 7) == 0 && (!a.res || (a.rcs & 7) == 0);
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
    case 4442: launch<4, 4, 2, 2>(a, pl.splits, st); break;   // 256x256 on 4 waves of 128x128
    case 4242: launch<4, 2, 2, 2>(a, pl.splits, st); break;   // 256x128 on 4 waves of 128x64
    case 4243: launch<4, 2, 2, 3>(a, pl.splits, st); break;
    case 2442: launch<2, 4, 2, 2>(a, pl.splits, st); break;   // 128x256 on 4 waves of 64x128
    case 2443: launch<2, 4, 2, 3>(a, pl.splits, st); break;
    default: launch<1, 1, 2, 2>(a, pl.splits, st); break;     // unreachable: plan_for() validates with conv_f16s_plan_ok
    }
}

}  // namespace srcnn
