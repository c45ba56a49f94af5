// Chained and grouped launches of the SPLIT16 convolution engine: more than one convolution per kernel launch.
//
// CHAIN (srcnn_conv2d_chain): up to three convolutions over the SAME output rows, each reading what the one before it wrote:
//   a workgroup owns one M tile (BM consecutive output pixels) and walks every N tile of phase 0, then of phase 1, then of
//   phase 2.  Phases after the first must be 1x1 / stride 1 / pad 0 convolutions of the previous phase's output: their input
//   rows are exactly the rows this workgroup has just written, so the only synchronisation is workgroup-local -- the stores
//   of a phase are complete (`s_waitcnt vmcnt(0)`: acknowledged by the XCD's L2, which every CU of the XCD reads through)
//   before the barrier that opens the next phase, and no other workgroup ever reads those rows inside the launch.  The
//   intermediate tensors still exist in memory (they are written once), but they are read back from the L2 while the lines
//   are hot, and two of three launch boundaries -- drain, ramp, cold prologue -- are gone.  The ResNet bottleneck
//   (/root/reference/lib/model/stereo_rcnn/resnet.py:82-102) maps onto it SHIFTED BY ONE CONVOLUTION:
//       [conv2 (3x3) -> conv3 (+ residual / projection shortcut) -> conv1 of the NEXT block]
//   so that the one convolution with a spatial footprint (the 3x3, whose halo rows belong to other workgroups) is always the
//   FIRST phase and reads a tensor completed by the previous launch.  The caller double-buffers the tensor the last phase
//   writes when the first phase reads its predecessor (a fast workgroup's phase 2 would otherwise overwrite halo rows a
//   slow neighbour's phase 0 still needs).
//   Arithmetic: every phase is the tile code of conv_f16s_body.inc, unsplit -- bit-identical to the same convolutions
//   launched one by one with the same tiles (same products, same order).
//
// GROUP (srcnn_conv2d_group): up to five independent convolutions that share one tile configuration (and usually their
//   weights), one launch: the logical tile index runs over the tiles of all problems.  The stereo RPN
//   (/root/reference/lib/model/rpn/stereo_rpn.py:73-95) applies RPN_Conv + heads to five pyramid levels with shared weights;
//   P4-P6 have 76, 20 and 6 tiles of work -- launches that cannot fill 256 CUs on their own.
#include "conv_f16s_tile.h"

namespace srcnn {

constexpr int CHAIN_MAX = 3, GROUP_MAX = 5;

struct ChainArgs {
    ConvArgs a[CHAIN_MAX];
    int n;                       // phases
    int mtiles;                  // M tiles = workgroups
    int step_end[CHAIN_MAX];     // running count of N tiles: phase i owns steps [step_end[i-1], step_end[i])
    int wide[CHAIN_MAX];         // phase i runs on the kernel's second N tile width
};

struct GroupArgs {
    ConvArgs a[GROUP_MAX];
    int n;
    int tile_end[GROUP_MAX];     // running tile count: problem i owns logical tiles [tile_end[i-1], tile_end[i])
};

// between two tiles of one workgroup: every wave has finished reading the epilogue's LDS tile before the next prologue's DMA
// lands in the ring (the reads' results have been consumed by the stores that follow them; the wait is formal)
__device__ __forceinline__ void tile_fence() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// between two phases: additionally, all of this wave's stores have been acknowledged by the L2
__device__ __forceinline__ void phase_fence() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int xcd_chunked(int bid, int nblk)
{
    // consecutive logical tiles are one XCD's share (they share halo rows / weights in that XCD's L2)
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// MR_, WM_, NS_: common to the phases (BM = 32 * MR_ * WM_ rows per workgroup); NRA / NRB: the two N tile widths (64 * NR columns)
// a phase can choose between (ChainArgs.wide[phase]: 0 = NRA, 1 = NRB) -- the narrow convolutions of a bottleneck (P outputs) and
// its wide one (4P outputs).
// ONE flat loop over the (phase, N tile) steps, the phase's arguments addressed by the step: written as nested loops the
// compiler hoists every kernel-argument load of a phase out of its N-tile loop -- 3 x 84 dwords live across the loop bodies,
// 300 spilled SGPRs and scratch.
// The kernel reads its arguments through the kernarg segment pointer (constant address space, scalar loads with a dynamic
// offset), never through the by-value parameter: indexing the parameter with a run-time phase makes clang keep a private copy of
// the whole struct -- 1 KB of scratch per lane and every field in a VGPR.
#define SRCNN_AS4 __attribute__((address_space(4)))

template <int MR_, int WM_, int NS_, int NRA, int NRB>
__global__ __launch_bounds__(128 * WM_, (MR_ * (NRA > NRB ? NRA : NRB) >= 16 ? 1 : 2)) void conv_chain_kernel(const ChainArgs)
{
    constexpr int MR = MR_, WM = WM_, NS = NS_, HEAD = 0;
    constexpr bool OUT_SPLIT = true;
    const SRCNN_AS4 ChainArgs &ch = *(const SRCNN_AS4 ChainArgs *)(unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const int mt_ = xcd_chunked(blockIdx.x, ch.mtiles);
    const int steps = ch.step_end[ch.n - 1];
#pragma clang loop unroll(disable)
    for (int s = 0; s < steps; ++s) {
        int ph = 0;
        while (ph + 1 < ch.n && s >= ch.step_end[ph]) ++ph;          // (uniform)
        const int nt_ = s - (ph ? ch.step_end[ph - 1] : 0);
        const SRCNN_AS4 ConvArgs &p = ch.a[ph];
        if (s > 0) {
            if (nt_ == 0) phase_fence();
            else tile_fence();
        }
        TileCtxPlain ctx;
        ctx.mt = mt_; ctx.nt = nt_; ctx.m_rows = p.M; ctx.kt_begin = 0; ctx.kt_end = p.nkt;
        if (NRA == NRB || !ch.wide[ph]) {
            constexpr int NR = NRA;
#include "conv_f16s_body.inc"
        } else {
            constexpr int NR = NRB;
#include "conv_f16s_body.inc"
        }
    }
}

template <int MR_, int NR_, bool OUT_SPLIT_, int WM_, int NS_, int HEAD_>
__global__ __launch_bounds__(128 * WM_, (MR_ * NR_ >= 16 ? 1 : 2)) void conv_group_kernel(const GroupArgs)
{
    constexpr int MR = MR_, NR = NR_, WM = WM_, NS = NS_, HEAD = HEAD_;
    constexpr bool OUT_SPLIT = OUT_SPLIT_;
    const SRCNN_AS4 GroupArgs &g = *(const SRCNN_AS4 GroupArgs *)(unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const int logical = xcd_chunked(blockIdx.x, g.tile_end[g.n - 1]);
    int i = 0;
    while (i + 1 < g.n && logical >= g.tile_end[i]) ++i;               // (uniform)
    const SRCNN_AS4 ConvArgs &p = g.a[i];
    const int local = logical - (i ? g.tile_end[i - 1] : 0);
    TileCtxPlain ctx;
    ctx.mt = local / p.ntiles; ctx.nt = local - ctx.mt * p.ntiles; ctx.m_rows = p.M; ctx.kt_begin = 0; ctx.kt_end = p.nkt;
    {
#include "conv_f16s_body.inc"
    }
}

// ---------------------------------------------------------------- host side
static int chain_check(const srcnn_conv_desc *d, int n, const ConvArgs *a)
{
    for (int i = 0; i < n; ++i) {
        SRCNN_REQUIRE(d[i].precision == 1 && d[i].x_format == 1 && d[i].y_format == 1 && d[i].y, "chain: SPLIT16 f16x3 engine, SPLIT16 in and out");
        SRCNN_REQUIRE(d[i].mode == 0 && !d[i].m_limit && !d[i].head_w && !d[i].head_wf && !d[i].up_top,
                      "chain: plain convolutions only (mode 0, no row limit / fused head / top-down addition)");
        SRCNN_REQUIRE((a[i].Cout & 7) == 0 && (a[i].ycs & 7) == 0 && (a[i].yco & 7) == 0 && (!a[i].res || (a[i].rcs & 7) == 0),
                      "chain: channel counts / strides multiples of 8 (vector epilogue)");
        SRCNN_REQUIRE(a[i].M == a[0].M, "chain: every phase has the same output rows");
        if (i > 0) {
            SRCNN_REQUIRE(d[i].KH == 1 && d[i].KW == 1 && d[i].stride == 1 && d[i].pad == 0 && d[i].H == d[i].OH && d[i].W == d[i].OW,
                          "chain: phases after the first are 1x1 / stride 1 convolutions (their input rows = the rows the workgroup wrote)");
            SRCNN_REQUIRE(static_cast<const void *>(d[i].x) == static_cast<const void *>(d[i - 1].y) && d[i].x_cstride == d[i - 1].y_cstride &&
                              d[i - 1].y_coffset == 0 && d[i].Cin == d[i - 1].Cout,
                          "chain: phase i reads exactly what phase i-1 wrote");
        }
        SRCNN_REQUIRE(static_cast<const void *>(d[i].y) != static_cast<const void *>(d[0].x),
                      "chain: no phase may overwrite the tensor the first phase reads (other workgroups still need its halo rows)");
        for (int j = 0; j < n; ++j)
            SRCNN_REQUIRE(static_cast<const void *>(d[i].y) != static_cast<const void *>(d[j].residual) &&
                              static_cast<const void *>(d[i].y) != d[j].x2,
                          "chain: residuals and second inputs must be tensors no phase of the chain writes");
    }
    return SRCNN_OK;
}

template <int MR, int WM, int NS, int NRA, int NRB>
static void launch_chain(const ChainArgs &c, hipStream_t st)
{
    constexpr int NRM = NRA > NRB ? NRA : NRB;
    constexpr size_t lds = (size_t)NS * 128 * (32 * MR * WM + 64 * NRM);
    auto *k = conv_chain_kernel<MR, WM, NS, NRA, NRB>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = true;
    }
    SRCNN_LAUNCH(k, dim3(c.mtiles), dim3(128 * WM), lds, st, c);
}

// the instantiated chains: key = (tile_mr, waves, stages, narrow nr, wide nr) with tile_mr = BM / 64 as in srcnn_conv_desc;
// every phase's tile_nr must be one of the two widths
static bool dispatch_chain(int mr, int waves, int stages, int na, int nb, const ChainArgs *c, hipStream_t st)
{
    const long key = (((long)mr * 10 + waves) * 10 + stages) * 100 + na * 10 + nb;
#define SRCNN_CHAIN(KEY, MR, WM, NS, A, B) \
    case KEY: if (c) launch_chain<MR, WM, NS, A, B>(*c, st); return true;
    switch (key) {
    // 128 rows, 4 waves of 64x32 / 64x64 (P = 64: layer1)
    SRCNN_CHAIN(24212, 2, 2, 2, 1, 2)
    SRCNN_CHAIN(24211, 2, 2, 2, 1, 1)
    SRCNN_CHAIN(24222, 2, 2, 2, 2, 2)
    // 128 rows, 8 waves of 32x64
    SRCNN_CHAIN(28222, 1, 4, 2, 2, 2)
    SRCNN_CHAIN(28422, 1, 4, 4, 2, 2)
    // 256 rows, 8 waves of 64x64 / 64x128
    SRCNN_CHAIN(48322, 2, 4, 3, 2, 2)
    SRCNN_CHAIN(48244, 2, 4, 2, 4, 4)
    default: return false;
    }
#undef SRCNN_CHAIN
}

template <int MR, int NR, int WM, int NS>
static void launch_group(const GroupArgs &g, bool out_split, hipStream_t st)
{
    const size_t lds0 = (size_t)NS * 128 * (32 * MR * WM + 64 * NR);
    const int blocks = g.tile_end[g.n - 1];
    if constexpr (WM == 4 && NS == 2 && ((MR == 2 && NR == 4) || (MR == 1 && NR == 2))) {
        if (g.a[0].head_wf) {
            auto *kh = conv_group_kernel<MR, NR, false, WM, NS, 2>;
            constexpr size_t BN_ = 64 * NR;
            const size_t lds2 = lds0 + BN_ * 4 + (BN_ / 16) * (size_t)g.a[0].head_rows * 64;
            static bool hc = false;
            if (!hc) {
                const size_t mx = lds0 + BN_ * 4 + (BN_ / 16) * 32 * 64;
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(mx > 163840 ? 163840 : mx));
                hc = true;
            }
            SRCNN_LAUNCH(kh, dim3(blocks), dim3(128 * WM), lds2, st, g);
            return;
        }
    }
    auto *k1 = conv_group_kernel<MR, NR, true, WM, NS, 0>;
    auto *k0 = conv_group_kernel<MR, NR, false, WM, NS, 0>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
        configured = true;
    }
    if (out_split) SRCNN_LAUNCH(k1, dim3(blocks), dim3(128 * WM), lds0, st, g);
    else SRCNN_LAUNCH(k0, dim3(blocks), dim3(128 * WM), lds0, st, g);
}

}  // namespace srcnn

extern "C" {

int srcnn_conv2d_chain_supported(int tile_mr, int tile_waves, int tile_stages, int nr_narrow, int nr_wide)
{
    return srcnn::dispatch_chain(tile_mr, tile_waves, tile_stages, nr_narrow, nr_wide, nullptr, nullptr) ? 1 : 0;
}

int srcnn_conv2d_chain(const srcnn_conv_desc *descs, int n, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(descs && n >= 1 && n <= CHAIN_MAX, "1 to 3 convolutions");
    ChainArgs c;
    for (int i = 0; i < n; ++i) {
        const int rc = conv_fill_args(&descs[i], c.a[i]);
        if (rc != SRCNN_OK) return rc;
    }
    const int rc = chain_check(descs, n, c.a);
    if (rc != SRCNN_OK) return rc;
    const int mr = descs[0].tile_mr, waves = descs[0].tile_waves > 0 ? descs[0].tile_waves : 4;
    const int stages = descs[0].tile_stages > 0 ? descs[0].tile_stages : 2;
    int na = descs[0].tile_nr, nb = descs[0].tile_nr;
    for (int i = 0; i < n; ++i) {
        SRCNN_REQUIRE(descs[i].tile_mr == mr && (descs[i].tile_waves > 0 ? descs[i].tile_waves : 4) == waves &&
                          (descs[i].tile_stages > 0 ? descs[i].tile_stages : 2) == stages,
                      "chain: the phases share tile_mr / tile_waves / tile_stages");
        na = min(na, descs[i].tile_nr);
        nb = max(nb, descs[i].tile_nr);
    }
    for (int i = 0; i < n; ++i)
        SRCNN_REQUIRE(descs[i].tile_nr == na || descs[i].tile_nr == nb, "chain: at most two N tile widths");
    SRCNN_REQUIRE(dispatch_chain(mr, waves, stages, na, nb, nullptr, nullptr),
                  "chain: no such tile configuration (srcnn_conv2d_chain_supported)");
    c.n = n;
    c.mtiles = cdiv(c.a[0].M, 64 * mr);
    int steps = 0;
    for (int i = 0; i < CHAIN_MAX; ++i) {
        if (i < n) {
            // (the 256x256 tile has no second-input path: register budget)
            SRCNN_REQUIRE(!(c.a[i].x2 && mr == 4 && descs[i].tile_nr == 4), "chain: the 256x256 tile takes no second input");
            c.a[i].mtiles = c.mtiles;
            c.a[i].ntiles = cdiv(c.a[i].Cout, 64 * descs[i].tile_nr);
            c.a[i].kt_per_split = c.a[i].nkt;
            c.a[i].stamp = nullptr;
            c.a[i].partial = nullptr;
            c.wide[i] = (na != nb && descs[i].tile_nr == nb) ? 1 : 0;
            steps += c.a[i].ntiles;
        } else {
            c.a[i] = c.a[n - 1];       // never read
            c.wide[i] = 0;
        }
        c.step_end[i] = steps;
    }
    hipStream_t st = as_stream(stream);
    const bool prof = prof_enabled();
    if (prof) prof_begin(st);
    dispatch_chain(mr, waves, stages, na, nb, &c, st);
    if (prof) {
        double fl = 0;
        for (int i = 0; i < n; ++i) fl += 2.0 * (double)c.a[i].M * (double)c.a[i].Cout * (double)c.a[i].K;
        prof_end(st, fl);
    }
    return check_launch("srcnn_conv2d_chain");
}

int srcnn_conv2d_group(const srcnn_conv_desc *descs, int n, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(descs && n >= 1 && n <= GROUP_MAX, "1 to 5 convolutions");
    GroupArgs g;
    const int mr = descs[0].tile_mr, nr = descs[0].tile_nr, waves = descs[0].tile_waves > 0 ? descs[0].tile_waves : 4;
    const int stages = descs[0].tile_stages > 0 ? descs[0].tile_stages : 2;
    int total = 0;
    double fl = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = conv_fill_args(&descs[i], g.a[i]);
        if (rc != SRCNN_OK) return rc;
        SRCNN_REQUIRE(descs[i].precision == 1 && descs[i].x_format == 1 && !descs[i].m_limit && !descs[i].head_w,
                      "group: SPLIT16 f16x3 engine, no row limit, no fp32-FMA head");
        SRCNN_REQUIRE(descs[i].tile_mr == mr && descs[i].tile_nr == nr && (descs[i].tile_waves > 0 ? descs[i].tile_waves : 4) == waves &&
                          (descs[i].tile_stages > 0 ? descs[i].tile_stages : 2) == stages,
                      "group: one tile configuration for all problems");
        SRCNN_REQUIRE(descs[i].y_format == descs[0].y_format && (descs[i].head_wf != nullptr) == (descs[0].head_wf != nullptr) &&
                          descs[i].head_rows == descs[0].head_rows,
                      "group: one output format / head form for all problems");
        Plan pl;
        pl.mr = mr; pl.nr = nr; pl.waves = waves; pl.stages = stages; pl.splits = 1; pl.kt_per_split = g.a[i].nkt;
        SRCNN_REQUIRE(conv_f16s_plan_ok(pl, g.a[i]), "group: no such tile configuration for this convolution");
        if (g.a[i].head_wf) SRCNN_REQUIRE(waves == 8 && stages == 2 && ((mr == 4 && nr == 4) || (mr == 2 && nr == 2)),
                                          "group: the MFMA-form head lives in the 256x256 and 128x128 8-wave tiles");
        g.a[i].mtiles = cdiv(g.a[i].M, 64 * mr);
        g.a[i].ntiles = cdiv(g.a[i].Cout, 64 * nr);
        g.a[i].kt_per_split = g.a[i].nkt;
        g.a[i].stamp = nullptr;
        g.a[i].partial = nullptr;
        g.a[i].m_fast = 0;
        total += g.a[i].mtiles * g.a[i].ntiles;
        g.tile_end[i] = total;
        fl += 2.0 * (double)g.a[i].M * (double)g.a[i].Cout * (double)g.a[i].K;
    }
    for (int i = n; i < GROUP_MAX; ++i) g.tile_end[i] = total;
    g.n = n;
    hipStream_t st = as_stream(stream);
    const bool prof = prof_enabled();
    if (prof) prof_begin(st);
    const bool out_split = descs[0].y_format == 1;
    const int t = mr * 1000 + nr * 100 + (waves == 8 ? 80 : 40) + stages;
    switch (t) {
    case 2282: launch_group<1, 2, 4, 2>(g, out_split, st); break;     // 128x128 on 8 waves
    case 4283: launch_group<2, 2, 4, 3>(g, out_split, st); break;     // 256x128
    case 4482: launch_group<2, 4, 4, 2>(g, out_split, st); break;     // 256x256
    case 2142: launch_group<2, 1, 2, 2>(g, out_split, st); break;     // 128x64 on 4 waves
    default:
        set_error("srcnn_conv2d_group: tile configuration not instantiated for grouped launches");
        return SRCNN_ERR_ARG;
    }
    if (prof) prof_end(st, fl);
    return check_launch("srcnn_conv2d_group");
}

}  // extern "C"
