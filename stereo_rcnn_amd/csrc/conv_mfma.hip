// Convolution engine: implicit GEMM on the gfx950 fp32 matrix core.
//
// Replaces the cuDNN convolutions behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear on the
// reference's hot path (stereo_rcnn/resnet.py:66-146,243-286; rpn/stereo_rpn.py:32-40).
//
//   y[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + residual[m, n] )
//   m = (b, oh, ow)   k = (kh, kw, c)   n = cout          NHWC activations, W = [Cout][KH][KW][Cin]
//
// Design (MI355X-first, not a cuDNN/CUTLASS shape):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 157 TF peak.  It is paced at
//     64 cycles/instruction, so LDS/HBM pressure per flop is 16x lower than a bf16 GEMM: a 2x2
//     wave grid with (32*MR)x(32*NR) wave tiles saturates the pipe without deep pipelining.
//   * K order inside a 32-wide K tile is permuted so that ONE ds_read_b128 feeds FOUR MFMAs:
//     lane (i, g) reads k = kk*8 + g*4 .. +3 of row i; MFMA s pairs k=kk*8+s (g=0) with
//     k=kk*8+4+s (g=1) on both operands.  The sum over k is order-independent in exact
//     arithmetic; in fp32 it is one fixed, deterministic order.
//   * LDS rows are 32 floats + 4 pad (144 B): the 16-lane groups of ds_read_b128 then touch 16
//     distinct 4-bank slots -> conflict-free; ds_write_b128 writes one row per 8 lanes.
//   * im2col is never materialised: a K tile is 32 contiguous channels of one (kh, kw) tap,
//     i.e. one 128-B run per output pixel, fetched as 8 lanes x 16 B (coalesced), zero-filled
//     outside the image.  Global->register prefetch of tile t+1 overlaps the MFMAs of tile t;
//     LDS is double-buffered -> one barrier per K tile.
//   * 256 CUs / 8 XCDs: tile ids are remapped so that consecutive logical tiles (which share the
//     activation rows) run on the same XCD and hit its L2; small-M layers use split-K so that
//     the grid still covers the chip (partials in the caller's workspace, deterministic reduce).
//   * epilogue fuses folded-BN bias, residual add, ReLU, channel-offset writes (concat in place)
//     and the ConvTranspose2d(2,2) pixel scatter.
#include "conv_common.h"
#include <cstdlib>

namespace srcnn {

template <int MR, int NR>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs p)
{
    constexpr int BM = 64 * MR, BN = 64 * NR;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;   // float4 loads per thread per tile
    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDS_ROW];

    const int t = threadIdx.x;
    // ---- XCD-aware tile mapping (bijective; blocks b -> XCD b%8)
    const int nblk = p.mtiles * p.ntiles;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_begin = blockIdx.y * p.kt_per_split;
    const int kt_end = min(p.nkt, kt_begin + p.kt_per_split);

    // ---- per-thread staging geometry
    const int lrow = t >> 3;          // 0..31
    const int lcol = (t & 7) * 4;     // float offset inside the 32-float run
    int a_ih0[A_LD], a_iw0[A_LD], a_pix[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_pix[i] = (b * p.H + a_ih0[i]) * p.W + a_iw0[i];
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = 0;
            a_pix[i] = 0;
        }
    }
    const float *b_ptr[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_ok[i] = n < p.Cout;
        b_ptr[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + lcol;
    }

    float4 ra[A_LD], rb[B_LD];
    auto load_tile = [&](int kt) {
        const int tap = kt / p.ctiles;
        const int c0 = (kt - tap * p.ctiles) * BK;
        const int kh = tap / p.KW;
        const int kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const int pix = ok ? a_pix[i] + kh * p.W + kw : 0;
            const float4 v = *reinterpret_cast<const float4 *>(p.x + (size_t)pix * p.xcs + c0 + lcol);
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(b_ptr[i] + (size_t)kt * BK);
            rb[i] = b_ok[i] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float *sa = smem[buf];
        float *sb = smem[buf] + BM * LDS_ROW;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<float4 *>(sa + (lrow + 32 * i) * LDS_ROW + lcol) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<float4 *>(sb + (lrow + 32 * i) * LDS_ROW + lcol) = rb[i];
    };

    // ---- wave geometry
    const int wave = t >> 6, lane = t & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lg = lane >> 5;
    floatx16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) load_tile(kt + 1);
        const float *sa = smem[buf] + (wm * 32 * MR + li) * LDS_ROW + lg * 4;
        const float *sb = smem[buf] + BM * LDS_ROW + (wn * 32 * NR + li) * LDS_ROW + lg * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 fa[MR], fb[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) fa[i] = *reinterpret_cast<const float4 *>(sa + i * 32 * LDS_ROW + kk * 8);
#pragma unroll
            for (int j = 0; j < NR; ++j) fb[j] = *reinterpret_cast<const float4 *>(sb + j * 32 * LDS_ROW + kk * 8);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const bool split = gridDim.y > 1;
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
                float v = acc[i][j][e];
                if (split) {
                    p.partial[((size_t)blockIdx.y * p.M + row) * p.Cout + col] = v;
                    continue;
                }
                v += bv;
                if (p.mode == 0) {
                    if (p.res) v += p.res[(size_t)row * p.rcs + col];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[(size_t)row * p.ycs + p.yco + col] = v;
                } else {   // ConvTranspose2d(k=2, s=2): col = (i2*2 + j2)*Cq + co
                    const int cq = p.Cout >> 2;
                    const int ij = col / cq, co = col - ij * cq;
                    const int ohw = p.OH * p.OW;
                    const int b = row / ohw, rem = row - b * ohw;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    const size_t opix = ((size_t)b * 2 * p.OH + 2 * oh + (ij >> 1)) * (2 * p.OW) + 2 * ow + (ij & 1);
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[opix * p.ycs + p.yco + co] = v;
                }
            }
        }
    }
}

// deterministic split-K reduction + epilogue (modes 0 and 2)
__global__ void splitk_reduce_kernel(const ConvArgs p, int splits)
{
    // rows beyond the device-side limit were not computed by the conv launch: leave them alone (their partials are stale)
    const size_t row_limit = p.m_limit ? (size_t)max(*p.m_limit, 0) * (size_t)p.m_limit_mul : (size_t)p.M;
    const size_t total = (size_t)p.M * p.Cout;
    if ((p.Cout & 7) == 0 && (p.ycs & 7) == 0 && (p.yco & 7) == 0 && (!p.res || (p.rcs & 7) == 0)) {
        // 8 channels per thread: 32-byte reads of every slab, vector residual / store in either format
        const int G = p.Cout >> 3;
        const size_t groups = (size_t)p.M * G;
        for (size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups;
             gi += (size_t)gridDim.x * blockDim.x) {
            const size_t row = gi / G;
            const int g = (int)(gi - row * G);
            if (row >= row_limit) continue;
            float8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = 0.f;
            // four slices' loads in flight per step (independent 32-byte reads; a one-slice-per-iteration loop serialises one memory
            // round trip per slice), added in slice order: the sum is the same as before, bit for bit
            for (int s0 = 0; s0 < splits; s0 += 4) {
                float4 a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    b[j] = a[j];
                    if (s0 + j < splits) {
                        const float *src = p.partial + (size_t)(s0 + j) * total + row * p.Cout + g * 8;
                        a[j] = *reinterpret_cast<const float4 *>(src);
                        b[j] = *reinterpret_cast<const float4 *>(src + 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (s0 + j < splits) {
                        v.v[0] += a[j].x; v.v[1] += a[j].y; v.v[2] += a[j].z; v.v[3] += a[j].w;
                        v.v[4] += b[j].x; v.v[5] += b[j].y; v.v[6] += b[j].z; v.v[7] += b[j].w;
                    }
                }
            }
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += p.bias[g * 8 + e];
            }
            if (p.res) {
                const float8 rr = act_load8(p.res, p.res_fmt, row, p.rcs, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += rr.v[e];
            }
            bool nan_pre = false;                    // fmaxf(NaN, 0) = 0: look before the ReLU launders an inf - inf
            if (p.y_fmt == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) nan_pre = nan_pre || (v.v[e] != v.v[e]);
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(v.v[e], 0.f);
            }
            if (p.y_fmt == 1) {                      // SPLIT16 result: same range guard as the conv kernels' epilogues
                if (nan_pre) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
                split16_guard(v, p.range_flag, p.tag);
            }
            const size_t half_rows = (size_t)(p.nimg >> 1) * p.OH * p.OW;
            const bool second = p.mode == 2 && row >= half_rows;      // see srcnn_conv_desc.mode
            act_store8(p.y, p.y_fmt, second ? row - half_rows : row, p.ycs, ((p.yco + (second ? p.Cout : 0)) >> 3) + g, v);
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / p.Cout), col = (int)(idx - (size_t)row * p.Cout);
        if ((size_t)row >= row_limit) continue;
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += p.partial[(size_t)s * total + idx];
        if (p.bias) v += p.bias[col];
        if (p.res) v += act_load(p.res, p.res_fmt, (size_t)row, p.rcs, col);
        const bool nan_pre = v != v;                                  // before the ReLU launders it
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.y_fmt == 1 && (nan_pre || !(fabsf(v) <= 65504.f))) atomicMax(p.range_flag, (unsigned)(p.tag + 1));
        const int half_rows = (p.nimg >> 1) * p.OH * p.OW;
        const bool second = p.mode == 2 && row >= half_rows;
        act_store(p.y, p.y_fmt, (size_t)(second ? row - half_rows : row), p.ycs, p.yco + col + (second ? p.Cout : 0), v);
    }
}

static Plan make_plan(int M, int N, int nkt, int mode, int precision)
{
    (void)precision;
    static const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
    Plan pl{1, 1, 1, nkt};
    const long target = 512;   // >= 2 workgroups per CU
    bool found = false;
    for (auto &c : cand) {
        if (c[1] == 2 && N <= 64) continue;
        long blocks = (long)cdiv(M, 64 * c[0]) * cdiv(N, 64 * c[1]);
        if (blocks >= target) {
            pl.mr = c[0];
            pl.nr = c[1];
            found = true;
            break;
        }
    }
    if (!found) {
        pl.mr = 1;
        pl.nr = 1;
        long blocks = (long)cdiv(M, 64) * cdiv(N, 64);
        if (mode != 1 && blocks < 256 && nkt >= 16) {
            int s = (int)((target + blocks - 1) / blocks);
            s = min(s, nkt / 8);   // keep >= 8 K tiles per slice
            s = min(s, 32);
            if (s > 1) {
                pl.kt_per_split = cdiv(nkt, s);
                pl.splits = cdiv(nkt, pl.kt_per_split);
            }
        }
    }
    return pl;
}

static Plan plan_for(const srcnn_conv_desc *d, const ConvArgs &a)
{
    Plan pl = make_plan(a.M, a.Cout, a.nkt, a.mode, d->precision);
    if (a.x2 || a.up_top) {        // the K walk of a second input has no mid-K entry points: never split; the fused top-down
        pl.splits = 1;             // addition lives in the conv kernel's epilogue, not in the split-K reduction
        pl.kt_per_split = a.nkt;
    }
    if (a.head_w) {                // the fused head exists for the 256x256 tile only (a workgroup owns all 256 channels of its pixels)
        Plan h = pl;
        h.mr = 4; h.nr = 4; h.waves = 8; h.stages = 2; h.splits = 1; h.kt_per_split = a.nkt;
        return h;
    }
    if (a.head_wf) {               // MFMA-form head: the 256x256 tile or (partial form only) the 128x128 8-wave tile; 2-stage ring;
        Plan h = pl;               // never split: the head needs the finished, activated sums
        const bool small = a.head_parts > 0 && ((d->tile_mr == 2 && d->tile_nr == 2) || (d->tile_mr <= 0 && a.M < 256 * 8));
        h.mr = small ? 2 : 4; h.nr = small ? 2 : 4; h.waves = 8; h.stages = 2; h.splits = 1; h.kt_per_split = a.nkt;
        return h;
    }
    if (d->tile_mr <= 0 || d->tile_nr <= 0) return pl;
    Plan req = pl;
    req.mr = d->tile_mr;
    req.nr = d->tile_nr;
    req.waves = d->tile_waves > 0 ? d->tile_waves : 4;
    req.stages = d->tile_stages > 0 ? d->tile_stages : 2;
    const bool f16s = d->precision == 1 && d->x_format == 1;
    // the fp32 and fp32-input f16x3 kernels have the four 4-wave 2-stage tiles; the SPLIT16 kernel has more
    const bool ok = f16s ? conv_f16s_plan_ok(req, a)
                         : (req.mr <= 2 && req.nr <= 2 && req.waves == 4 && req.stages == 2);
    if (!ok) return pl;            // unknown override: fall back to the heuristic plan (never an error)
    int s = d->splits >= 1 ? d->splits : 1;
    if (a.mode == 1 || a.x2 || a.up_top) s = 1;
    s = min(s, a.nkt);
    req.kt_per_split = cdiv(a.nkt, s);
    req.splits = cdiv(a.nkt, req.kt_per_split);
    return req;
}

int conv_fill_args(const srcnn_conv_desc *d, ConvArgs &a)
{
    SRCNN_REQUIRE(d && d->x && d->w && (d->y || d->head_w || d->head_wf), "null pointer");
    SRCNN_REQUIRE(d->Cin > 0 && d->Cin % BK == 0, "Cin must be a positive multiple of 32");
    SRCNN_REQUIRE(d->x_cstride % 4 == 0, "x_cstride must be a multiple of 4 floats (16-B loads)");
    SRCNN_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0, "bad output shape");
    SRCNN_REQUIRE(d->mode == 0 || (d->mode == 1 && d->Cout % 4 == 0 && d->KH == 1 && d->KW == 1 && !d->residual) ||
                      (d->mode == 2 && d->B % 2 == 0 && !d->residual && d->precision == 1 && d->x_format == 1),
                  "bad mode");
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->residual; a.y = d->y; a.partial = nullptr;
    SRCNN_REQUIRE(d->precision == 0 || (d->precision == 1 && d->w_lo), "bad precision / missing w_lo");
    a.w_lo = d->w_lo;
    a.out_scale = d->precision == 1 ? d->w_inv_scale : 1.0f;
    a.x_fmt = d->x_format; a.y_fmt = d->y_format; a.res_fmt = d->res_format;
    SRCNN_REQUIRE((unsigned)a.x_fmt <= 1 && (unsigned)a.y_fmt <= 1 && (unsigned)a.res_fmt <= 1, "bad format");
    if (d->precision == 0)
        SRCNN_REQUIRE(a.x_fmt == 0 && a.y_fmt == 0 && (a.res_fmt == 0 || !d->residual), "fp32 engine needs F32 formats");
    // SPLIT16 input: every 32-element K-tile run must start on a group boundary.  Channel strides that are multiples of 8
    // give that for any geometry; the packed stem image (srcnn_stem_pack: x_cstride 4, groups aligned per row) gives it
    // for even strides with KW = 1 and no padding.
    if (a.x_fmt == 1)
        SRCNN_REQUIRE(d->x_cstride % 8 == 0 ||
                          (d->x_cstride == 4 && d->KW == 1 && d->pad == 0 && d->stride % 2 == 0 && d->Cin == 32),
                      "SPLIT16 needs channel strides that are multiples of 8 (or the packed-stem geometry)");
    if (a.y_fmt == 1) SRCNN_REQUIRE(d->y_cstride % 8 == 0 && d->y_coffset % 8 == 0, "SPLIT16 output alignment");
    if (d->precision == 1 && a.x_fmt == 0) SRCNN_REQUIRE(a.y_fmt == 0 && a.res_fmt == 0, "f16x3 with F32 input writes F32");
    a.stamp = debug_stamp_buffer();
    a.range_flag = nullptr;
    a.tag = d->layer_tag > 0 ? d->layer_tag : 0;
    a.m_limit = d->m_limit;
    a.m_limit_mul = d->m_limit_mul;
    if (a.m_limit) SRCNN_REQUIRE(d->precision == 1 && d->x_format == 1 && d->m_limit_mul > 0, "m_limit: SPLIT16 f16x3 engine, m_limit_mul > 0");
    a.x2 = static_cast<const float *>(d->x2);
    a.Cin2 = a.H2 = a.W2 = a.xcs2 = a.stride2 = 0;
    if (a.x2) {
        SRCNN_REQUIRE(d->precision == 1 && d->x_format == 1 && d->KH == 1 && d->KW == 1 && d->pad == 0 && d->mode == 0,
                      "x2: SPLIT16 f16x3 engine, 1x1 conv, mode 0");
        SRCNN_REQUIRE(d->Cin2 > 0 && d->Cin2 % BK == 0 && d->x2_cstride % 8 == 0 && d->x2_cstride >= d->Cin2 && d->stride2 >= 1,
                      "x2: Cin2 must be a positive multiple of 32, x2_cstride a multiple of 8");
        SRCNN_REQUIRE((d->OH - 1) * d->stride2 < d->H2 && (d->OW - 1) * d->stride2 < d->W2, "x2: output grid exceeds the second input");
        SRCNN_REQUIRE((long long)d->W2 * d->x2_cstride * 4 * ((256 / d->OW + 3) * (long long)d->stride2 + 1) < (1LL << 31),
                      "x2: the input rows under one 256-pixel tile must span < 2 GB");
        a.Cin2 = d->Cin2; a.H2 = d->H2; a.W2 = d->W2; a.xcs2 = d->x2_cstride; a.stride2 = d->stride2;
    }
    a.head_w = static_cast<const float *>(d->head_w);
    a.head_b = static_cast<const float *>(d->head_bias);
    a.head_y = static_cast<float *>(d->head_y);
    a.head_scale = d->head_scale;
    if (a.head_w) {
        const int cq = d->mode == 1 ? d->Cout / 4 : d->Cout;
        SRCNN_REQUIRE(d->precision == 1 && d->x_format == 1 && a.head_y && a.head_b && d->head_cout == 6 && cq == 256 && !d->x2 &&
                          !d->residual && d->mode != 2,
                      "fused head: SPLIT16 f16x3 engine, 6 head channels over 256-channel pixels, no residual / second input / mode 2");
        // the head lives in the vector epilogue of the 256x256 tile (the general path is compiled out there): a channel stride or
        // offset that is not a multiple of 8 would skip it and leave head_y unwritten
        SRCNN_REQUIRE((d->y_cstride & 7) == 0 && (d->y_coffset & 7) == 0, "fused head: y_cstride and y_coffset must be multiples of 8");
    }
    a.head_wf = d->head_wf;
    a.head_rows = d->head_rows; a.head_n = d->head_cout; a.head_parts = d->head_parts; a.head_plane = d->head_plane;
    if (a.head_wf) {
        const int cq = d->mode == 1 ? d->Cout / 4 : d->Cout;
        SRCNN_REQUIRE(!a.head_w && d->precision == 1 && d->x_format == 1 && a.head_y && !d->x2 && !d->residual,
                      "MFMA-form head: SPLIT16 f16x3 engine, no residual / second input, not together with head_w");
        SRCNN_REQUIRE(d->head_rows >= 8 && d->head_rows <= 24 && (d->head_rows & 7) == 0 && d->head_cout >= 1 && d->head_cout <= d->head_rows,
                      "MFMA-form head: head_rows a multiple of 8 up to 24, 1 <= head_cout <= head_rows");
        SRCNN_REQUIRE((d->y_cstride & 7) == 0 && (d->y_coffset & 7) == 0 && (cq & 7) == 0, "MFMA-form head: channel counts / strides multiples of 8");
        if (d->head_parts == 0)
            SRCNN_REQUIRE(cq == 256 && a.head_b && d->mode != 2, "MFMA-form head, final form: 256-channel pixels (one 256x256 tile owns them), bias, not mode 2");
        else
            SRCNN_REQUIRE(d->mode != 1 && d->Cout % 256 == 0 && d->head_parts >= (d->mode == 2 ? 2 : 1) * (d->Cout / 128) && d->head_plane > 0,
                          "MFMA-form head, partial form: Cout a multiple of 256, head_parts planes for every (eye, 128-column tile)");
    }
    {
        static const int nt_default = [] { const char *e = std::getenv("SRCNN_NT_STORES"); return e ? std::atoi(e) : 0; }();   // A/B switch
        a.nt_out = nt_default;
    }
    a.up_top = d->up_top;
    a.up_fmt = d->up_format; a.up_TH = d->up_H; a.up_TW = d->up_W;
    if (a.up_top) {
        SRCNN_REQUIRE(d->precision == 1 && d->x_format == 1 && d->mode == 0 && !d->residual && !a.head_w && !a.head_wf && !a.m_limit,
                      "up_top: SPLIT16 f16x3 engine, mode 0, no residual / fused head / row limit");
        SRCNN_REQUIRE((unsigned)d->up_format <= 1 && d->up_H > 0 && d->up_W > 0 && d->up_H <= d->OH && d->up_W <= d->OW,
                      "up_top: bad format / the top map must not be larger than the output");
        SRCNN_REQUIRE((d->Cout & 7) == 0 && (d->y_cstride & 7) == 0 && (d->y_coffset & 7) == 0 && d->y,
                      "up_top: channel counts / strides multiples of 8 (the addition lives in the vector epilogue)");
    }
    if (a.y_fmt == 1 || a.head_wf) {
        a.range_flag = range_flag_word();
        SRCNN_REQUIRE(a.range_flag != nullptr, "range flag allocation failed");
    }
    // the SPLIT16 engine addresses x through a buffer descriptor: 32-bit lane offsets relative to the first input row of a
    // tile (<= 256 output pixels: a few rows, also across an image boundary) -- any tensor size, but one tile's rows must
    // span < 2 GB
    if (d->precision == 1 && a.x_fmt == 1)
        SRCNN_REQUIRE(d->KH <= 8 && d->KW <= 8, "SPLIT16 engine: kernel sizes up to 8 (tap validity is kept as two 8-bit fields)");
    if (d->precision == 1 && a.x_fmt == 1)
        SRCNN_REQUIRE((long long)d->W * d->x_cstride * 4 * ((256 / d->OW + 3) * (long long)d->stride + d->KH) < (1LL << 31),
                      "SPLIT16 engine: the input rows under one 256-pixel tile must span < 2 GB");
    a.nimg = d->B;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.xcs = d->x_cstride;
    a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.ycs = d->y_cstride; a.yco = d->y_coffset; a.rcs = d->res_cstride; a.relu = d->relu; a.mode = d->mode;
    {
        static const int tap_inner_default = [] { const char *e = std::getenv("SRCNN_TAP_INNER"); return e ? std::atoi(e) : 1; }();     // A/B switch; default on
        a.tap_inner = (tap_inner_default && d->KH * d->KW > 1 && d->Cin > BK) ? 1 : 0;
    }
    const long long M = (long long)d->B * d->OH * d->OW;
    SRCNN_REQUIRE(M < (1LL << 31) && (long long)d->B * d->H * d->W < (1LL << 31), "tensor too large");
    a.M = (int)M;
    {
        // fully connected shapes (fewer rows than output channels): the weights are the big operand, so consecutive workgroups
        // -- one XCD's share, running at the same time -- take the few M tiles of ONE weight slab, which then comes from HBM once
        static const int m_fast_default = [] { const char *e = std::getenv("SRCNN_M_FAST"); return e ? std::atoi(e) : 1; }();   // A/B switch
        a.m_fast = (m_fast_default && a.M < d->Cout) ? 1 : 0;
    }
    a.K = d->KH * d->KW * d->Cin + a.Cin2;
    a.ctiles = d->Cin / BK;
    a.nkt = a.K / BK;
    return SRCNN_OK;
}

template <int MR, int NR>
static void launch(const ConvArgs &a, int splits, hipStream_t st)
{
    SRCNN_LAUNCH((conv_mfma_kernel<MR, NR>), dim3(a.mtiles * a.ntiles, splits), dim3(256), 0, st, a);
}

}  // namespace srcnn

extern "C" {

size_t srcnn_conv2d_workspace_bytes(const srcnn_conv_desc *d)
{
    using namespace srcnn;
    ConvArgs a;
    if (conv_fill_args(d, a) != SRCNN_OK) return 0;
    Plan pl = plan_for(d, a);
    if (pl.splits <= 1) return 256;
    return align_up((size_t)pl.splits * a.M * a.Cout * sizeof(float), 256);
}

int srcnn_conv2d(const srcnn_conv_desc *d, void *workspace, size_t workspace_bytes, srcnn_stream_t stream)
{
    using namespace srcnn;
    ConvArgs a;
    int rc = conv_fill_args(d, a);
    if (rc != SRCNN_OK) return rc;
    Plan pl = plan_for(d, a);
    a.kt_per_split = pl.kt_per_split;
    a.mtiles = cdiv(a.M, 64 * pl.mr);
    a.ntiles = cdiv(a.Cout, 64 * pl.nr);
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * a.M * a.Cout * sizeof(float);
        if (!workspace || workspace_bytes < need) {
            set_error("srcnn_conv2d: workspace too small (%zu < %zu)", workspace_bytes, need);
            return SRCNN_ERR_WORKSPACE;
        }
        a.partial = static_cast<float *>(workspace);
    }
    hipStream_t st = as_stream(stream);
    const bool prof = prof_enabled();
    if (prof) prof_begin(st);
    if (d->precision == 1 && a.x_fmt == 1 && a.stamp == nullptr)       // debug: a stamp arena hands every launch its own region
        a.stamp = debug_stamp_region(a.mtiles * a.ntiles * pl.splits, a.tag, pl.stages * 128 * 64 * (pl.mr + pl.nr), pl.waves * 64,
                                     a.M, a.Cout, a.K);
    if (d->precision == 1 && a.x_fmt == 1) launch_conv_f16s(a, pl, st);
    else if (d->precision == 1) launch_conv_f16x3(a, pl, st);
    else if (pl.mr == 2 && pl.nr == 2) launch<2, 2>(a, pl.splits, st);
    else if (pl.mr == 2 && pl.nr == 1) launch<2, 1>(a, pl.splits, st);
    else if (pl.mr == 1 && pl.nr == 2) launch<1, 2>(a, pl.splits, st);
    else launch<1, 1>(a, pl.splits, st);
    if (pl.splits > 1) {
        const size_t total = (size_t)a.M * a.Cout;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        SRCNN_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a, pl.splits);
    }
    if (prof) prof_end(st, 2.0 * (double)a.M * (double)a.Cout * (double)a.K);
    return check_launch("srcnn_conv2d");
}

}  // extern "C"
