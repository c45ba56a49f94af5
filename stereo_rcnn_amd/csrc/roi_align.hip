// Legacy point-lattice ROIAlign (gfx950).
//
// Reference: lib/model/roi_align/src/roi_align_kernel.cu:15-91 (one thread per output
// element, NCHW) + modules/roi_align.py:26-29 (RoIAlignAvg = (A+1)^2 lattice, then
// avg_pool2d(2, stride 1)) + stereo_rcnn/stereo_rcnn.py:110-139 (pyramid level routing).
//
// Two entry points:
//   * roi_align_forward_cuda : the reference's operator, same layout/semantics (NCHW in,
//     (n,C,ah,aw) out), kept as the drop-in symbol and as the op-level parity target.
//   * srcnn_pyramid_roi_align: the MI355X-native fused form used by the forward pass:
//     NHWC maps (the 4 bilinear taps of a lattice point are 4 coalesced channel runs),
//     level routing on the device (no nonzero()/host sync), both lattice rows of an output
//     row in registers, 2x2 average fused, result written straight into the (left|right)
//     channel slice of the head's GEMM operand.
// The float/double promotion pattern of the reference kernel (its `1.` literals) is
// reproduced operation by operation; the library is built with -ffp-contract=off.
#include "conv_common.h"
#include <cstdlib>

namespace srcnn {

struct RoiGeom {
    float start_w, start_h, bin_w, bin_h;
    int batch;
};

__device__ __forceinline__ RoiGeom roi_geom(const float *r, float scale, int ah, int aw)
{
    RoiGeom g;
    g.batch = (int)r[0];
    g.start_w = r[1] * scale;
    g.start_h = r[2] * scale;
    float end_w = r[3] * scale;
    float end_h = r[4] * scale;
    float roi_w = fmaxf((float)((double)(end_w - g.start_w) + 1.), 0.0f);   // roi_align_kernel.cu:40
    float roi_h = fmaxf((float)((double)(end_h - g.start_h) + 1.), 0.0f);   // :41
    g.bin_h = (float)((double)roi_h / ((double)ah - 1.));                   // :42
    g.bin_w = (float)((double)roi_w / ((double)aw - 1.));                   // :43
    return g;
}

// value of one lattice point; `at(y, x)` fetches the feature value
template <typename Fetch>
__device__ __forceinline__ float lattice_point(float h, float w, int height, int width, Fetch at)
{
    if (h < 0 || h >= height || w < 0 || w >= width) return 0.0f;           // :54-55
    int hstart = (int)fminf(floorf(h), (float)(height - 2));                 // :48
    int wstart = (int)fminf(floorf(w), (float)(width - 2));                  // :49
    float h_ratio = h - (float)hstart;
    float w_ratio = w - (float)wstart;
    // :64-67 with C++'s usual arithmetic conversions, left to right: `1.` is a double, so the first two terms are double
    // products; `down * h_ratio` is float x float (rounded to float) before it meets a double, and the last term is a
    // float product throughout.  (Checked against the reference's own kernel built for gfx950: tests/test_ref_kernels_gpu.py.)
    const double hr1 = 1. - (double)h_ratio, wr1 = 1. - (double)w_ratio;
    const float dl_h = at(hstart + 1, wstart) * h_ratio;
    const float dr_hw = at(hstart + 1, wstart + 1) * h_ratio * w_ratio;
    double v = (double)at(hstart, wstart) * hr1 * wr1
             + (double)at(hstart, wstart + 1) * hr1 * (double)w_ratio
             + (double)dl_h * wr1
             + (double)dr_hw;
    return (float)v;
}

__global__ void roi_align_nchw_kernel(int total, const float *__restrict__ feat, float scale, int height,
                                      int width, int channels, int ah, int aw, const float *__restrict__ rois,
                                      float *__restrict__ out)
{
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += blockDim.x * gridDim.x) {
        int pw = idx % aw;
        int ph = (idx / aw) % ah;
        int c = (idx / aw / ah) % channels;
        int n = idx / aw / ah / channels;
        const float *r = rois + (size_t)n * 5;
        RoiGeom g = roi_geom(r, scale, ah, aw);
        // reference: int img_start = roi_batch_ind * channels * height * width (float product, :51)
        int img_start = (int)(r[0] * (float)channels * (float)height * (float)width);
        float h = (float)ph * g.bin_h + g.start_h;
        float w = (float)pw * g.bin_w + g.start_w;
        const float *plane = feat + img_start + (size_t)c * height * width;
        out[idx] = lattice_point(h, w, height, width,
                                 [&](int y, int x) { return plane[y * width + x]; });
    }
}

struct PyramidArgs {
    const float *maps[4];
    int mh[4], mw[4];
    float scale[4];
    const int *roi_limit;        // device-side count of the rois that matter (blocks of later rois exit) or nullptr
};

// grid (A, n); block = C threads (C multiple of 64, <= 1024). One output row per block.
template <int A>
__global__ void pyramid_roi_align_kernel(PyramidArgs pa, int channels, const float *__restrict__ rois,
                                         float *__restrict__ out, int out_cstride, int out_coffset, int mfmt, int ofmt)
{
    const int n = blockIdx.y, py = blockIdx.x, c = threadIdx.x;
    if (pa.roi_limit && n >= *pa.roi_limit) return;
    const float *r = rois + (size_t)n * 5;
    // level routing, stereo_rcnn.py:113-119 (natural log; round half away from zero; clamp 2..5)
    float bh = r[4] - r[2] + 1.0f;
    float bw = r[3] - r[1] + 1.0f;
    float lv = logf(sqrtf(bh * bw) / 224.0f) + 4.0f;
    lv = copysignf(floorf(fabsf(lv) + 0.5f), lv);
    lv = fminf(fmaxf(lv, 2.0f), 5.0f);
    const int l = __builtin_amdgcn_readfirstlane((int)lv - 2);   // same roi for the whole block
    const int height = pa.mh[l], width = pa.mw[l];
    RoiGeom g = roi_geom(r, pa.scale[l], A + 1, A + 1);
    const float *base = pa.maps[l];
    const size_t img = (size_t)g.batch * height * width;
    auto at = [&](int y, int x) { return act_load(base, mfmt, img + (size_t)y * width + x, channels, c); };
    float top[A + 1], bot[A + 1];
    const float h0 = (float)py * g.bin_h + g.start_h;
    const float h1 = (float)(py + 1) * g.bin_h + g.start_h;
#pragma unroll
    for (int px = 0; px <= A; ++px) {
        float w = (float)px * g.bin_w + g.start_w;
        top[px] = lattice_point(h0, w, height, width, at);
        bot[px] = lattice_point(h1, w, height, width, at);
    }
#pragma unroll
    for (int px = 0; px < A; ++px) {
        float s = top[px];
        s = s + top[px + 1];
        s = s + bot[px];
        s = s + bot[px + 1];
        act_store(out, ofmt, (size_t)(n * A + py) * A + px, out_cstride, out_coffset + c, s * 0.25f);
    }
}

// value of one lattice point for 8 consecutive channels; `at8(y, x)` fetches the 8-channel group.  Same arithmetic per
// channel as lattice_point() above.
template <typename Fetch8>
__device__ __forceinline__ float8 lattice_point8(float h, float w, int height, int width, Fetch8 at8)
{
    float8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = 0.0f;
    if (h < 0 || h >= height || w < 0 || w >= width) return r;
    const int hstart = (int)fminf(floorf(h), (float)(height - 2));
    const int wstart = (int)fminf(floorf(w), (float)(width - 2));
    const float h_ratio = h - (float)hstart;
    const float w_ratio = w - (float)wstart;
    const double hr1 = 1. - (double)h_ratio, wr1 = 1. - (double)w_ratio;
    const float8 ul = at8(hstart, wstart), ur = at8(hstart, wstart + 1);
    const float8 dl = at8(hstart + 1, wstart), dr = at8(hstart + 1, wstart + 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float dl_h = dl.v[e] * h_ratio;
        const float dr_hw = dr.v[e] * h_ratio * w_ratio;
        const double v = (double)ul.v[e] * hr1 * wr1 + (double)ur.v[e] * hr1 * (double)w_ratio + (double)dl_h * wr1 +
                         (double)dr_hw;
        r.v[e] = (float)v;
    }
    return r;
}

// The form the forward uses: one thread per (roi, output row, 8-channel group), block (C/8, rows) of one or two wavefronts,
// grid (ceil(A/rows), n).
// A thread walks the A+1 lattice columns of its two lattice rows, keeps the previous column, and emits one 32-byte output
// group per step: every tap is a 32-byte load (both activation formats), every store 2 x 16 bytes, and the 32 groups of a
// row read 1 KB contiguous per tap.  (The per-channel kernel above issued 2-byte accesses: 8x the memory instructions.)
template <int A>
__global__ void pyramid_roi_align8_kernel(PyramidArgs pa, int channels, const float *__restrict__ rois,
                                          float *__restrict__ out, int out_cstride, int out_coffset, int mfmt, int ofmt)
{
    const int n = blockIdx.y, py = blockIdx.x * blockDim.y + threadIdx.y, g = threadIdx.x;
    if (py >= A || (pa.roi_limit && n >= *pa.roi_limit)) return;
    const float *r = rois + (size_t)n * 5;
    // level routing, stereo_rcnn.py:113-119 (natural log; round half away from zero; clamp 2..5)
    float bh = r[4] - r[2] + 1.0f;
    float bw = r[3] - r[1] + 1.0f;
    float lv = logf(sqrtf(bh * bw) / 224.0f) + 4.0f;
    lv = copysignf(floorf(fabsf(lv) + 0.5f), lv);
    lv = fminf(fmaxf(lv, 2.0f), 5.0f);
    const int l = __builtin_amdgcn_readfirstlane((int)lv - 2);   // same roi for the whole block
    const int height = pa.mh[l], width = pa.mw[l];
    const RoiGeom geo = roi_geom(r, pa.scale[l], A + 1, A + 1);
    const float *base = pa.maps[l];
    const size_t img = (size_t)geo.batch * height * width;
    auto at8 = [&](int y, int x) { return act_load8(base, mfmt, img + (size_t)y * width + x, channels, g); };
    const float h0 = (float)py * geo.bin_h + geo.start_h;
    const float h1 = (float)(py + 1) * geo.bin_h + geo.start_h;
    const float w0 = (float)0 * geo.bin_w + geo.start_w;                         // the reference's expression at px = 0
    float8 top_prev = lattice_point8(h0, w0, height, width, at8);
    float8 bot_prev = lattice_point8(h1, w0, height, width, at8);
#pragma unroll 2
    for (int px = 1; px <= A; ++px) {
        const float w = (float)px * geo.bin_w + geo.start_w;
        const float8 top = lattice_point8(h0, w, height, width, at8);
        const float8 bot = lattice_point8(h1, w, height, width, at8);
        float8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = top_prev.v[e];
            s = s + top.v[e];
            s = s + bot_prev.v[e];
            s = s + bot.v[e];
            o.v[e] = s * 0.25f;
        }
        act_store8(out, ofmt, (size_t)(n * A + py) * A + (px - 1), out_cstride, (out_coffset >> 3) + g, o);
        top_prev = top;
        bot_prev = bot;
    }
}


// The form the forward uses since round 5: ONE workgroup per roi, thread (g, ry) = 8-channel group g of LATTICE row ry (A+1 rows).
// Every lattice point is computed once (the row-pair form above computes each lattice row twice, for the output rows above and
// below it: twice the tap loads and twice the double-precision blends); neighbouring rows meet through a double-buffered LDS
// slot per lattice column, behind a barrier that leaves global loads in flight, and the taps of column px + 1 are requested
// before column px is blended.  Same arithmetic per lattice point and the same sum order as above: bit-identical output.
struct LatticeTaps {
    float8 ul, ur, dl, dr;
    float w_ratio;
    bool ok;
};

template <int A>
__global__ __launch_bounds__(32 * (A + 1)) void pyramid_roi_align8_roi_kernel(PyramidArgs pa, int channels, const float *__restrict__ rois,
                                                                              float *__restrict__ out, int out_cstride, int out_coffset,
                                                                              int mfmt, int ofmt)
{
    __shared__ float4 lat[2][A + 1][32][2];                     // [column parity][lattice row][group][8 floats]
    const int n = blockIdx.x, ry = threadIdx.y, g = threadIdx.x;
    if (pa.roi_limit && n >= *pa.roi_limit) return;             // (uniform: the whole workgroup leaves)
    const float *r = rois + (size_t)n * 5;
    // level routing, stereo_rcnn.py:113-119 (natural log; round half away from zero; clamp 2..5)
    float bh = r[4] - r[2] + 1.0f;
    float bw = r[3] - r[1] + 1.0f;
    float lv = logf(sqrtf(bh * bw) / 224.0f) + 4.0f;
    lv = copysignf(floorf(fabsf(lv) + 0.5f), lv);
    lv = fminf(fmaxf(lv, 2.0f), 5.0f);
    const int l = __builtin_amdgcn_readfirstlane((int)lv - 2);   // same roi for the whole block
    const int height = pa.mh[l], width = pa.mw[l];
    const RoiGeom geo = roi_geom(r, pa.scale[l], A + 1, A + 1);
    const float *base = pa.maps[l];
    const size_t img = (size_t)geo.batch * height * width;
    const bool live = g * 8 < channels;                          // (channels < 256: the upper groups only keep the barriers company)
    const float h = (float)ry * geo.bin_h + geo.start_h;
    const bool h_ok = !(h < 0 || h >= height);
    const int hstart = h_ok ? (int)fminf(floorf(h), (float)(height - 2)) : 0;
    const float h_ratio = h - (float)hstart;
    const double hr1 = 1. - (double)h_ratio;
    auto request = [&](int px) {
        LatticeTaps t;
        const float w = (float)px * geo.bin_w + geo.start_w;
        t.ok = h_ok && !(w < 0 || w >= width);                   // roi_align_kernel.cu:54-55
        const int wstart = t.ok ? (int)fminf(floorf(w), (float)(width - 2)) : 0;
        t.w_ratio = w - (float)wstart;
        const size_t p0 = img + (size_t)hstart * width + wstart; // (a point outside the map reads the map's first pixels and drops them)
        const int gl = live ? g : 0;
        t.ul = act_load8(base, mfmt, p0, channels, gl);
        t.ur = act_load8(base, mfmt, p0 + 1, channels, gl);
        t.dl = act_load8(base, mfmt, p0 + width, channels, gl);
        t.dr = act_load8(base, mfmt, p0 + width + 1, channels, gl);
        return t;
    };
    auto blend = [&](const LatticeTaps &t) {                     // lattice_point8's arithmetic
        float8 v;
        const double wr1 = 1. - (double)t.w_ratio;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dl_h = t.dl.v[e] * h_ratio;
            const float dr_hw = t.dr.v[e] * h_ratio * t.w_ratio;
            const double d = (double)t.ul.v[e] * hr1 * wr1 + (double)t.ur.v[e] * hr1 * (double)t.w_ratio + (double)dl_h * wr1 + (double)dr_hw;
            v.v[e] = t.ok ? (float)d : 0.0f;
        }
        return v;
    };
    LatticeTaps cur = request(0);
    float8 top_prev, bot_prev;
#pragma unroll
    for (int e = 0; e < 8; ++e) top_prev.v[e] = bot_prev.v[e] = 0.f;
#pragma unroll 1
    for (int px = 0; px <= A; ++px) {
        LatticeTaps nxt = cur;
        if (px < A) nxt = request(px + 1);                       // in flight across the barrier below
        const float8 top = blend(cur);
        float4 *slot = &lat[px & 1][ry][g][0];
        slot[0] = make_float4(top.v[0], top.v[1], top.v[2], top.v[3]);
        slot[1] = make_float4(top.v[4], top.v[5], top.v[6], top.v[7]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS visibility only: the next column's taps stay in flight
        if (ry < A) {
            const float4 b0 = lat[px & 1][ry + 1][g][0], b1 = lat[px & 1][ry + 1][g][1];
            const float8 bot = {{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}};
            if (px >= 1 && live) {
                float8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = top_prev.v[e];
                    s = s + top.v[e];
                    s = s + bot_prev.v[e];
                    s = s + bot.v[e];
                    o.v[e] = s * 0.25f;
                }
                act_store8(out, ofmt, (size_t)(n * A + ry) * A + (px - 1), out_cstride, (out_coffset >> 3) + g, o);
            }
            bot_prev = bot;
        }
        top_prev = top;
        cur = nxt;
    }
}

}  // namespace srcnn

// avg_pool2d / max_pool2d (kernel 2, stride 1) over the (planes, h, w) lattice of the legacy op: the reduction behind
// RoIAlignAvg / RoIAlignMax (modules/roi_align.py:26-29, 41-44).  Sum order of ATen's avg_pool2d (rows, then columns), x 0.25.
__global__ void pool2x2_s1_kernel(const float *__restrict__ x, size_t planes, int h, int w, float *__restrict__ y, int take_max)
{
    const int oh = h - 1, ow = w - 1;
    const size_t total = planes * oh * ow;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(idx % ow);
        const int py = (int)((idx / ow) % oh);
        const size_t pl = idx / ((size_t)ow * oh);
        const float *q = x + (pl * h + py) * w + px;
        const float a = q[0], b = q[1], c = q[w], d = q[w + 1];
        if (take_max) {
            // max_pool2d propagates NaN; fmaxf would drop it
            float m = a;
            m = (b > m || b != b) ? b : m;
            m = (c > m || c != c) ? c : m;
            m = (d > m || d != d) ? d : m;
            y[idx] = m;
        } else {
            float s = a;
            s = s + b;
            s = s + c;
            s = s + d;
            y[idx] = s * 0.25f;
        }
    }
}

extern "C" {

int srcnn_pool2x2_s1(const float *x, long long planes, int h, int w, float *y, int take_max, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(x && y && planes >= 0 && h >= 2 && w >= 2, "bad args (the lattice must be at least 2 x 2)");
    const size_t total = (size_t)planes * (h - 1) * (w - 1);
    if (total == 0) return SRCNN_OK;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    SRCNN_LAUNCH(pool2x2_s1_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, (size_t)planes, h, w, y, take_max);
    return check_launch("srcnn_pool2x2_s1");
}

int roi_align_forward_cuda(int aligned_height, int aligned_width, float spatial_scale, const float *features,
                           int batch, int channels, int height, int width, const float *rois, int num_rois,
                           int roi_cols, float *output, srcnn_stream_t stream)
{
    using namespace srcnn;
    (void)batch;
    if (roi_cols != 5) return 0;   // roi_align_cuda.c:19-22
    const long long total = (long long)num_rois * aligned_height * aligned_width * channels;
    if (total == 0) return 1;
    if (total > 0x7fffffffLL) {
        set_error("roi_align_forward_cuda: output too large");
        return 0;
    }
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads);
    SRCNN_LAUNCH(roi_align_nchw_kernel, dim3(blocks), dim3(threads), 0, as_stream(stream), (int)total,
                       features, spatial_scale, height, width, channels, aligned_height, aligned_width, rois,
                       output);
    return check_launch("roi_align_forward_cuda") == SRCNN_OK ? 1 : 0;
}

int srcnn_pyramid_roi_align(const float *const *maps_host, const int *mh_host, const int *mw_host, int channels,
                            float im_height, const float *rois, int num_rois, int A, float *out, int out_cstride,
                            int out_coffset, int maps_format, int out_format, const int *roi_limit, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(channels % 64 == 0 && channels <= 1024, "channels must be a multiple of 64, <= 1024");
    SRCNN_REQUIRE(A == 7 || A == 14, "A must be 7 or 14");
    SRCNN_REQUIRE((unsigned)maps_format <= 1 && (unsigned)out_format <= 1, "bad format");
    if (out_format == 1) SRCNN_REQUIRE(out_cstride % 8 == 0 && out_coffset % 8 == 0, "SPLIT16 output alignment");
    if (num_rois == 0) return SRCNN_OK;
    PyramidArgs pa;
    pa.roi_limit = roi_limit;
    for (int l = 0; l < 4; ++l) {
        pa.maps[l] = maps_host[l];
        pa.mh[l] = mh_host[l];
        pa.mw[l] = mw_host[l];
        // python: feat_maps[i].size(2) / im_info[0][0] -> double, narrowed to float at the C boundary
        pa.scale[l] = (float)((double)mh_host[l] / (double)im_height);
    }
    static const int roi_form = [] { const char *e = std::getenv("SRCNN_ROI_ALIGN_FORM"); return e ? std::atoi(e) : 1; }();   // A/B switch
    if (roi_form && out_cstride % 8 == 0 && out_coffset % 8 == 0 && channels <= 256) {
        // one workgroup per roi, one thread row per lattice row: every lattice point computed once
        dim3 grid(num_rois), block(32, A + 1);
        if (A == 7)
            SRCNN_LAUNCH(pyramid_roi_align8_roi_kernel<7>, grid, block, 0, as_stream(stream), pa, channels, rois, out,
                               out_cstride, out_coffset, maps_format, out_format);
        else
            SRCNN_LAUNCH(pyramid_roi_align8_roi_kernel<14>, grid, block, 0, as_stream(stream), pa, channels, rois, out,
                               out_cstride, out_coffset, maps_format, out_format);
        return check_launch("srcnn_pyramid_roi_align");
    }
    if (out_cstride % 8 == 0 && out_coffset % 8 == 0) {
        const int G = channels / 8, rows = G <= 32 ? 64 / G : 1;   // one or two wavefronts per block: thousands of small blocks
        dim3 grid8((A + rows - 1) / rows, num_rois), block8(G, rows);
        if (A == 7)
            SRCNN_LAUNCH(pyramid_roi_align8_kernel<7>, grid8, block8, 0, as_stream(stream), pa, channels, rois, out,
                               out_cstride, out_coffset, maps_format, out_format);
        else
            SRCNN_LAUNCH(pyramid_roi_align8_kernel<14>, grid8, block8, 0, as_stream(stream), pa, channels, rois, out,
                               out_cstride, out_coffset, maps_format, out_format);
        return check_launch("srcnn_pyramid_roi_align");
    }
    dim3 grid(A, num_rois), block(channels);
    if (A == 7)
        SRCNN_LAUNCH(pyramid_roi_align_kernel<7>, grid, block, 0, as_stream(stream), pa, channels, rois, out,
                           out_cstride, out_coffset, maps_format, out_format);
    else
        SRCNN_LAUNCH(pyramid_roi_align_kernel<14>, grid, block, 0, as_stream(stream), pa, channels, rois,
                           out, out_cstride, out_coffset, maps_format, out_format);
    return check_launch("srcnn_pyramid_roi_align");
}

}  // extern "C"
