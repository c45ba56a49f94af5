// HBM-bound NHWC helpers of the trunk/FPN: stem repack, ceil-mode max-pool,
// align_corners bilinear upsample + add, stride-2 subsample, layout edge transposes.
// All are one-pass, channel-vectorised (float4 = 16 B/lane, coalesced over C).
//
// Reference semantics: resnet.py:113 (MaxPool2d(3,2,0,ceil_mode=True)),
// stereo_rcnn.py:91-108 (_upsample_add; torch-0.3 bilinear == align_corners=True),
// stereo_rcnn.py:39,168 (MaxPool2d(1, stride=2)).
#include "conv_common.h"

namespace srcnn {

// NCHW (B,3,H,W) -> NHWC4 with a zero border: out (B, H+6, W+8, 4); pixel (y,x) -> (y+3, x+3).
__global__ void stem_pack_kernel(const float *__restrict__ im, int B, int H, int W, float4 *__restrict__ out)
{
    const int HP = H + 6, WP = W + 8;
    const size_t total = (size_t)B * HP * WP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int xp = (int)(idx % WP);
        const int yp = (int)((idx / WP) % HP);
        const int b = (int)(idx / ((size_t)WP * HP));
        const int y = yp - 3, x = xp - 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const float *p = im + (size_t)b * 3 * H * W + (size_t)y * W + x;
            v.x = p[0];
            v.y = p[(size_t)H * W];
            v.z = p[(size_t)2 * H * W];
        }
        out[idx] = v;
    }
}

// Same image in the SPLIT16 layout the DMA conv engine reads: per padded row, every two pixels (8 floats of the NHWC4 row)
// become one 32-byte group [8 x f16 hi][8 x f16 lo] at the same byte offset the floats had.  Groups are aligned to the
// start of each ROW (the row pitch, (W+8)*16 B, need not be a multiple of 32 B): every 8-pixel tap run of the stride-2
// stem starts on an even pixel, i.e. on a group boundary.  An odd last pixel of a row is never read by the stem and
// is not written.
__global__ void stem_pack_split16_kernel(const float *__restrict__ im, int B, int H, int W, char *__restrict__ out)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int HP = H + 6, WP = W + 8, GP = WP / 2;
    const size_t total = (size_t)B * HP * GP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int gp = (int)(idx % GP);
        const int yp = (int)((idx / GP) % HP);
        const int b = (int)(idx / ((size_t)GP * HP));
        const int y = yp - 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if ((unsigned)y < (unsigned)H) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = 2 * gp + q - 3;
                if ((unsigned)x < (unsigned)W) {
                    const float *p = im + (size_t)b * 3 * H * W + (size_t)y * W + x;
                    v[4 * q + 0] = p[0];
                    v[4 * q + 1] = p[(size_t)H * W];
                    v[4 * q + 2] = p[(size_t)2 * H * W];
                }
            }
        }
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (_Float16)v[e];
            lo[e] = (_Float16)(v[e] - (float)hi[e]);
        }
        char *dst = out + ((size_t)b * HP + yp) * WP * 16 + (size_t)gp * 32;
        *reinterpret_cast<h8 *>(dst) = hi;
        *reinterpret_cast<h8 *>(dst + 16) = lo;
    }
}

// works on 8-channel groups; input F32, output F32 or SPLIT16
__global__ void maxpool3x3s2_kernel(const float *__restrict__ x, int B, int H, int W, int C, float *__restrict__ y,
                                    int OH, int OW, int yfmt)
{
    const int G = C / 8;
    const size_t total = (size_t)B * OH * OW * G;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        size_t r = idx / G;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH);
        const int b = (int)(r / OH);
        const int h0 = oh * 2, w0 = ow * 2;
        const int h1 = min(h0 + 3, H), w1 = min(w0 + 3, W);   // ceil_mode windows are clipped at the edge
        float8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m.v[e] = -INFINITY;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const float8 v = act_load8(x, 0, ((size_t)b * H + h) * W + w, C, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) m.v[e] = fmaxf(m.v[e], v.v[e]);
            }
        act_store8(y, yfmt, ((size_t)b * OH + oh) * OW + ow, C, g, m);
    }
}

// y = bilinear(top, align_corners=True -> (H,W)) + lateral.  Index/weight arithmetic follows
// ATen's upsample_bilinear2d (area_pixel_compute_scale: (in-1)/(out-1); h1 = (int)h1r;
// lambda = h1r - h1), accumulation order w0*(... ) as written there.
// grid (ceil(W * C/8 / 256), B * H): one output row per blockIdx.y, so the row geometry (source rows, vertical weights) is
// wave-uniform and the per-thread index arithmetic is one division by the group count.
__global__ void upsample_add_kernel(const float *__restrict__ top, int TH, int TW, const float *__restrict__ lat,
                                    int B, int H, int W, int C, float *__restrict__ y, int top_fmt, int yfmt)
{
    (void)B;
    const float rh = H > 1 ? (float)(TH - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(TW - 1) / (float)(W - 1) : 0.f;
    const int G = C / 8;
    const int row = blockIdx.y;                      // b * H + h
    const int b = row / H, h = row - b * H;
    const float h1r = rh * (float)h;
    const int h1 = (int)h1r;
    const int h1p = (h1 < TH - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const size_t trow0 = ((size_t)b * TH + h1) * TW, trow1 = trow0 + (size_t)h1p * TW;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)(W * G)) return;
    const int w = (int)(idx / (unsigned)G), g = (int)(idx - (unsigned)w * (unsigned)G);
    const float w1r = rw * (float)w;
    const int w1 = (int)w1r;
    const int w1p = (w1 < TW - 1) ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float8 a = act_load8(top, top_fmt, trow0 + w1, C, g), bb = act_load8(top, top_fmt, trow0 + w1 + w1p, C, g);
    const float8 cc = act_load8(top, top_fmt, trow1 + w1, C, g), d = act_load8(top, top_fmt, trow1 + w1 + w1p, C, g);
    const size_t pix = (size_t)row * W + w;
    const float8 l = act_load8(lat, 0, pix, C, g);
    float8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        o.v[e] = (h0l * (w0l * a.v[e] + w1l * bb.v[e]) + h1l * (w0l * cc.v[e] + w1l * d.v[e])) + l.v[e];
    act_store8(y, yfmt, pix, C, g, o);
}

// NHWC activation format conversion on 8-channel groups
__global__ void act_convert_kernel(const void *__restrict__ x, int xfmt, void *__restrict__ y, int yfmt, size_t pixels,
                                   int C)
{
    const int G = C / 8;
    const size_t total = pixels * G;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = idx / G;
        const int g = (int)(idx - pix * G);
        act_store8(y, yfmt, pix, C, g, act_load8(x, xfmt, pix, C, g));
    }
}

// A0 (demo.py:103-129, blob.py:39-64): uint8 RGB (H, W, 3) -> float32 BGR planes, minus PIXEL_MEANS,
// bilinear resize by `scale` with OpenCV INTER_LINEAR geometry (half-pixel centres, source = (dst+0.5)/scale-0.5,
// border replicate).  Mean subtraction happens BEFORE the resize, as in the reference.
__global__ void preprocess_kernel(const unsigned char *__restrict__ img, int H, int W, float *__restrict__ out, int OH,
                                  int OW, float inv_scale, float m0, float m1, float m2)
{
    const size_t total = (size_t)OH * OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW), oy = (int)(idx / OW);
        // ATen upsample_bilinear2d(align_corners=False, given scale): src = scale*(dst+0.5)-0.5, clamped at 0
        float sy = inv_scale * ((float)oy + 0.5f) - 0.5f;
        float sx = inv_scale * ((float)ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const unsigned char *p00 = img + ((size_t)y0 * W + x0) * 3, *p01 = img + ((size_t)y0 * W + x1) * 3;
        const unsigned char *p10 = img + ((size_t)y1 * W + x0) * 3, *p11 = img + ((size_t)y1 * W + x1) * 3;
        const float mean[3] = {m0, m1, m2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {      // output channel c = B,G,R  <-  input channel 2-c
            const int ic = 2 - c;
            const float a = (float)p00[ic] - mean[c], b = (float)p01[ic] - mean[c];
            const float cc = (float)p10[ic] - mean[c], d = (float)p11[ic] - mean[c];
            out[(size_t)c * total + idx] = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
        }
    }
}

__global__ void subsample2_kernel(const float4 *__restrict__ x, int B, int H, int W, int C4, float4 *__restrict__ y,
                                  int OH, int OW)
{
    const size_t total = (size_t)B * OH * OW * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        size_t r = idx / C4;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH);
        const int b = (int)(r / OH);
        y[idx] = x[(((size_t)b * H + 2 * oh) * W + 2 * ow) * C4 + c];
    }
}

// tiled transpose of the innermost two "axes": in (B, R, S) -> out (B, S, R)
__global__ void transpose_kernel(const float *__restrict__ in, int R, int S, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const float *src = in + (size_t)b * R * S;
    float *dst = out + (size_t)b * R * S;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, s = s0 + threadIdx.x;
        if (r < R && s < S) tile[i][threadIdx.x] = src[(size_t)r * S + s];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int s = s0 + i, r = r0 + threadIdx.x;
        if (r < R && s < S) dst[(size_t)s * R + r] = tile[threadIdx.x][i];
    }
}

static inline int grid_for(size_t total, int threads) { return (int)std::min<size_t>((total + threads - 1) / threads, 16384); }

}  // namespace srcnn

extern "C" {

int srcnn_stem_pack(const float *im_nchw, int B, int H, int W, float *out, int out_format, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(im_nchw && out && B > 0 && H > 0 && W > 0, "bad args");
    SRCNN_REQUIRE((unsigned)out_format <= 1, "bad format");
    if (out_format == 1) {
        const size_t groups = (size_t)B * (H + 6) * ((W + 8) / 2);
        hipLaunchKernelGGL(stem_pack_split16_kernel, dim3(grid_for(groups, 256)), dim3(256), 0, as_stream(stream), im_nchw, B,
                           H, W, reinterpret_cast<char *>(out));
        return check_launch("srcnn_stem_pack");
    }
    const size_t total = (size_t)B * (H + 6) * (W + 8);
    hipLaunchKernelGGL(stem_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), im_nchw, B, H, W,
                       reinterpret_cast<float4 *>(out));
    return check_launch("srcnn_stem_pack");
}

int srcnn_maxpool3x3s2_ceil(const float *x, int B, int H, int W, int C, float *y, int OH, int OW, int y_format,
                            srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    SRCNN_REQUIRE((OH - 1) * 2 < H && (OW - 1) * 2 < W, "output too large for input");
    SRCNN_REQUIRE((unsigned)y_format <= 1, "bad format");
    const size_t total = (size_t)B * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, B, H, W, C,
                       y, OH, OW, y_format);
    return check_launch("srcnn_maxpool3x3s2_ceil");
}

int srcnn_upsample_add(const float *top, int TH, int TW, const float *lateral, int B, int H, int W, int C, float *y,
                       int top_format, int y_format, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    SRCNN_REQUIRE((unsigned)top_format <= 1 && (unsigned)y_format <= 1, "bad format");
    SRCNN_REQUIRE((long long)B * H <= 65535 && (long long)W * (C / 8) < (1LL << 31), "map too large");
    hipLaunchKernelGGL(upsample_add_kernel, dim3((W * (C / 8) + 255) / 256, B * H), dim3(256), 0, as_stream(stream), top, TH, TW,
                       lateral, B, H, W, C, y, top_format, y_format);
    return check_launch("srcnn_upsample_add");
}

int srcnn_act_convert(const void *x, int x_format, void *y, int y_format, long long pixels, int C,
                      srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(x && y && pixels >= 0 && C > 0 && C % 8 == 0, "bad args (C must be a multiple of 8)");
    SRCNN_REQUIRE((unsigned)x_format <= 1 && (unsigned)y_format <= 1, "bad format");
    if (pixels == 0) return SRCNN_OK;
    const size_t total = (size_t)pixels * (C / 8);
    hipLaunchKernelGGL(act_convert_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, x_format, y,
                       y_format, (size_t)pixels, C);
    return check_launch("srcnn_act_convert");
}

int srcnn_preprocess(const unsigned char *img_rgb, int H, int W, float scale, float *out_nchw, int OH, int OW,
                     srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(img_rgb && out_nchw && H > 0 && W > 0 && OH > 0 && OW > 0 && scale > 0.f, "bad args");
    // PIXEL_MEANS (config.py:170), BGR, narrowed to float32 after the float32-minus-float64 subtraction of the reference
    const size_t total = (size_t)OH * OW;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), img_rgb, H, W,
                       out_nchw, OH, OW, 1.0f / scale, 102.9801f, 115.9465f, 122.7717f);
    return check_launch("srcnn_preprocess");
}

int srcnn_subsample2(const float *x, int B, int H, int W, int C, float *y, int OH, int OW, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(C % 4 == 0, "C must be a multiple of 4");
    SRCNN_REQUIRE((OH - 1) * 2 < H && (OW - 1) * 2 < W, "output too large for input");
    const size_t total = (size_t)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), B, H, W, C / 4, reinterpret_cast<float4 *>(y), OH, OW);
    return check_launch("srcnn_subsample2");
}

int srcnn_nhwc_to_nchw(const float *x, int B, int H, int W, int C, float *y, srcnn_stream_t stream)
{
    using namespace srcnn;
    const int R = H * W, S = C;   // (B, HW, C) -> (B, C, HW)
    SRCNN_REQUIRE(B <= 65535, "batch too large");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(S, 32), cdiv(R, 32), B), dim3(32, 8), 0, as_stream(stream), x, R, S, y);
    return check_launch("srcnn_nhwc_to_nchw");
}

int srcnn_nchw_to_nhwc(const float *x, int B, int C, int H, int W, float *y, srcnn_stream_t stream)
{
    using namespace srcnn;
    const int R = C, S = H * W;   // (B, C, HW) -> (B, HW, C)
    SRCNN_REQUIRE(B <= 65535, "batch too large");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(S, 32), cdiv(R, 32), B), dim3(32, 8), 0, as_stream(stream), x, R, S, y);
    return check_launch("srcnn_nchw_to_nhwc");
}

}  // extern "C"
