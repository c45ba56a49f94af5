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
// (im2: images b >= B1 come from there -- the right eyes of a stereo batch packed behind the left ones in ONE launch; B1 = B: none)
__global__ void stem_pack_kernel(const float *__restrict__ im, const float *__restrict__ im2, int B1, int B, int H, int W,
                                 float4 *__restrict__ out)
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
            const float *p = (b < B1 ? im + (size_t)b * 3 * H * W : im2 + (size_t)(b - B1) * 3 * H * W) + (size_t)y * W + x;
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
__global__ void stem_pack_split16_kernel(const float *__restrict__ im, const float *__restrict__ im2, int B1, int B, int H, int W,
                                        char *__restrict__ out, unsigned *__restrict__ range_flag)
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
                    const float *p = (b < B1 ? im + (size_t)b * 3 * H * W : im2 + (size_t)(b - B1) * 3 * H * W) + (size_t)y * W + x;
                    v[4 * q + 0] = p[0];
                    v[4 * q + 1] = p[(size_t)H * W];
                    v[4 * q + 2] = p[(size_t)2 * H * W];
                }
            }
        }
        h8 hi, lo;
        float8 chk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            chk.v[e] = v[e];
            hi[e] = (_Float16)v[e];
            lo[e] = (_Float16)(v[e] - (float)hi[e]);
        }
        split16_guard(chk, range_flag, 9001);            // an input image beyond the f16 range: flag 9002 = input conversion
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
                                    int B, int H, int W, int C, float *__restrict__ y, int top_fmt, int yfmt,
                                    unsigned *__restrict__ range_flag)
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
    if (yfmt == 1) split16_guard(o, range_flag, 9000);        // a sum of two in-range maps can leave the f16 range
    act_store8(y, yfmt, pix, C, g, o);
}

// NHWC activation format conversion on 8-channel groups
__global__ void act_convert_kernel(const void *__restrict__ x, int xfmt, void *__restrict__ y, int yfmt, size_t pixels,
                                   int C, unsigned *__restrict__ range_flag)
{
    const int G = C / 8;
    const size_t total = pixels * G;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = idx / G;
        const int g = (int)(idx - pix * G);
        const float8 v = act_load8(x, xfmt, pix, C, g);
        if (yfmt == 1 && xfmt == 0) split16_guard(v, range_flag, 9001);
        act_store8(y, yfmt, pix, C, g, v);
    }
}

// A0 (demo.py:103-129, blob.py:39-64): uint8 RGB (H, W, 3) -> float32 BGR, minus PIXEL_MEANS, resized by
// cv2.resize(img, None, None, fx=s, fy=s, INTER_LINEAR).  The resize follows OpenCV's published float path operation by
// operation (modules/imgproc/src/resize.cpp; restated and pinned in oracle/preprocess.py): tap position
// (float)((d + 0.5) * (1/s) - 0.5) in double then float, cvFloor, float fraction; horizontal taps clamp AND zero the
// fraction at the borders, vertical taps clip the two rows but keep the fraction; HResizeLinear S[sx]*a0 + S[sx+1]*a1,
// VResizeLinear S0*b0 + S1*b1, each product and sum rounded to float32 on its own (this file is built with
// -ffp-contract=off).  Mean subtraction happens BEFORE the resize and in double, as numpy does for float32 -= float64.
struct ResizeTap {
    int s0, s1;
    float w0, w1;
    bool copy;     // horizontal only: dx >= xmax, D = S[cols-1] * 1.f
};

__device__ __forceinline__ ResizeTap resize_tap(int d, int n_src, double step, bool horizontal)
{
    ResizeTap t;
    float f = (float)(((double)d + 0.5) * step - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    t.copy = false;
    if (horizontal) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= n_src - 1) { s = n_src - 1; f = 0.f; t.copy = true; }
        t.s0 = s;
        t.s1 = min(s + 1, n_src - 1);
    } else {
        t.s0 = min(max(s, 0), n_src - 1);
        t.s1 = min(max(s + 1, 0), n_src - 1);
    }
    t.w1 = f;
    t.w0 = 1.f - f;
    return t;
}

__device__ __forceinline__ float u8_minus_mean(unsigned char v, double mean) { return (float)((double)v - mean); }

// resized BGR value of output pixel (oy, ox): bgr[0..2]
__device__ __forceinline__ void preprocess_pixel(const unsigned char *__restrict__ img, int H, int W, int oy, int ox,
                                                 double step, float bgr[3])
{
    const ResizeTap tx = resize_tap(ox, W, step, true), ty = resize_tap(oy, H, step, false);
    const unsigned char *r0 = img + (size_t)ty.s0 * W * 3, *r1 = img + (size_t)ty.s1 * W * 3;
    const double mean[3] = {102.9801, 115.9465, 122.7717};        // PIXEL_MEANS (config.py:170), BGR
#pragma unroll
    for (int c = 0; c < 3; ++c) {          // output channel c = B,G,R  <-  input channel 2-c
        const int ic = 2 - c;
        const float a = u8_minus_mean(r0[tx.s0 * 3 + ic], mean[c]), b = u8_minus_mean(r0[tx.s1 * 3 + ic], mean[c]);
        const float cc = u8_minus_mean(r1[tx.s0 * 3 + ic], mean[c]), d = u8_minus_mean(r1[tx.s1 * 3 + ic], mean[c]);
        const float h0 = tx.copy ? a : a * tx.w0 + b * tx.w1;
        const float h1 = tx.copy ? cc : cc * tx.w0 + d * tx.w1;
        bgr[c] = h0 * ty.w0 + h1 * ty.w1;
    }
}

// One thread per PAIR of padded stem-input pixels (the 32-byte group of the packed layouts, see stem_pack*_kernel):
// computes the (up to) two resized pixels once and writes them wherever asked -- the planar float32 network input
// (what forward() / dense alignment take) and/or the zero-bordered NHWC4 stem input in F32 or SPLIT16 form, so that
// no float32 intermediate has to be re-read to feed the stem.
__global__ void preprocess_pack_kernel(const unsigned char *__restrict__ img, int H, int W, double step, int OH, int OW,
                                       float *__restrict__ planar, char *__restrict__ packed, int packed_fmt)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int HP = OH + 6, WP = OW + 8, GP = (WP + 1) / 2;
    const size_t total = (size_t)HP * GP, plane = (size_t)OH * OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int gp = (int)(idx % GP), yp = (int)(idx / GP);
        const int y = yp - 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if ((unsigned)y < (unsigned)OH) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = 2 * gp + q - 3;
                if ((unsigned)x < (unsigned)OW) {
                    preprocess_pixel(img, H, W, y, x, step, v + 4 * q);
                    if (planar) {
                        float *o = planar + (size_t)y * OW + x;
                        o[0] = v[4 * q + 0];
                        o[plane] = v[4 * q + 1];
                        o[2 * plane] = v[4 * q + 2];
                    }
                }
            }
        }
        if (!packed) continue;
        char *dst = packed + (size_t)yp * WP * 16 + (size_t)gp * 32;
        const bool second = 2 * gp + 1 < WP;            // odd row length: the last group holds one pixel only
        if (packed_fmt == 0) {
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            if (second) *reinterpret_cast<float4 *>(dst + 16) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (second) {                            // SPLIT16 groups are whole or absent (stem_pack_split16_kernel)
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (_Float16)v[e];
                lo[e] = (_Float16)(v[e] - (float)hi[e]);
            }
            *reinterpret_cast<h8 *>(dst) = hi;
            *reinterpret_cast<h8 *>(dst + 16) = lo;
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

static int stem_pack_launch(const float *im, const float *im2, int B1, int B, int H, int W, float *out, int out_format,
                            srcnn_stream_t stream, const char *what)
{
    using namespace srcnn;
    if (out_format == 1) {
        const size_t groups = (size_t)B * (H + 6) * ((W + 8) / 2);
        SRCNN_LAUNCH(stem_pack_split16_kernel, dim3(grid_for(groups, 256)), dim3(256), 0, as_stream(stream), im, im2, B1, B,
                           H, W, reinterpret_cast<char *>(out), range_flag_word());
        return check_launch(what);
    }
    const size_t total = (size_t)B * (H + 6) * (W + 8);
    SRCNN_LAUNCH(stem_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), im, im2, B1, B, H, W,
                       reinterpret_cast<float4 *>(out));
    return check_launch(what);
}

int srcnn_stem_pack(const float *im_nchw, int B, int H, int W, float *out, int out_format, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(im_nchw && out && B > 0 && H > 0 && W > 0, "bad args");
    SRCNN_REQUIRE((unsigned)out_format <= 1, "bad format");
    return stem_pack_launch(im_nchw, im_nchw, B, B, H, W, out, out_format, stream, "srcnn_stem_pack");
}

int srcnn_stem_pack_pair(const float *left_nchw, const float *right_nchw, int B, int H, int W, float *out, int out_format,
                         srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(left_nchw && right_nchw && out && B > 0 && H > 0 && W > 0, "bad args");
    SRCNN_REQUIRE((unsigned)out_format <= 1, "bad format");
    return stem_pack_launch(left_nchw, right_nchw, B, 2 * B, H, W, out, out_format, stream, "srcnn_stem_pack_pair");
}

int srcnn_maxpool3x3s2_ceil(const float *x, int B, int H, int W, int C, float *y, int OH, int OW, int y_format,
                            srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
    SRCNN_REQUIRE((OH - 1) * 2 < H && (OW - 1) * 2 < W, "output too large for input");
    SRCNN_REQUIRE((unsigned)y_format <= 1, "bad format");
    const size_t total = (size_t)B * OH * OW * (C / 8);
    SRCNN_LAUNCH(maxpool3x3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, B, H, W, C,
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
    SRCNN_LAUNCH(upsample_add_kernel, dim3((W * (C / 8) + 255) / 256, B * H), dim3(256), 0, as_stream(stream), top, TH, TW,
                       lateral, B, H, W, C, y, top_format, y_format, range_flag_word());
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
    SRCNN_LAUNCH(act_convert_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, x_format, y,
                       y_format, (size_t)pixels, C, range_flag_word());
    return check_launch("srcnn_act_convert");
}

int srcnn_preprocess(const unsigned char *img_rgb, int H, int W, double scale, float *out_nchw, int OH, int OW,
                     float *packed, int packed_format, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(img_rgb && (out_nchw || packed) && H > 0 && W > 0 && OH > 0 && OW > 0 && scale > 0.0, "bad args");
    SRCNN_REQUIRE((unsigned)packed_format <= 1, "bad format");
    // cv::resize derives the output size itself (cvRound = nearest, ties to even); a caller passing another size is in error
    SRCNN_REQUIRE(OH == (int)nearbyint((double)H * scale) && OW == (int)nearbyint((double)W * scale),
                  "OH/OW must be cvRound(H*scale), cvRound(W*scale)");
    const size_t total = (size_t)(OH + 6) * ((OW + 8 + 1) / 2);
    SRCNN_LAUNCH(preprocess_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), img_rgb, H, W,
                       1.0 / scale, OH, OW, out_nchw, reinterpret_cast<char *>(packed), packed_format);
    return check_launch("srcnn_preprocess");
}

int srcnn_subsample2(const float *x, int B, int H, int W, int C, float *y, int OH, int OW, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(C % 4 == 0, "C must be a multiple of 4");
    SRCNN_REQUIRE((OH - 1) * 2 < H && (OW - 1) * 2 < W, "output too large for input");
    const size_t total = (size_t)B * OH * OW * (C / 4);
    SRCNN_LAUNCH(subsample2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), B, H, W, C / 4, reinterpret_cast<float4 *>(y), OH, OW);
    return check_launch("srcnn_subsample2");
}

int srcnn_nhwc_to_nchw(const float *x, int B, int H, int W, int C, float *y, srcnn_stream_t stream)
{
    using namespace srcnn;
    const int R = H * W, S = C;   // (B, HW, C) -> (B, C, HW)
    SRCNN_REQUIRE(B <= 65535, "batch too large");
    SRCNN_LAUNCH(transpose_kernel, dim3(cdiv(S, 32), cdiv(R, 32), B), dim3(32, 8), 0, as_stream(stream), x, R, S, y);
    return check_launch("srcnn_nhwc_to_nchw");
}

int srcnn_nchw_to_nhwc(const float *x, int B, int C, int H, int W, float *y, srcnn_stream_t stream)
{
    using namespace srcnn;
    const int R = C, S = H * W;   // (B, C, HW) -> (B, HW, C)
    SRCNN_REQUIRE(B <= 65535, "batch too large");
    SRCNN_LAUNCH(transpose_kernel, dim3(cdiv(S, 32), cdiv(R, 32), B), dim3(32, 8), 0, as_stream(stream), x, R, S, y);
    return check_launch("srcnn_nchw_to_nhwc");
}

}  // extern "C"
