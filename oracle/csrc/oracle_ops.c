/*
 * oracle_ops.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, plain C).
 *
 * Scalar restatement of the two native operators of the reference's hot path,
 * written so that every float32 operation happens in the order the reference's
 * device code performs it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (stereo_rcnn_amd/) never does.
 *
 * Follows (reference paths relative to /root/reference):
 *   - IoU predicate     lib/model/nms/src/nms_cuda_kernel.cu:31-39  (devIoU)
 *   - suppression rule  lib/model/nms/src/nms_cuda_kernel.cu:74-79  (j > i, strict >)
 *   - greedy reduction  lib/model/nms/src/nms_cuda_kernel.cu:131-144
 *   - ROIAlign lattice  lib/model/roi_align/src/roi_align_kernel.cu:27-68
 *   - 2x2/s1 avg-pool   lib/model/roi_align/modules/roi_align.py:26-29
 *
 * Parity status: PINNED.  The reference's own .cu files compile unchanged with hipcc for gfx950 (oracle/build.py:build_ref,
 * oracle/ref_cuda_on_hip.h) and run on the MI355X: their keep lists and ROIAlign outputs are bit-identical to this file
 * (tests/test_ref_kernels_gpu.py).  That check found and fixed a promotion error of the first restatement (the two lower
 * taps of the bilinear blend start as float x float products, see below).  Also pinned against an independent
 * pure-Python loop restatement (oracle/ops.py) and committed golden vectors (tests/golden/).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* nms_cuda_kernel.cu:31-39 -- "+1" pixel convention, plain float32 ops. */
static float iou_plus1(const float *a, const float *b)
{
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1.0f, 0.0f);
    float height = fmaxf(bottom - top + 1.0f, 0.0f);
    float inter = width * height;
    float sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
    float sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
    return inter / (sa + sb - inter);
}

/*
 * Greedy NMS over boxes that are ALREADY sorted by descending score (the
 * reference kernel never looks at column 4).  dets: n x dim floats (dim >= 4).
 * keep: n ints (first *num_out valid).  Equivalent to building the 64-bit
 * suppression masks (kernel) and OR-reducing them in index order (host loop).
 */
int oracle_nms(const float *dets, int n, int dim, float thresh, int *keep, int *num_out)
{
    unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int k = 0;
    if (!removed) return -1;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep[k++] = i;
        const float *bi = dets + (size_t)i * dim;
        for (int j = i + 1; j < n; ++j) {
            if (removed[j]) continue; /* masks of removed boxes are never OR-ed in; */
            /* skipping already-removed j is output-equivalent */
            if (iou_plus1(bi, dets + (size_t)j * dim) > thresh) removed[j] = 1;
        }
    }
    *num_out = k;
    free(removed);
    return 0;
}

/* Full 64-bit mask matrix exactly as nms_kernel writes it (for mask-level tests). */
int oracle_nms_mask(const float *dets, int n, int dim, float thresh, uint64_t *mask)
{
    int col_blocks = (n + 63) / 64;
    for (int i = 0; i < n; ++i) {
        for (int cb = 0; cb < col_blocks; ++cb) {
            uint64_t t = 0;
            int col_size = n - cb * 64 < 64 ? n - cb * 64 : 64;
            int start = (i / 64 == cb) ? (i % 64) + 1 : 0;
            for (int c = start; c < col_size; ++c) {
                if (iou_plus1(dets + (size_t)i * dim, dets + (size_t)(cb * 64 + c) * dim) > thresh)
                    t |= 1ULL << c;
            }
            mask[(size_t)i * col_blocks + cb] = t;
        }
    }
    return 0;
}

/*
 * Legacy point-lattice ROIAlign forward, NCHW features, output (n, C, ah, aw).
 * roi_align_kernel.cu:27-68.  The reference mixes float and double: literals
 * such as `1.` and `0.` promote parts of the expressions to double; those
 * promotions are reproduced one by one below.
 */
int oracle_roi_align_forward(int ah, int aw, float spatial_scale,
                             const float *feat, int batch, int channels, int height, int width,
                             const float *rois, int num_rois, float *out)
{
    (void)batch;
    for (int n = 0; n < num_rois; ++n) {
        const float *r = rois + (size_t)n * 5;
        float roi_batch_ind = r[0];
        float roi_start_w = r[1] * spatial_scale;
        float roi_start_h = r[2] * spatial_scale;
        float roi_end_w = r[3] * spatial_scale;
        float roi_end_h = r[4] * spatial_scale;
        /* fmaxf(float - float + 1. , 0.): sum in double, narrowed to float by fmaxf */
        float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.0f);
        float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.0f);
        /* float / (int - 1.) : double division, narrowed on assignment */
        float bin_size_h = (float)((double)roi_height / ((double)ah - 1.));
        float bin_size_w = (float)((double)roi_width / ((double)aw - 1.));
        int img_start = (int)(roi_batch_ind * (float)channels * (float)height * (float)width);
        /* reference: int = float*int*int*int evaluated left to right in float */
        for (int c = 0; c < channels; ++c) {
            for (int ph = 0; ph < ah; ++ph) {
                for (int pw = 0; pw < aw; ++pw) {
                    float h = (float)ph * bin_size_h + roi_start_h;
                    float w = (float)pw * bin_size_w + roi_start_w;
                    int hstart = (int)fminf(floorf(h), (float)(height - 2));
                    int wstart = (int)fminf(floorf(w), (float)(width - 2));
                    size_t o = (((size_t)n * channels + c) * ah + ph) * aw + pw;
                    if (h < 0 || h >= height || w < 0 || w >= width) {
                        out[o] = 0.0f;
                    } else {
                        float h_ratio = h - (float)hstart;
                        float w_ratio = w - (float)wstart;
                        int upleft = img_start + (c * height + hstart) * width + wstart;
                        int upright = upleft + 1;
                        int downleft = upleft + width;
                        int downright = downleft + 1;
                        /* roi_align_kernel.cu:64-67, usual arithmetic conversions left to right: terms 1 and 2 meet the
                         * double `(1. - h_ratio)` first and are double products; `feat * h_ratio` is float x float (one
                         * float rounding) before `(1. - w_ratio)` promotes it; term 4 is a float product throughout.
                         * (Pinned by running the reference's own kernel: tests/test_ref_kernels_gpu.py.) */
                        float dl_h = feat[downleft] * h_ratio;
                        float dr_h = feat[downright] * h_ratio;
                        float dr_hw = dr_h * w_ratio;
                        double v = (double)feat[upleft] * (1. - (double)h_ratio) * (1. - (double)w_ratio)
                                 + (double)feat[upright] * (1. - (double)h_ratio) * (double)w_ratio
                                 + (double)dl_h * (1. - (double)w_ratio)
                                 + (double)dr_hw;
                        out[o] = (float)v;
                    }
                }
            }
        }
    }
    return 1;
}

/* avg_pool2d(kernel=2, stride=1) over (n*C) planes of (ah, aw) -> (ah-1, aw-1).
 * Sum order row-major inside the window, then * 0.25 (== / 4 exactly). */
int oracle_avgpool2x2_s1(const float *in, int planes, int ah, int aw, float *out)
{
    int oh = ah - 1, ow = aw - 1;
    for (int p = 0; p < planes; ++p) {
        const float *src = in + (size_t)p * ah * aw;
        float *dst = out + (size_t)p * oh * ow;
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                float s = src[y * aw + x];
                s = s + src[y * aw + x + 1];
                s = s + src[(y + 1) * aw + x];
                s = s + src[(y + 1) * aw + x + 1];
                dst[y * ow + x] = s * 0.25f;
            }
    }
    return 0;
}
