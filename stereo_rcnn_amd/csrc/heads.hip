// Head tails and detection post-processing (gfx950): class softmax, keypoint tail,
// detection decode and per-class filter + sort + NMS.  All tiny, all on the device so that
// the forward pass and the decode never bounce through the host.
//
// Reference: stereo_rcnn/stereo_rcnn.py:256-271 (softmaxes, sum over H),
// demo.py:144-218 (de-interleave, de-normalise, decode, clip, /scale),
// rpn/bbox_transform.py:133-155 (keypoint / border decode), demo.py:231-257 (per-class NMS).
#include "conv_common.h"

namespace srcnn {

__global__ void softmax_rows_kernel(const float *__restrict__ x, int rows, int cols, int xs, float *__restrict__ y)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = x + (size_t)r * xs;
    float m = p[0];
    for (int c = 1; c < cols; ++c) m = fmaxf(m, p[c]);
    float s = 0.f;
    for (int c = 0; c < cols; ++c) s += expf(p[c] - m);
    for (int c = 0; c < cols; ++c) y[(size_t)r * cols + c] = expf(p[c] - m) / s;
}

// Tail of the box head in one launch: the stacked fc output rows [bbox n_bbox | dim_orien n_dim | cls logits n_cls] are split into
// the three tensors the forward returns -- bbox_pred and dim_orien_pred copied bit for bit into contiguous rows, the class
// logits through the softmax of softmax_rows_kernel (same expression order).
__global__ void box_head_tail_kernel(const float *__restrict__ fc, int rows, int n_bbox, int n_dim, int n_cls, int xs,
                                     float *__restrict__ bbox, float *__restrict__ dim, float *__restrict__ cls)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = fc + (size_t)r * xs;
    for (int c = 0; c < n_bbox; ++c) bbox[(size_t)r * n_bbox + c] = p[c];
    for (int c = 0; c < n_dim; ++c) dim[(size_t)r * n_dim + c] = p[n_bbox + c];
    p += n_bbox + n_dim;
    float m = p[0];
    for (int c = 1; c < n_cls; ++c) m = fmaxf(m, p[c]);
    float s = 0.f;
    for (int c = 0; c < n_cls; ++c) s += expf(p[c] - m);
    for (int c = 0; c < n_cls; ++c) cls[(size_t)r * n_cls + c] = expf(p[c] - m) / s;
}

// one block per roi; logits (n, G, G, 6) NHWC.  G <= 32.
__global__ __launch_bounds__(256) void kpts_tail_kernel(const float *__restrict__ logits, int G,
                                                        float *__restrict__ kpts_prob, float *__restrict__ left_prob,
                                                        float *__restrict__ right_prob, const int *__restrict__ roi_limit)
{
    __shared__ float col[6][32];
    __shared__ float red[3][2];   // {max, sum} for kpts / left / right
    const int n = blockIdx.x, tid = threadIdx.x;
    if (roi_limit && n >= *roi_limit) return;
    const float *lg = logits + (size_t)n * G * G * 6;
    if (tid < G * 6) {
        const int w = tid / 6, ch = tid - w * 6;
        float s = 0.f;
        for (int h = 0; h < G; ++h) s += lg[((size_t)h * G + w) * 6 + ch];   // .sum(2), stereo_rcnn.py:263
        col[ch][w] = s;
    }
    __syncthreads();
    if (tid < 3) {   // group 0: channels 0..3 (4G bins); 1: channel 4; 2: channel 5
        const int c0 = tid == 0 ? 0 : 3 + tid, c1 = tid == 0 ? 4 : 4 + tid;
        float m = -INFINITY;
        for (int c = c0; c < c1; ++c)
            for (int w = 0; w < G; ++w) m = fmaxf(m, col[c][w]);
        float s = 0.f;
        for (int c = c0; c < c1; ++c)
            for (int w = 0; w < G; ++w) s += expf(col[c][w] - m);
        red[tid][0] = m;
        red[tid][1] = s;
    }
    __syncthreads();
    if (tid < G * 6) {
        const int ch = tid / G, w = tid - ch * G;
        const int grp = ch < 4 ? 0 : ch - 3;
        const float v = expf(col[ch][w] - red[grp][0]) / red[grp][1];
        if (ch < 4) kpts_prob[(size_t)n * 4 * G + ch * G + w] = v;
        else if (ch == 4) left_prob[(size_t)n * G + w] = v;
        else right_prob[(size_t)n * G + w] = v;
    }
}

__device__ __forceinline__ void decode_clip(const float *roi, float dx, float dy, float dw, float dh, float wmax,
                                            float hmax, float inv_scale_div, float *o)
{
    const float widths = roi[2] - roi[0] + 1.0f;
    const float heights = roi[3] - roi[1] + 1.0f;
    const float ctr_x = roi[0] + 0.5f * widths;
    const float ctr_y = roi[1] + 0.5f * heights;
    const float pcx = dx * widths + ctr_x;
    const float pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths;
    const float ph = expf(dh) * heights;
    o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.f), wmax) / inv_scale_div;   // clip (demo.py:203-204) then /scale (:206-207)
    o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.f), hmax) / inv_scale_div;
    o[2] = fminf(fmaxf(pcx + 0.5f * pw, 0.f), wmax) / inv_scale_div;
    o[3] = fminf(fmaxf(pcy + 0.5f * ph, 0.f), hmax) / inv_scale_div;
}

__device__ __forceinline__ int argmax_first(const float *p, int n, float *vmax)
{
    int best = 0;
    float m = p[0];
    for (int i = 1; i < n; ++i)
        if (p[i] > m) { m = p[i]; best = i; }
    if (vmax) *vmax = m;
    return best;
}

__global__ void decode_detections_kernel(const float *__restrict__ rois_l, const float *__restrict__ rois_r,
                                         const float *__restrict__ bbox_pred, const float *__restrict__ dim_pred,
                                         const float *__restrict__ kpts_prob, const float *__restrict__ left_prob,
                                         const float *__restrict__ right_prob, const float *__restrict__ im_info,
                                         int n, int ncls, int G, float *__restrict__ boxes_l,
                                         float *__restrict__ boxes_r, float *__restrict__ dim_out,
                                         float *__restrict__ kpts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float hmax = im_info[0] - 1.0f, wmax = im_info[1] - 1.0f, scale = im_info[2];
    const float *rl = rois_l + (size_t)i * 5 + 1, *rr = rois_r + (size_t)i * 5 + 1;
    const float std4[4] = {0.1f, 0.1f, 0.2f, 0.2f};                 // BBOX_NORMALIZE_STDS (config.py:78); means are 0 (:77)
    const float dmean[5] = {1.6f, 1.5f, 4.0f, 0.0f, 0.0f};          // DIM_NORMALIZE_MEANS (config.py:81)
    for (int j = 0; j < ncls; ++j) {
        const float *d = bbox_pred + ((size_t)i * ncls + j) * 6;     // demo.py:152-161
        const float lx = d[0] * std4[0] + 0.0f, ly = d[1] * std4[1] + 0.0f;
        const float lw = d[2] * std4[2] + 0.0f, lh = d[3] * std4[3] + 0.0f;
        const float rx = d[4] * std4[0] + 0.0f, rw = d[5] * std4[2] + 0.0f;
        decode_clip(rl, lx, ly, lw, lh, wmax, hmax, scale, boxes_l + ((size_t)i * ncls + j) * 4);
        decode_clip(rr, rx, ly, rw, lh, wmax, hmax, scale, boxes_r + ((size_t)i * ncls + j) * 4);
        const float *q = dim_pred + ((size_t)i * ncls + j) * 5;
#pragma unroll
        for (int c = 0; c < 5; ++c) dim_out[((size_t)i * ncls + j) * 5 + c] = q[c] * 0.5f + dmean[c];   // demo.py:185-186
    }
    float pmax;
    const int kd = argmax_first(kpts_prob + (size_t)i * 4 * G, 4 * G, &pmax);
    const int ld = argmax_first(left_prob + (size_t)i * G, G, nullptr);
    const int rd = argmax_first(right_prob + (size_t)i * G, G, nullptr);
    const float widths = rl[2] - rl[0] + 1.0f;    // proposal width (bbox_transform.py:135,149)
    const float g = (float)G;
    const float dk = (float)kd;
    float *o = kpts + (size_t)i * 5;
    o[0] = ((float)(kd % G) * widths / g + rl[0]) / scale;   // :141 then demo.py:208
    o[1] = dk / g;                                           // kpts_type stays a float (:139)
    o[2] = pmax;
    o[3] = ((float)ld * widths / g + rl[0]) / scale;
    o[4] = ((float)rd * widths / g + rl[0]) / scale;
}

// ------------------------------------------------------------------ keypoint head on the kept detections only
__global__ void gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ idx, int n_idx, int cols,
                                   float *__restrict__ dst)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_idx * cols) return;
    const int r = t / cols, c = t - r * cols;
    const int i = max(idx[r], 0);                 // -1 padded keep lists: repeat row 0 (never read back)
    dst[t] = src[(size_t)i * cols + c];
}

// the keypoint part of decode_detections_kernel for the rows of a keep list: probabilities in KEPT order (row r belongs to
// roi keep_idx[r]), result written to the roi's own row of `kpts`
__global__ void decode_kept_kpts_kernel(const float *__restrict__ rois_l, const float *__restrict__ kpts_prob,
                                        const float *__restrict__ left_prob, const float *__restrict__ right_prob,
                                        const int *__restrict__ keep_idx, const int *__restrict__ num,
                                        const float *__restrict__ im_info, int n, int G, float *__restrict__ kpts)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || r >= *num) return;
    const int i = keep_idx[r];
    if (i < 0) return;
    const float scale = im_info[2];
    const float *rl = rois_l + (size_t)i * 5 + 1;
    float pmax;
    const int kd = argmax_first(kpts_prob + (size_t)r * 4 * G, 4 * G, &pmax);
    const int ld = argmax_first(left_prob + (size_t)r * G, G, nullptr);
    const int rd = argmax_first(right_prob + (size_t)r * G, G, nullptr);
    const float widths = rl[2] - rl[0] + 1.0f;    // proposal width (bbox_transform.py:135,149)
    const float g = (float)G;
    const float dk = (float)kd;
    float *o = kpts + (size_t)i * 5;
    o[0] = ((float)(kd % G) * widths / g + rl[0]) / scale;   // :141 then demo.py:208
    o[1] = dk / g;                                           // kpts_type stays a float (:139)
    o[2] = pmax;
    o[3] = ((float)ld * widths / g + rl[0]) / scale;
    o[4] = ((float)rd * widths / g + rl[0]) / scale;
}

// ------------------------------------------------------------------ per-class filter + sort
__device__ __forceinline__ unsigned score_key32(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// single block; n <= 2048.  Writes sorted original indices, the (m,5) dets and m.
__global__ __launch_bounds__(1024) void class_select_sort_kernel(const float *__restrict__ scores, int n, int ncls,
                                                                int j, const float *__restrict__ boxes, float thresh,
                                                                int *__restrict__ sorted_idx, float *__restrict__ dets,
                                                                int *__restrict__ count)
{
    __shared__ unsigned long long cand[2048];
    __shared__ unsigned s_cnt;
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < 2048; i += 1024) cand[i] = 0ULL;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const float s = scores[(size_t)i * ncls + j];
        if (s > thresh) {   // demo.py:232
            const unsigned pos = atomicAdd(&s_cnt, 1u);
            cand[pos] = ((unsigned long long)score_key32(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    const int m = (int)s_cnt;
    int np = 1;
    while (np < m) np <<= 1;
    for (int k = 2; k <= np; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < np; i += 1024) {
                const int ixj = i ^ jj;
                if (ixj > i) {
                    const unsigned long long a = cand[i], c = cand[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < c) : (a > c)) { cand[i] = c; cand[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int r = tid; r < n; r += 1024) {
        float *d = dets + (size_t)r * 5;
        if (r < m) {
            const int i = (int)(0xFFFFFFFFu - (unsigned)(cand[r] & 0xFFFFFFFFULL));
            sorted_idx[r] = i;
            const float *bx = boxes + ((size_t)i * ncls + j) * 4;
            d[0] = bx[0]; d[1] = bx[1]; d[2] = bx[2]; d[3] = bx[3];
            d[4] = scores[(size_t)i * ncls + j];
        } else {
            sorted_idx[r] = -1;
            d[0] = d[1] = d[2] = d[3] = d[4] = 0.f;
        }
    }
    if (tid == 0) *count = m;
}

__global__ void map_keep_kernel(const int *__restrict__ keep, const int *__restrict__ num,
                                const int *__restrict__ sorted_idx, int n, int *__restrict__ keep_idx,
                                int *__restrict__ num_keep)
{
    const int k = *num;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x)
        keep_idx[i] = i < k ? sorted_idx[keep[i]] : -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *num_keep = k;
}

// fixed-size detection record for the multi-GPU gather: row 0 = [count, 0...], row 1+r = r-th kept detection
// [score, left box 4, right box 4, dim_orien 5, kpts 5, roi index, 0 0 0 0]; rows past the count are zero.
__global__ void pack_detections_kernel(const float *__restrict__ scores, const float *__restrict__ boxes_l,
                                       const float *__restrict__ boxes_r, const float *__restrict__ dim_orien,
                                       const float *__restrict__ kpts, const int *__restrict__ keep_idx,
                                       const int *__restrict__ num, int n, int ncls, int j, int cols,
                                       float *__restrict__ rec, unsigned *__restrict__ range_flag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;     // record row, 0..n
    if (r > n) return;
    float *o = rec + (size_t)r * cols;
    for (int c = 0; c < cols; ++c) o[c] = 0.f;
    const int k = *num;
    if (r == 0) {
        o[0] = (float)k;
        o[1] = (float)*range_flag;           // SPLIT16 range guard of the forward that produced these detections (0 = clean)
        *range_flag = 0;                     // consumed: the flag travels in the record, the next forward starts clean
        return;
    }
    if (r - 1 >= k) return;
    const int i = keep_idx[r - 1];
    o[0] = scores[(size_t)i * ncls + j];
    for (int c = 0; c < 4; ++c) {
        o[1 + c] = boxes_l[((size_t)i * ncls + j) * 4 + c];
        o[5 + c] = boxes_r[((size_t)i * ncls + j) * 4 + c];
    }
    for (int c = 0; c < 5; ++c) {
        o[9 + c] = dim_orien[((size_t)i * ncls + j) * 5 + c];
        o[14 + c] = kpts[(size_t)i * 5 + c];
    }
    o[19] = (float)i;
}

struct ClassNmsLayout {
    size_t sorted_idx, dets, keep, num, count, nms, total;
};

static ClassNmsLayout class_nms_layout(int n)
{
    ClassNmsLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    L.sorted_idx = take((size_t)n * sizeof(int));
    L.dets = take((size_t)n * 5 * sizeof(float));
    L.keep = take((size_t)n * sizeof(int));
    L.num = take(sizeof(int));
    L.count = take(sizeof(int));
    L.nms = take(srcnn_nms_workspace_bytes(n));
    L.total = off;
    return L;
}

}  // namespace srcnn

extern "C" {

int srcnn_softmax_rows(const float *x, int rows, int cols, int x_stride, float *y, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(x && y && rows >= 0 && cols > 0 && x_stride >= cols, "bad args");
    if (rows == 0) return SRCNN_OK;
    SRCNN_LAUNCH(softmax_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, as_stream(stream), x, rows, cols,
                       x_stride, y);
    return check_launch("srcnn_softmax_rows");
}

int srcnn_box_head_tail(const float *fc, int rows, int n_bbox, int n_dim, int n_cls, int fc_stride, float *bbox_pred,
                        float *dim_orien_pred, float *cls_prob, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(fc && bbox_pred && dim_orien_pred && cls_prob && rows >= 0 && n_bbox > 0 && n_dim > 0 && n_cls > 0 &&
                      fc_stride >= n_bbox + n_dim + n_cls, "bad args");
    if (rows == 0) return SRCNN_OK;
    SRCNN_LAUNCH(box_head_tail_kernel, dim3(cdiv(rows, 128)), dim3(128), 0, as_stream(stream), fc, rows, n_bbox, n_dim, n_cls,
                       fc_stride, bbox_pred, dim_orien_pred, cls_prob);
    return check_launch("srcnn_box_head_tail");
}

int srcnn_kpts_tail(const float *logits, int n, int G, float *kpts_prob, float *left_prob, float *right_prob,
                    const int *roi_limit, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(logits && kpts_prob && left_prob && right_prob && G > 0 && G <= 32, "bad args");
    if (n == 0) return SRCNN_OK;
    SRCNN_LAUNCH(kpts_tail_kernel, dim3(n), dim3(256), 0, as_stream(stream), logits, G, kpts_prob, left_prob,
                       right_prob, roi_limit);
    return check_launch("srcnn_kpts_tail");
}

int srcnn_gather_rows(const float *src, const int *idx, int n_idx, int cols, float *dst, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(src && idx && dst && n_idx >= 0 && cols > 0, "bad args");
    if (n_idx == 0) return SRCNN_OK;
    SRCNN_LAUNCH(gather_rows_kernel, dim3(cdiv(n_idx * cols, 256)), dim3(256), 0, as_stream(stream), src, idx, n_idx, cols, dst);
    return check_launch("srcnn_gather_rows");
}

int srcnn_decode_kept_kpts(const float *rois_left, const float *kpts_prob, const float *left_prob, const float *right_prob,
                           const int *keep_idx, const int *num_keep, const float *im_info, int n, int G, float *kpts,
                           srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rois_left && kpts_prob && left_prob && right_prob && keep_idx && num_keep && im_info && kpts && G > 0 && G <= 32,
                  "bad args");
    if (n == 0) return SRCNN_OK;
    SRCNN_LAUNCH(decode_kept_kpts_kernel, dim3(cdiv(n, 128)), dim3(128), 0, as_stream(stream), rois_left, kpts_prob, left_prob,
                 right_prob, keep_idx, num_keep, im_info, n, G, kpts);
    return check_launch("srcnn_decode_kept_kpts");
}

int srcnn_decode_detections(const float *rois_left, const float *rois_right, const float *bbox_pred,
                            const float *dim_orien_pred, const float *kpts_prob, const float *left_prob,
                            const float *right_prob, const float *im_info, int n, int n_cls, int G,
                            float *boxes_left, float *boxes_right, float *dim_orien, float *kpts,
                            srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(rois_left && rois_right && bbox_pred && dim_orien_pred && kpts_prob && left_prob && right_prob &&
                      im_info && boxes_left && boxes_right && dim_orien && kpts, "null pointer");
    if (n == 0) return SRCNN_OK;
    SRCNN_LAUNCH(decode_detections_kernel, dim3(cdiv(n, 128)), dim3(128), 0, as_stream(stream), rois_left,
                       rois_right, bbox_pred, dim_orien_pred, kpts_prob, left_prob, right_prob, im_info, n, n_cls, G,
                       boxes_left, boxes_right, dim_orien, kpts);
    return check_launch("srcnn_decode_detections");
}

int srcnn_pack_detections(const float *scores, const float *boxes_left, const float *boxes_right,
                          const float *dim_orien, const float *kpts, const int *keep_idx, const int *num_keep, int n,
                          int n_cls, int j, int rec_cols, float *rec, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(scores && boxes_left && boxes_right && dim_orien && kpts && keep_idx && num_keep && rec, "null pointer");
    SRCNN_REQUIRE(rec_cols >= 20 && n > 0 && j >= 0 && j < n_cls, "bad sizes (rec_cols >= 20)");
    SRCNN_LAUNCH(pack_detections_kernel, dim3(cdiv(n + 1, 128)), dim3(128), 0, as_stream(stream), scores,
                       boxes_left, boxes_right, dim_orien, kpts, keep_idx, num_keep, n, n_cls, j, rec_cols, rec,
                       range_flag_word());
    return check_launch("srcnn_pack_detections");
}

size_t srcnn_class_nms_workspace_bytes(int n) { return srcnn::class_nms_layout(n > 0 ? n : 1).total; }

int srcnn_class_nms(const float *scores, int n, int n_cls, int j, const float *boxes_left, float score_thresh,
                    float nms_thresh, int *keep_idx, int *num_keep, void *workspace, size_t workspace_bytes,
                    srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(scores && boxes_left && keep_idx && num_keep, "null pointer");
    SRCNN_REQUIRE(n > 0 && n <= 2048 && j >= 0 && j < n_cls, "bad sizes (n <= 2048)");
    ClassNmsLayout L = class_nms_layout(n);
    if (!workspace || workspace_bytes < L.total) {
        set_error("srcnn_class_nms: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return SRCNN_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(workspace);
    int *sorted_idx = reinterpret_cast<int *>(ws + L.sorted_idx);
    float *dets = reinterpret_cast<float *>(ws + L.dets);
    int *keep = reinterpret_cast<int *>(ws + L.keep);
    int *num = reinterpret_cast<int *>(ws + L.num);
    int *count = reinterpret_cast<int *>(ws + L.count);
    hipStream_t st = as_stream(stream);
    SRCNN_LAUNCH(class_select_sort_kernel, dim3(1), dim3(1024), 0, st, scores, n, n_cls, j, boxes_left,
                       score_thresh, sorted_idx, dets, count);
    int rc = check_launch("class_nms: select");
    if (rc != SRCNN_OK) return rc;
    rc = srcnn_nms_batched(keep, dets, num, count, 1, n, 5, nms_thresh, ws + L.nms, workspace_bytes - L.nms, stream);
    if (rc != SRCNN_OK) return rc;
    SRCNN_LAUNCH(map_keep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, keep, num, sorted_idx, n, keep_idx,
                       num_keep);
    return check_launch("class_nms: map");
}

}  // extern "C"
