// Dense photometric alignment on the device (gfx950).
//
// Replaces lib/model/dense_align/dense_align.py:13-69 (sample), :175-238 (enumeration_depth),
// :240-300 (align_parallel) and box_3d.py:12-106 (Box3d, BoxRayInsec).  The reference builds
// the sample lattice with a per-ROI Python loop of dozens of tiny GPU ops with implicit syncs
// (`int(tensor)`), materialises (1,3,50R,P) tensors and re-samples the hypothesis-independent
// left image 70 times.  Here:
//   upsample2x_kernel   both images, bilinear align_corners=True (F.upsample of torch 0.3)
//   sample_kernel       one workgroup per object: ray / 3-nearest-faces intersection per lattice
//                       pixel, in-box mask, ORDER-PRESERVING compaction (wave ballots + prefix)
//   left_sample_kernel  the left taps, once per valid pixel
//   make_enum_kernel    depth hypotheses (50 coarse, then 20 fine around the coarse optimum)
//   cost_kernel         one workgroup per (object, hypothesis): per-pixel disparity, right taps
//                       (border clamp), SAD (L1!, dense_align.py:231), wave-shuffle reduction
//   argmin_kernel       first minimum, final disparity / status
// float32 arithmetic follows the reference's operation order (the library is built with
// -ffp-contract=off); Python-double scalars of the reference (f, bl, cx, cy, slice arithmetic
// on torch-0.3 scalar-indexed values) are doubles here.
#include "common.h"
#include <cstdlib>

namespace srcnn {

struct DaCalib {
    double scale2, f, cx, cy, bl, fb;   // all already multiplied by scale2 where the reference does
    int H2, W2;                         // upsampled image size
};

// ------------------------------------------------------------------ 2x bilinear upsample (align_corners)
// The element's arithmetic is F.upsample's, operation by operation; the taps come from the 29 MB source through the caches.
// Round 5's form (one float per thread, flat index decomposed with 64-bit % and /) moved its 143 MB at 1.6 TB/s.
// A flat-index form with four floats per thread and 32-bit unsigned divisions on the vector unit was bit-exact alone and
// NOT repeatable beside other forwards' MFMA kernels: 13-41 of 306 frames of tests/test_pipeline_gpu.py's three-in-flight soak
// came back with a neighbouring depth hypothesis for one or two objects (same box, same day: 0 of 306 with round 5's kernel,
// 0 of 306 with the division-free form below) -- the second kernel of this file, after sample_kernel's box geometry (see
// there), whose v_rcp-based sequences misbehave under that co-residency.  Cause not established; the soak test is the gate.
__device__ __forceinline__ float upsample2x_at(const float *__restrict__ src, int H, int W, int x, int y, int c, float rh, float rw)
{
    const float h1r = rh * (float)y;
    const int h1 = (int)h1r;
    const int h1p = h1 < H - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float w1r = rw * (float)x;
    const int w1 = (int)w1r;
    const int w1p = w1 < W - 1 ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float *p = src + ((size_t)c * H + h1) * W + w1;
    return h0l * (w0l * p[0] + w1l * p[w1p]) + h1l * (w0l * p[(size_t)h1p * W] + w1l * p[(size_t)h1p * W + w1p]);
}

// grid (ceil(W / 256), ceil(2H / UP_ROWS), 6): a thread owns one output column pair (one 8-byte store per row: 2W is even, every
// row starts 8-byte aligned) of UP_ROWS consecutive output rows -- its column taps and weights are computed once; z = image * 3 +
// plane.  No integer division on the vector unit.  (One row per workgroup was 57 600 workgroups of 0.5 us of work each:
// dispatch-bound, 74 us.)
constexpr int UP_ROWS = 8;
__global__ __launch_bounds__(256) void upsample2x_kernel(const float *__restrict__ a, const float *__restrict__ b, int H, int W,
                                                         float *__restrict__ oa, float *__restrict__ ob)
{
    const int H2 = 2 * H, W2 = 2 * W;
    const float rh = (float)(H - 1) / (float)(H2 - 1), rw = (float)(W - 1) / (float)(W2 - 1);
    const int img = blockIdx.z / 3, c = blockIdx.z - 3 * img;
    const float *src = img ? b : a;
    float *dst = (img ? ob : oa) + (size_t)c * H2 * W2;
    const int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= W2) return;
    const int y0 = blockIdx.y * UP_ROWS, y1 = min(y0 + UP_ROWS, H2);
    for (int y = y0; y < y1; ++y) {
        const float v0 = upsample2x_at(src, H, W, x, y, c, rh, rw), v1 = upsample2x_at(src, H, W, x + 1, y, c, rh, rw);
        *reinterpret_cast<float2 *>(dst + (size_t)y * W2 + x) = make_float2(v0, v1);
    }
}

// ------------------------------------------------------------------ sample lattice + ray/box intersection
struct BoxGeom {
    float T[3], R[3][3], lo[3], hi[3];
    float planes[3][4];
};

// Pc: the 8 vertices in camera coordinates live in LDS (`pc`, written identically by every calling lane): they are read back by
// data-dependent index, and a dynamically indexed REGISTER array would be lowered to GPR-index mode
__device__ void build_box(const float *pose, BoxGeom &g, float (*Pc)[3])
{
    const double sx = (double)pose[3], sy = (double)pose[4], sz = (double)pose[5], th = (double)pose[6];
    const float c = (float)cos(th), s = (float)sin(th);
    g.T[0] = pose[0]; g.T[1] = pose[1]; g.T[2] = pose[2];
    g.R[0][0] = c;  g.R[0][1] = 0.f; g.R[0][2] = s;
    g.R[1][0] = 0.f; g.R[1][1] = 1.f; g.R[1][2] = 0.f;
    g.R[2][0] = -s; g.R[2][1] = 0.f; g.R[2][2] = c;
    const float hx = (float)(sx / 2), hz = (float)(sz / 2.0), hy = (float)sy;
    const float Po[8][3] = {{-hx, 0, -hz}, {-hx, 0, hz}, {hx, 0, hz}, {hx, 0, -hz},
                            {-hx, -hy, -hz}, {-hx, -hy, hz}, {hx, -hy, hz}, {hx, -hy, -hz}};   // box_3d.py:21-29
    int nearest = 0;
    float best = 100000000.f;
    for (int i = 0; i < 8; ++i) {
        for (int r = 0; r < 3; ++r)
            Pc[i][r] = (g.R[r][0] * Po[i][0] + g.R[r][1] * Po[i][1] + g.R[r][2] * Po[i][2]) + g.T[r];
        const float d = sqrtf(Pc[i][0] * Pc[i][0] + Pc[i][1] * Pc[i][1] + Pc[i][2] * Pc[i][2]);
        if (d < best) { best = d; nearest = i; }   // strict <: first nearest vertex (box_3d.py:55-60)
    }
    // DOUBLE_EPS slack, thresholds narrowed to float32 like a tensor-vs-python-scalar compare
    for (int k = 0; k < 3; ++k) {
        g.lo[k] = (float)((double)Po[4][k] - 0.01);
        g.hi[k] = (float)((double)Po[2][k] + 0.01);
    }
    const int tri[6][3] = {{0, 3, 4}, {2, 3, 6}, {1, 2, 5}, {0, 1, 4}, {0, 1, 2}, {4, 5, 6}};   // box_3d.py:47-52
    const int group[8][3] = {{0, 3, 4}, {2, 3, 4}, {1, 2, 4}, {0, 1, 4}, {0, 3, 5}, {2, 3, 5}, {1, 2, 5}, {0, 1, 5}};
    for (int i = 0; i < 3; ++i) {
        const int pl = group[nearest][i];
        const float *p1 = Pc[tri[pl][0]], *p2 = Pc[tri[pl][1]], *p3 = Pc[tri[pl][2]];
        const float a1[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
        const float a2[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
        const float n0 = a1[1] * a2[2] - a1[2] * a2[1];
        const float n1 = a1[2] * a2[0] - a1[0] * a2[2];
        const float n2 = a1[0] * a2[1] - a1[1] * a2[0];
        g.planes[i][0] = n0; g.planes[i][1] = n1; g.planes[i][2] = n2;
        g.planes[i][3] = ((-n0 * p1[0]) - n1 * p1[1]) - n2 * p1[2];   // box_3d.py:43
    }
}

// one workgroup (256 threads) per object
__global__ __launch_bounds__(256) void sample_kernel(const float *__restrict__ boxes, const float *__restrict__ borders,
                                                     const float *__restrict__ poses, const float *__restrict__ valid,
                                                     DaCalib cal, int max_pixels,
                                                     float *__restrict__ uvz, int *__restrict__ cnt)
{
    __shared__ int wave_cnt[4];
    __shared__ int s_base;
    __shared__ BoxGeom s_g;
    __shared__ float s_pc[8][3];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (valid && !(valid[r] > 0.f)) {            // masked row of a fixed-size batch: an object with no sample (status 0)
        if (tid == 0) { cnt[r] = 0; cnt[gridDim.x + r] = 0; }
        return;
    }
    const float sc = (float)cal.scale2;
    // box_left * scale, keypoints * scale: float32 tensor * python float (dense_align.py:261-262)
    const double b0 = (double)(boxes[r * 4 + 0] * sc), b1 = (double)(boxes[r * 4 + 1] * sc);
    const double b3 = (double)(boxes[r * 4 + 3] * sc);
    (void)b0;
    const double bl = (double)(borders[r * 2 + 0] * sc), br = (double)(borders[r * 2 + 1] * sc);
    // torch-0.3 scalar indexing -> Python doubles -> int() truncation (dense_align.py:42-45)
    int wstep = (int)((br - bl) / 56.0); if (wstep < 1) wstep = 1;
    int hstep = (int)((b3 - b1) / 56.0); if (hstep < 1) hstep = 1;
    int r0 = (int)((b1 + b3) / 2.0 + 0.5), r1 = (int)(b3 - (b3 - b1) * 0.1 + 0.5);
    int c0 = (int)(bl + 0.5), c1 = (int)(br + 0.5);
    r0 = min(max(r0, 0), cal.H2); r1 = min(max(r1, 0), cal.H2);      // python slice clamping
    c0 = min(max(c0, 0), cal.W2); c1 = min(max(c1, 0), cal.W2);
    const int nr = r1 > r0 ? (r1 - r0 + hstep - 1) / hstep : 0;
    const int nc = c1 > c0 ? (c1 - c0 + wstep - 1) / wstep : 0;
    const int total = nr * nc;
    // The box geometry is per OBJECT: one wave builds it, every pixel thread reads it from LDS (broadcast reads).
    // Round 2 finding (tools/da_probe2.py, profiles/dense_align_repeatability_r02.txt): when every one of the 256 threads
    // built its own copy in registers, about 1 call in 400 that ran BESIDE a forward pass on another stream came back with
    // lanes 48-63 of ONE wave of ONE far object holding a different geometry (the same wrong values every time; 85 of 1561
    // lattice pixels rejected, the coarse optimum moved from 31 m to 56 m).  Inputs, workspace and every other lane were
    // identical; the kernel uses no scratch; an instrumented build of the same code never reproduced it.  That points at an
    // instruction-timing hazard in that particular generated sequence, not at memory.  This arrangement removed the symptom
    // (0 of 3000 calls) and does 1/4 of the double-precision trigonometry; the cause itself is not established.
    if (wv == 0) {
        BoxGeom mine;
        build_box(poses + r * 7, mine, s_pc);
        if (lane == 0) s_g = mine;
    }
    const float fcx = (float)cal.cx, fcy = (float)cal.cy, ff = (float)cal.f;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const BoxGeom &g = s_g;
    float *out = uvz + (size_t)r * max_pixels * 3;
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + tid;
        bool ok = false;
        float u = 0.f, v = 0.f, dz = 0.f;
        if (i < total) {
            const int row = i / nc, col = i - row * nc;
            u = (float)(c0 + col * wstep);
            v = (float)(r0 + row * hstep);
            const float nu = (u - fcx) / ff, nv = (v - fcy) / ff;      // dense_align.py:48-49
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (ok) break;                                        // first valid plane wins (box_3d.py:79-82)
                float t = (nu * g.planes[pl][0] + nv * g.planes[pl][1]) + 1.0f * g.planes[pl][2];
                t = -(1.0f / t) * g.planes[pl][3];
                const float ic0 = nu * t - g.T[0], ic1 = nv * t - g.T[1], ic2 = 1.0f * t - g.T[2];
                bool in = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // R^T row k = column k of R (box_3d.py:64-69)
                    const float io = (g.R[0][k] * ic0 + g.R[1][k] * ic1) + g.R[2][k] * ic2;
                    in = in && io >= g.lo[k] && io <= g.hi[k];
                }
                if (in) { ok = true; dz = ic2; }
            }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int pre = s_base;
        for (int w = 0; w < wv; ++w) pre += wave_cnt[w];
        const int pos = pre + __popcll(bal & ((1ULL << lane) - 1ULL));
        if (ok && pos < max_pixels) { out[pos * 3 + 0] = u; out[pos * 3 + 1] = v; out[pos * 3 + 2] = dz; }
        __syncthreads();
        if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) {
        cnt[r] = min(s_base, max_pixels);
        cnt[gridDim.x + r] = s_base > max_pixels ? s_base : 0;      // overflow: reported as status -1, never silently truncated
    }
}

// ------------------------------------------------------------------ bilinear tap (grid_sample, align_corners, border)
__device__ __forceinline__ void grid_taps(float px, float py, int H, int W, int &x0, int &y0, float &wnw, float &wne,
                                          float &wsw, float &wse, bool &has_e, bool &has_s)
{
    float ix = fminf(fmaxf(px, 0.f), (float)(W - 1));
    float iy = fminf(fmaxf(py, 0.f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    x0 = (int)fx; y0 = (int)fy;
    const float ex = fx + 1.f, sy = fy + 1.f;
    wnw = (ex - ix) * (sy - iy);
    wne = (ix - fx) * (sy - iy);
    wsw = (ex - ix) * (iy - fy);
    wse = (ix - fx) * (iy - fy);
    has_e = x0 + 1 < W;
    has_s = y0 + 1 < H;
}

__device__ __forceinline__ float sample_plane(const float *__restrict__ p, int W, int x0, int y0, float wnw, float wne,
                                              float wsw, float wse, bool has_e, bool has_s)
{
    const float *q = p + (size_t)y0 * W + x0;
    float v = q[0] * wnw;
    if (has_e) v += q[1] * wne;
    if (has_s) v += q[W] * wsw;
    if (has_e && has_s) v += q[W + 1] * wse;
    return v;
}

// normalised grid coordinate -> pixel, exactly as F.grid_sample(align_corners=True) un-normalises it
__device__ __forceinline__ float unnorm(float coord, float half, int size)
{
    const float g = (coord - half) / half;              // dense_align.py:194-202,219
    return ((g + 1.f) / 2.f) * (float)(size - 1);
}

__global__ void left_sample_kernel(const float *__restrict__ up_l, const float *__restrict__ uvz,
                                   const int *__restrict__ cnt, int max_pixels, DaCalib cal,
                                   float *__restrict__ left_val)
{
    const int r = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cnt[r]) return;
    const int H = cal.H2, W = cal.W2;
    const float hw = (float)((double)(W - 1) / 2), hh = (float)((double)(H - 1) / 2);
    const float *q = uvz + ((size_t)r * max_pixels + p) * 3;
    int x0, y0; float a, b, c, d; bool he, hs;
    grid_taps(unnorm(q[0], hw, W), unnorm(q[1], hh, H), H, W, x0, y0, a, b, c, d, he, hs);
    float *o = left_val + ((size_t)r * max_pixels + p) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[ch] = sample_plane(up_l + (size_t)ch * H * W, W, x0, y0, a, b, c, d, he, hs);
}

// stage 0: coarse enumeration from the initial pose; stage 1: fine enumeration around best_depth
__global__ void make_enum_kernel(const float *__restrict__ poses, const float *__restrict__ best_depth, DaCalib cal,
                                 int R, int iters, int stage, float *__restrict__ depth_enum)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * iters) return;
    const int i = idx / R, r = idx - i * R;
    float d;
    if (stage == 0) {
        const float fbf = (float)cal.fb;
        const float dis_init = (1.0f / poses[r * 7 + 2]) * fbf;         // dense_align.py:265: scalar / tensor = tensor.reciprocal() * scalar in torch
        d = (((1.0f / dis_init) * (float)cal.f) * (float)cal.bl - (float)(iters * 0.5 / 2)) + (float)(0.5 * i);   // :283
        if (d < 1.5f) d = 1.5f;                                                         // :285
    } else {
        const double tint = 0.5 * 2.0 / iters;                                          // :291
        d = (best_depth[r] - (float)(iters * tint / 2)) + (float)(tint * i);    // :294
    }
    depth_enum[idx] = d;
}

// grid (iters, R); block 256
__global__ __launch_bounds__(256) void cost_kernel(const float *__restrict__ up_r, const float *__restrict__ uvz,
                                                   const int *__restrict__ cnt, const float *__restrict__ left_val,
                                                   const float *__restrict__ depth_enum, int max_pixels, int R,
                                                   DaCalib cal, float *__restrict__ cost)
{
    __shared__ float red[4];
    const int it = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    const int H = cal.H2, W = cal.W2, n = cnt[r];
    const float hw = (float)((double)(W - 1) / 2), hh = (float)((double)(H - 1) / 2);
    const float fbf = (float)cal.fb;
    const float depth = depth_enum[it * R + r];
    const float dis_inv = 1.0f / ((1.0f / depth) * fbf);                 // dis_enum.reciprocal(), :211,215
    float acc = 0.f;
    for (int p = tid; p < n; p += 256) {
        const float *q = uvz + ((size_t)r * max_pixels + p) * 3;
        const float gdd = 1.0f / (q[2] / fbf + dis_inv);                 // per-pixel disparity (:215)
        int x0, y0; float a, b, c, d; bool he, hs;
        grid_taps(unnorm(q[0] - gdd, hw, W), unnorm(q[1], hh, H), H, W, x0, y0, a, b, c, d, he, hs);
        const float *lv = left_val + ((size_t)r * max_pixels + p) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float rv = sample_plane(up_r + (size_t)ch * H * W, W, x0, y0, a, b, c, d, he, hs);
            acc += fabsf(lv[ch] - rv);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) cost[it * R + r] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void argmin_kernel(const float *__restrict__ cost, const float *__restrict__ depth_enum, int R, int iters,
                              float *__restrict__ best_depth)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int bi = 0;
    float bv = cost[r];
    for (int i = 1; i < iters; ++i) {
        const float v = cost[i * R + r];
        if (v < bv) { bv = v; bi = i; }                                  // first minimum (:232)
    }
    best_depth[r] = depth_enum[bi * R + r];
}

__global__ void finish_kernel(const float *__restrict__ poses, const float *__restrict__ best_depth,
                              const int *__restrict__ cnt, int R, DaCalib cal, float *__restrict__ status,
                              float *__restrict__ best_dis)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    long total = 0;
    for (int i = 0; i < R; ++i) total += cnt[i];
    const float fbf = (float)cal.fb;
    if (total == 0) {                                                    // dense_align.py:272-274
        status[r] = 0.f;
        best_dis[r] = (1.0f / poses[r * 7 + 2]) * fbf;                    // dis_init (:265), as above
        return;
    }
    status[r] = cnt[R + r] ? -1.f : (cnt[r] > 0 ? 1.f : 0.f);            // :276-277; -1 = lattice did not fit max_pixels
    best_dis[r] = (1.0f / (best_depth[r] * (float)cal.scale2)) * fbf + 0.5f;      // :298 (scalar / tensor again)
}

struct DaLayout {
    size_t up_l, up_r, uvz, cnt, left_val, depth_enum[2], cost[2], best[2], total;   // [stage]: coarse and fine keep their own slots
};

static DaLayout da_layout(int H, int W, int R, int max_pixels)
{
    DaLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t img = (size_t)3 * (2 * H) * (2 * W) * sizeof(float);
    L.up_l = take(img);
    L.up_r = take(img);
    L.uvz = take((size_t)R * max_pixels * 3 * sizeof(float));
    L.cnt = take((size_t)2 * R * sizeof(int));                  // [R] valid samples, [R] overflow flags
    L.left_val = take((size_t)R * max_pixels * 3 * sizeof(float));
    for (int stage = 0; stage < 2; ++stage) {
        L.depth_enum[stage] = take((size_t)50 * R * sizeof(float));
        L.cost[stage] = take((size_t)50 * R * sizeof(float));
        L.best[stage] = take((size_t)R * sizeof(float));
    }
    L.total = off;
    return L;
}

}  // namespace srcnn

extern "C" {

size_t srcnn_dense_align_workspace_bytes(int H, int W, int R, int max_pixels)
{
    return srcnn::da_layout(H, W, R > 0 ? R : 1, max_pixels > 0 ? max_pixels : 1).total;
}

int srcnn_dense_align_workspace_layout(int H, int W, int R, int max_pixels, size_t *offsets, int n_offsets)
{
    using namespace srcnn;
    SRCNN_REQUIRE(offsets && n_offsets >= 7 && H > 1 && W > 1 && R > 0 && max_pixels > 0, "7 offsets; positive sizes");
    const DaLayout L = da_layout(H, W, R, max_pixels);
    offsets[0] = L.cnt;
    for (int stage = 0; stage < 2; ++stage) {
        offsets[1 + 3 * stage] = L.depth_enum[stage];
        offsets[2 + 3 * stage] = L.cost[stage];
        offsets[3 + 3 * stage] = L.best[stage];
    }
    return SRCNN_OK;
}

int srcnn_dense_align(const float *im_left, const float *im_right, int H, int W, double scale, double p2_00,
                      double p2_02, double p2_12, double p2_03_minus_p3_03, const float *boxes, const float *borders,
                      const float *poses, const float *valid, int R, int max_pixels, float *status, float *best_dis,
                      void *workspace, size_t workspace_bytes, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(im_left && im_right && boxes && borders && poses && status && best_dis, "null pointer");
    SRCNN_REQUIRE(H > 1 && W > 1 && R >= 0 && max_pixels > 0, "bad sizes");
    if (R == 0) return SRCNN_OK;
    DaLayout L = da_layout(H, W, R, max_pixels);
    if (!workspace || workspace_bytes < L.total) {
        set_error("srcnn_dense_align: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return SRCNN_ERR_WORKSPACE;
    }
    DaCalib cal;
    cal.scale2 = scale * 2;                                      // dense_align.py:255
    cal.f = p2_00 * cal.scale2;                                  // :259
    cal.bl = p2_03_minus_p3_03 * cal.scale2 / cal.f;             // :260
    cal.fb = cal.f * cal.bl;
    cal.cx = p2_02 * cal.scale2;                                 // :29
    cal.cy = p2_12 * cal.scale2;
    cal.H2 = 2 * H;
    cal.W2 = 2 * W;
    char *ws = static_cast<char *>(workspace);
    float *up_l = reinterpret_cast<float *>(ws + L.up_l), *up_r = reinterpret_cast<float *>(ws + L.up_r);
    float *uvz = reinterpret_cast<float *>(ws + L.uvz), *left_val = reinterpret_cast<float *>(ws + L.left_val);
    int *cnt = reinterpret_cast<int *>(ws + L.cnt);
    float *best[2] = {reinterpret_cast<float *>(ws + L.best[0]), reinterpret_cast<float *>(ws + L.best[1])};
    hipStream_t st = as_stream(stream);
    SRCNN_LAUNCH(upsample2x_kernel, dim3(cdiv(W, 256), cdiv(2 * H, UP_ROWS), 6), dim3(256), 0, st, im_left, im_right, H, W, up_l, up_r);
    SRCNN_LAUNCH(sample_kernel, dim3(R), dim3(256), 0, st, boxes, borders, poses, valid, cal, max_pixels, uvz, cnt);
    SRCNN_LAUNCH(left_sample_kernel, dim3(cdiv(max_pixels, 256), R), dim3(256), 0, st, up_l, uvz, cnt, max_pixels,
                       cal, left_val);
    for (int stage = 0; stage < 2; ++stage) {
        const int iters = stage == 0 ? 50 : 20;                  // dense_align.py:280,290
        float *depth_enum = reinterpret_cast<float *>(ws + L.depth_enum[stage]);
        float *cost = reinterpret_cast<float *>(ws + L.cost[stage]);
        SRCNN_LAUNCH(make_enum_kernel, dim3(cdiv(R * iters, 256)), dim3(256), 0, st, poses, best[0], cal, R, iters,
                           stage, depth_enum);
        SRCNN_LAUNCH(cost_kernel, dim3(iters, R), dim3(256), 0, st, up_r, uvz, cnt, left_val, depth_enum,
                           max_pixels, R, cal, cost);
        SRCNN_LAUNCH(argmin_kernel, dim3(cdiv(R, 64)), dim3(64), 0, st, cost, depth_enum, R, iters, best[stage]);
    }
    SRCNN_LAUNCH(finish_kernel, dim3(cdiv(R, 64)), dim3(64), 0, st, poses, best[1], cnt, R, cal, status, best_dis);
    return check_launch("srcnn_dense_align");
}

}  // extern "C"
