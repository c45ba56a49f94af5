// Greedy NMS, fully on the device (gfx950).
//
// Replaces lib/model/nms/src/nms_cuda_kernel.cu:31-161 of the reference:
//   * pair_mask_kernel  == nms_kernel (:41-85): one 64-lane wavefront per 64x64 tile, one
//     uint64 suppression word per (box, column block); IoU arithmetic op-for-op as devIoU
//     (:31-39) so the keep list is bit-identical.  Tiles below the diagonal are skipped:
//     the greedy pass never reads them (:139-142 starts at j = nblock).
//   * greedy_scan_kernel replaces the HOST loop (:117-144): the reference copies the whole
//     mask to the CPU and reduces it serially; here one wavefront per problem walks the
//     column blocks, resolves the 64 boxes of a block against the diagonal tile with
//     wave shuffles, and ORs the kept rows into per-lane `removed` words.  No D2H copy,
//     no sync, no malloc: graph-capturable.
#include "common.h"
#include <mutex>

namespace srcnn {

__device__ __forceinline__ float iou_plus1(const float4 a, const float4 b)
{
    // nms_cuda_kernel.cu:31-39; every operation rounded separately (library is built
    // with -ffp-contract=off, and there is no mul+add to fuse here anyway).
    float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    float width = fmaxf(right - left + 1.0f, 0.0f);
    float height = fmaxf(bottom - top + 1.0f, 0.0f);
    float inter = width * height;
    float sa = (a.z - a.x + 1.0f) * (a.w - a.y + 1.0f);
    float sb = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
    return inter / (sa + sb - inter);
}

// grid: (col_blocks, col_blocks, nb); block: 64 threads (one wavefront == one mask word)
__global__ __launch_bounds__(64) void pair_mask_kernel(const float *__restrict__ dets, int n, int dim,
                                                       float thresh, unsigned long long *__restrict__ mask,
                                                       int col_blocks)
{
    const int col_start = blockIdx.x, row_start = blockIdx.y;
    if (row_start > col_start) return;
    const float *d = dets + (size_t)blockIdx.z * n * dim;
    unsigned long long *m = mask + (size_t)blockIdx.z * n * col_blocks;
    const int row_size = min(n - row_start * 64, 64);
    const int col_size = min(n - col_start * 64, 64);
    __shared__ float4 cols[64];
    const int t = threadIdx.x;
    if (t < col_size) {
        const float *p = d + (size_t)(col_start * 64 + t) * dim;
        cols[t] = make_float4(p[0], p[1], p[2], p[3]);
    }
    __syncthreads();
    if (t < row_size) {
        const int cur = row_start * 64 + t;
        const float *p = d + (size_t)cur * dim;
        const float4 me = make_float4(p[0], p[1], p[2], p[3]);
        unsigned long long bits = 0;
        const int start = (row_start == col_start) ? t + 1 : 0;
        for (int i = start; i < col_size; ++i)
            if (iou_plus1(me, cols[i]) > thresh) bits |= 1ULL << i;
        m[(size_t)cur * col_blocks + col_start] = bits;
    }
}

// One wavefront per problem. Lane l owns removed-words j = l + 64*s, s < WPL.
template <int WPL>
__global__ __launch_bounds__(64) void greedy_scan_kernel(const unsigned long long *__restrict__ mask, int n,
                                                         int col_blocks, const int *__restrict__ n_valid,
                                                         int *__restrict__ keep_out, int *__restrict__ num_out)
{
    const int prob = blockIdx.x;
    const unsigned long long *m = mask + (size_t)prob * n * col_blocks;
    int *keep = keep_out + (size_t)prob * n;
    const int lane = threadIdx.x;
    const int nv = n_valid ? min(n_valid[prob], n) : n;
    unsigned long long removed[WPL];
#pragma unroll
    for (int s = 0; s < WPL; ++s) removed[s] = 0;
    int count = 0;
    const int nblocks = (nv + 63) / 64;
    for (int b = 0; b < nblocks; ++b) {
        const int in_block = min(64, nv - b * 64);
        // current removed word of this block lives in lane (b & 63), slot (b >> 6)
        unsigned long long mine = 0;
#pragma unroll
        for (int s = 0; s < WPL; ++s)
            if ((b >> 6) == s) mine = removed[s];
        unsigned long long cur = __shfl(mine, b & 63);
        if (in_block < 64) cur |= ~0ULL << in_block;  // boxes past the end never get kept
        // diagonal tile: lane t holds the word of box 64b+t against its own block
        unsigned long long diag = (lane < in_block) ? m[(size_t)(b * 64 + lane) * col_blocks + b] : 0ULL;
        unsigned long long kept = 0;
        for (int t = 0; t < 64; ++t) {
            unsigned long long d = __shfl(diag, t);
            if (!((cur >> t) & 1ULL)) {  // wave-uniform
                kept |= 1ULL << t;
                cur |= d;
            }
        }
        if ((kept >> lane) & 1ULL)
            keep[count + __popcll(kept & ((1ULL << lane) - 1ULL))] = b * 64 + lane;
        count += __popcll(kept);
        // OR the kept rows into the removed words of the later blocks (16 loads in flight)
#pragma unroll
        for (int s = 0; s < WPL; ++s) {
            const int j = lane + 64 * s;
            if (j > b && j < nblocks) {
                unsigned long long acc = removed[s];
                for (int t0 = 0; t0 < 64; t0 += 16) {
                    if (((kept >> t0) & 0xFFFFULL) == 0) continue;  // wave-uniform skip
                    // unconditional loads (row clamped) so 16 stay in flight; select on the value
                    unsigned long long v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int r = min(b * 64 + t0 + u, n - 1);
                        v[u] = m[(size_t)r * col_blocks + j];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if ((kept >> (t0 + u)) & 1ULL) acc |= v[u];
                }
                removed[s] = acc;
            }
        }
    }
    if (lane == 0) num_out[prob] = count;
}

static int launch_nms(int *keep_out, const float *dets, int *num_out, const int *n_valid, int nb, int n,
                      int dim, float thresh, void *ws, size_t ws_bytes, hipStream_t st)
{
    SRCNN_REQUIRE(nb >= 0 && n >= 0 && dim >= 4, "bad sizes");
    SRCNN_REQUIRE(n <= 64 * 64 * 4, "n > 16384 boxes per problem not supported");
    if (nb == 0) return SRCNN_OK;
    if (n == 0) {
        SRCNN_HIP_TRY(hipMemsetAsync(num_out, 0, sizeof(int) * nb, st));
        return SRCNN_OK;
    }
    const int cb = cdiv(n, 64);
    const size_t need = (size_t)nb * n * cb * sizeof(unsigned long long);
    if (ws == nullptr || ws_bytes < need) {
        set_error("nms: workspace too small (%zu < %zu)", ws_bytes, need);
        return SRCNN_ERR_WORKSPACE;
    }
    auto *mask = static_cast<unsigned long long *>(ws);
    hipLaunchKernelGGL(pair_mask_kernel, dim3(cb, cb, nb), dim3(64), 0, st, dets, n, dim, thresh, mask, cb);
    if (cb <= 64)
        hipLaunchKernelGGL(greedy_scan_kernel<1>, dim3(nb), dim3(64), 0, st, mask, n, cb, n_valid, keep_out, num_out);
    else if (cb <= 128)
        hipLaunchKernelGGL(greedy_scan_kernel<2>, dim3(nb), dim3(64), 0, st, mask, n, cb, n_valid, keep_out, num_out);
    else
        hipLaunchKernelGGL(greedy_scan_kernel<4>, dim3(nb), dim3(64), 0, st, mask, n, cb, n_valid, keep_out, num_out);
    return check_launch("nms");
}

// library-owned scratch for the legacy-named entry point
static void *g_pool = nullptr;
static size_t g_pool_bytes = 0;
static std::mutex g_pool_mu;

}  // namespace srcnn

extern "C" {

size_t srcnn_nms_workspace_bytes(int n)
{
    if (n <= 0) return 256;
    return srcnn::align_up((size_t)n * srcnn::cdiv(n, 64) * 8, 256);
}

size_t srcnn_nms_batched_workspace_bytes(int nb, int n)
{
    if (n <= 0 || nb <= 0) return 256;
    return srcnn::align_up((size_t)nb * n * srcnn::cdiv(n, 64) * 8, 256);
}

int srcnn_nms(int *keep_out, const float *dets, int *num_out, int n, int dim, float thresh, void *workspace,
              size_t workspace_bytes, srcnn_stream_t stream)
{
    return srcnn::launch_nms(keep_out, dets, num_out, nullptr, 1, n, dim, thresh, workspace, workspace_bytes,
                             srcnn::as_stream(stream));
}

int srcnn_nms_batched(int *keep_out, const float *dets, int *num_out, const int *n_valid, int nb, int n, int dim,
                      float thresh, void *workspace, size_t workspace_bytes, srcnn_stream_t stream)
{
    return srcnn::launch_nms(keep_out, dets, num_out, n_valid, nb, n, dim, thresh, workspace, workspace_bytes,
                             srcnn::as_stream(stream));
}

int nms_cuda(int *keep_out, const float *boxes, int *num_out, int boxes_num, int boxes_dim,
             float nms_overlap_thresh, srcnn_stream_t stream)
{
    using namespace srcnn;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const size_t need = srcnn_nms_workspace_bytes(boxes_num);
    if (need > g_pool_bytes) {  // grow-only pool; never on the steady-state path
        if (g_pool) {
            if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) return 0;
            (void)hipFree(g_pool);
            g_pool = nullptr;
            g_pool_bytes = 0;
        }
        if (hipMalloc(&g_pool, need) != hipSuccess) {
            set_error("nms_cuda: hipMalloc(%zu) failed", need);
            return 0;
        }
        g_pool_bytes = need;
    }
    int rc = launch_nms(keep_out, boxes, num_out, nullptr, 1, boxes_num, boxes_dim, nms_overlap_thresh, g_pool,
                        g_pool_bytes, as_stream(stream));
    return rc == SRCNN_OK ? 1 : 0;
}

}  // extern "C"
