// Greedy NMS, fully on the device (gfx950).
//
// Replaces lib/model/nms/src/nms_cuda_kernel.cu:31-161 of the reference:
//   * pair_mask_kernel  == nms_kernel (:41-85): one 64-lane wavefront per 64x64 tile, one
//     uint64 suppression word per (box, column block); IoU arithmetic op-for-op as devIoU
//     (:31-39) so the keep list is bit-identical.  Tiles below the diagonal are skipped:
//     the greedy pass never reads them (:139-142 starts at j = nblock).
//   * greedy_scan_kernel replaces the HOST loop (:117-144): the reference copies the whole
//     mask to the CPU and reduces it serially; here one workgroup per problem walks the
//     column blocks on the device (see the kernel's comment).  No D2H copy, no sync, no
//     malloc: graph-capturable.
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

// One workgroup per problem, one wavefront per 64 mask words (column blocks): lane l of wave w owns
// the `removed` word of column block j = 64*w + l.  For each block b of 64 boxes (in score order):
//   1. every lane issues the loads of ALL 64 candidate rows' word j up front (64 independent loads
//      in flight: the walk pays ~one L2 round trip per block instead of one per kept row);
//   2. meanwhile the block's 64 boxes are resolved against the diagonal tile on the scalar unit:
//      only KEPT boxes cost an iteration (find-first-set on the availability mask + one readlane);
//   3. the rows of the kept boxes are OR-ed into the lane's word, the word of block b+1.. is
//      published through LDS for the next step.
// Waves of a workgroup stay in lockstep (two barriers per step); every wave resolves the diagonal
// redundantly so no cross-wave broadcast of the kept mask is needed.
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xFFFFFFFFULL), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void greedy_scan_kernel(const unsigned long long *__restrict__ mask, int n,
                                                          int col_blocks, const int *__restrict__ n_valid,
                                                          int *__restrict__ keep_out, int *__restrict__ num_out)
{
    __shared__ unsigned long long removed[256];
    const int prob = blockIdx.x;
    const unsigned long long *m = mask + (size_t)prob * n * col_blocks;
    int *keep = keep_out + (size_t)prob * n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nv = n_valid ? min(n_valid[prob], n) : n;
    const int nblocks = (nv + 63) / 64;
    const int j = min(tid, col_blocks - 1);           // my column block (clamped lanes do harmless work)
    removed[tid] = 0ULL;
    __syncthreads();
    int count = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int in_block = min(64, nv - b * 64);
        // 1. all 64 rows' words for my column, unconditionally (rows clamped), 64 loads in flight
        unsigned long long v[64];
        const unsigned long long *col = m + (size_t)(b * 64) * col_blocks + j;
        const int rmax = n - 1 - b * 64;
#pragma unroll
        for (int t = 0; t < 64; ++t) v[t] = col[(size_t)min(t, rmax) * col_blocks];
        // 2. diagonal resolve (wave-uniform; scalar unit)
        const unsigned long long diag = (lane < in_block) ? m[(size_t)(b * 64 + lane) * col_blocks + b] : 0ULL;
        unsigned long long cur = removed[b];
        if (in_block < 64) cur |= ~0ULL << in_block;
        unsigned cl = __builtin_amdgcn_readfirstlane((unsigned)(cur & 0xFFFFFFFFULL));
        unsigned ch = __builtin_amdgcn_readfirstlane((unsigned)(cur >> 32));
        cur = ((unsigned long long)ch << 32) | cl;
        unsigned long long kept = 0, avail = ~cur;
        while (avail) {
            const int t = __ffsll((long long)avail) - 1;
            kept |= 1ULL << t;
            cur |= readlane64(diag, t);                 // bits > t only (kernel writes the upper triangle)
            avail = ~cur & ~((2ULL << t) - 1ULL);
            if (t == 63) break;
        }
        if (tid < 64) {
            if ((kept >> lane) & 1ULL) keep[count + __popcll(kept & ((1ULL << lane) - 1ULL))] = b * 64 + lane;
        }
        count += __popcll(kept);
        // 3. OR the kept rows into my word
        unsigned long long acc = 0;
#pragma unroll
        for (int t = 0; t < 64; ++t)
            if ((kept >> t) & 1ULL) acc |= v[t];
        __syncthreads();                                // everyone has read removed[b]
        if (tid > b && tid < col_blocks) removed[tid] |= acc;
        __syncthreads();
    }
    if (tid == 0) num_out[prob] = count;
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
    const int waves = cdiv(cb, 64);                    // <= 4 (n <= 16384)
    hipLaunchKernelGGL(greedy_scan_kernel, dim3(nb), dim3(64 * waves), 0, st, mask, n, cb, n_valid, keep_out, num_out);
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
