// Greedy NMS, fully on the device (gfx950).
//
// Replaces lib/model/nms/src/nms_cuda_kernel.cu:31-161 of the reference:
//   * pair_mask_kernel  == nms_kernel (:41-85): one 64-lane wavefront per 64x64 tile, one
//     uint64 suppression word per (box, column block); IoU arithmetic op-for-op as devIoU
//     (:31-39) so the keep list is bit-identical.  Tiles below the diagonal are skipped:
//     the greedy pass never reads them (:139-142 starts at j = nblock).  The words are stored
//     COLUMN-BLOCK major (maskT[c][box]): what the scan needs for block c is then contiguous.
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

// grid: (col_blocks, ceil(col_blocks / PM_ROWS), nb); block: PM_ROWS wavefronts = PM_ROWS row blocks against ONE column block
// (one wavefront == one 64x64 tile == one mask word per lane; the column block's boxes are staged once per workgroup).
// 6000 boxes are 94 x 94 tiles per problem: one wavefront per workgroup was 17.7 k workgroups, dispatch-bound (44 us alone).
constexpr int PM_ROWS = 4;
__global__ __launch_bounds__(64 * PM_ROWS) void pair_mask_kernel(const float *__restrict__ dets, int n, int dim,
                                                                 float thresh, unsigned long long *__restrict__ mask,
                                                                 int col_blocks, unsigned long long *__restrict__ zero_word)
{
    const int col_start = blockIdx.x, row_start = blockIdx.y * PM_ROWS + (threadIdx.x >> 6);
    if ((blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) *zero_word = 0ULL;   // read by the scan's dummy DMAs
    if ((int)blockIdx.y * PM_ROWS > col_start) return;              // (uniform) every row block of this workgroup lies below the diagonal
    const float *d = dets + (size_t)blockIdx.z * n * dim;
    unsigned long long *m = mask + (size_t)blockIdx.z * n * col_blocks;
    const int row_size = min(n - row_start * 64, 64);               // (<= 0 for a row block past the end)
    const int col_size = min(n - col_start * 64, 64);
    __shared__ float4 cols[64];
    const int t = threadIdx.x & 63;
    if (threadIdx.x < col_size) {
        const float *p = d + (size_t)(col_start * 64 + threadIdx.x) * dim;
        cols[threadIdx.x] = make_float4(p[0], p[1], p[2], p[3]);
    }
    __syncthreads();
    if (row_start <= col_start && t < row_size) {
        const int cur = row_start * 64 + t;
        const float *p = d + (size_t)cur * dim;
        const float4 me = make_float4(p[0], p[1], p[2], p[3]);
        unsigned long long bits = 0;
        const int start = (row_start == col_start) ? t + 1 : 0;
        for (int i = start; i < col_size; ++i)
            if (iou_plus1(me, cols[i]) > thresh) bits |= 1ULL << i;
        m[(size_t)col_start * n + cur] = bits;          // column-block major
    }
}

// The greedy pass.  Block c of 64 boxes (score order) can be resolved once the suppression word of
// column block c is known for every box kept so far:  removed_c = OR_{kept i < 64c} maskT[c][i].
// Instead of OR-ing whole mask rows into a `removed` array after every block (4.5 MB of row reads for
// 6000 boxes, one dependent memory round trip per block), each step GATHERS only column c over the
// list of kept boxes (<= a few hundred words), and everything block t = c+D needs is requested D = 3
// steps ahead, before blocks c .. c+D-1 are resolved:
//   * the gather of column t over the boxes kept in blocks < c   (the kept list at the start of step c),
//   * all 64 words maskT[t][64(c+w) + lane] of block c+w, carried by wave w (speculative: selected by
//     that block's keep bits when step t consumes them),
//   * the diagonal tile of block t.
// The requests are global_load_lds DMAs into a per-wave LDS ring (lane-linear, so a lane reads back only
// what its own wave requested: no extra barrier), a fixed number per step (pieces that do not exist
// read a zero word kept behind the mask), and each step waits with a hand-written `s_waitcnt vmcnt(n)`
// for the OLDEST generation only: D-1 generations stay in flight across the two workgroup barriers
// of a step, so a step costs max(L2 round trip / D, resolve) instead of their sum.  Wave 0 resolves
// the block on the scalar unit: only KEPT boxes cost an iteration (find-first-set on the availability
// mask + one readlane of the diagonal word).
// PAIR = 2 (proposal layer): problems (2b, 2b+1) = the left and right boxes of one image walk in lockstep
// in one workgroup and stop as soon as the INTERSECTION of their keep lists has `stop_after` entries --
// all that proposal_layer.py:127-143 consumes (intersect1d, then [:post_nms_topN]).
typedef unsigned long long u64;
constexpr int SCAN_T = 256;                  // threads per problem (4 wavefronts)
constexpr int SCAN_G = 4;                    // gathered words per thread per generation (SCAN_G * SCAN_T kept boxes)
constexpr int SCAN_D = 3;                    // generations in flight (waves 0..D-1 carry the speculative blocks)
constexpr int SCAN_W = SCAN_G + 2;           // words per lane per generation: gather, spec, diag
constexpr int SCAN_LPW = 2 * SCAN_W;         // DMA instructions per wave per step (lo + hi dword of each word)
constexpr int SCAN_RING = SCAN_D * 4 * SCAN_LPW * 64;   // dwords of ring per problem

__device__ __forceinline__ u64 readlane64(u64 v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xFFFFFFFFULL), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// OR over the 64 lanes, returned wave-uniform.  DPP row shifts / row broadcasts (register crossbar, no LDS
// permutes): after row_shr 1,2,4,8 lane 15 of each 16-lane row holds the row's OR, row_bcast:15 / :31 fold the
// rows together into lane 63.  Lanes shifted in from outside a row read 0 (bound_ctrl), the identity of OR.
__device__ __forceinline__ unsigned wave_or32(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
#else
    return v;
#endif
}

__device__ __forceinline__ u64 wave_or(u64 v)
{
    const unsigned lo = wave_or32((unsigned)(v & 0xFFFFFFFFULL)), hi = wave_or32((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// LDS-visibility barrier that leaves vector-memory operations in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one 4-byte-per-lane DMA: lane l's dword lands at lds_wave_base + 4*l  (device-only builtin)
__device__ __forceinline__ void dma4(const void *gsrc, unsigned *lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 4, 0, 0);
#else
    (void)gsrc;
    (void)lds_wave_base;
#endif
}

extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];

template <int PAIR>
__global__ __launch_bounds__(SCAN_T *PAIR) void greedy_scan_kernel(const u64 *__restrict__ maskT, const u64 *__restrict__ zero_word,
                                                                    int n, int col_blocks, const int *__restrict__ n_valid,
                                                                    int *__restrict__ keep_out, int *__restrict__ num_out,
                                                                    int stop_after)
{
    u64 *red = reinterpret_cast<u64 *>(scan_lds);                     // [PAIR][4] per-wave partial ORs
    u64 *kbits = red + PAIR * 4;                                       // [PAIR] keep bits of the block just resolved
    unsigned *ring_all = reinterpret_cast<unsigned *>(kbits + PAIR);   // [PAIR][D][4 waves][LPW][64 lanes]
    int *kept_all = reinterpret_cast<int *>(ring_all + PAIR * SCAN_RING);   // [PAIR][n] kept boxes so far
    const int tid = threadIdx.x, grp = tid / SCAN_T, gt = tid % SCAN_T;
    const int wave = __builtin_amdgcn_readfirstlane(gt >> 6), lane = gt & 63;
    const int prob = blockIdx.x * PAIR + grp;
    const u64 *m = maskT + (size_t)prob * col_blocks * n;
    int *keep = keep_out + (size_t)prob * n;
    int *kept = kept_all + (size_t)grp * n;
    unsigned *ring = ring_all + grp * SCAN_RING + wave * (SCAN_LPW * 64);   // this wave's part of slot 0
    constexpr int SLOT = 4 * SCAN_LPW * 64;                                  // dwords per generation
    const int nv = n_valid ? min(n_valid[prob], n) : n;
    const int nblocks = (nv + 63) / 64;
    int nb_all = nblocks;
    if (PAIR == 2) {
        const int nvo = n_valid ? min(n_valid[prob ^ 1], n) : n;
        nb_all = max(nblocks, (nvo + 63) / 64);
    }
    int count = 0, both = 0;
    u64 hist[SCAN_D];                                                  // keep bits of blocks c-1, c-2, .. c-D
#pragma unroll
    for (int d = 0; d < SCAN_D; ++d) hist[d] = 0ULL;

    // request block t into ring slot `slot`: always SCAN_LPW DMAs per wave; missing pieces read the zero word
    auto issue = [&](int slot, int t, int count_now) {
        unsigned *dst = ring + slot * SLOT;
        const bool live = t < nblocks;
        const u64 *col = m + (size_t)(live ? t : 0) * n;
#pragma unroll
        for (int q = 0; q < SCAN_G; ++q) {
            const int k = gt + q * SCAN_T;
            const u64 *src = (live && k < count_now) ? col + kept[k] : zero_word;
            dma4(src, dst + (2 * q) * 64);
            dma4(reinterpret_cast<const unsigned *>(src) + 1, dst + (2 * q + 1) * 64);
        }
        const int sb = t - SCAN_D + wave;                              // waves >= D carry nothing
        const u64 *sp = (live && wave < SCAN_D && sb >= 0) ? col + sb * 64 + lane : zero_word;
        dma4(sp, dst + (2 * SCAN_G) * 64);
        dma4(reinterpret_cast<const unsigned *>(sp) + 1, dst + (2 * SCAN_G + 1) * 64);
        const int r = t * 64 + lane;
        const u64 *dg = (live && r < nv) ? col + r : zero_word;
        dma4(dg, dst + (2 * SCAN_G + 2) * 64);
        dma4(reinterpret_cast<const unsigned *>(dg) + 1, dst + (2 * SCAN_G + 3) * 64);
    };
    auto word = [&](int slot, int j) -> u64 {
        const unsigned *src = ring + slot * SLOT + (2 * j) * 64 + lane;
        return ((u64)src[64] << 32) | (u64)src[0];
    };

#pragma unroll
    for (int d = 0; d < SCAN_D; ++d) issue(d, d, 0);
    int slot = 0;
    int cnt_ring[SCAN_D];                                              // kept-list length each generation's gather covers
#pragma unroll
    for (int d = 0; d < SCAN_D; ++d) cnt_ring[d] = 0;

    for (int c = 0; c < nb_all; ++c) {
        const bool active = c < nblocks;
        // ---- the oldest generation (block c) has landed once at most (D-1) generations are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SCAN_D - 1) * SCAN_LPW) : "memory");
        u64 part = 0ULL;
#pragma unroll
        for (int q = 0; q < SCAN_G; ++q) part |= word(slot, q);
        const int cnt_c = slot == 0 ? cnt_ring[0] : slot == 1 ? cnt_ring[1] : cnt_ring[2];
        if (cnt_c > SCAN_G * SCAN_T) {                                 // kept list longer than the DMA window
            const u64 *colc = m + (size_t)min(c, col_blocks - 1) * n;
            int k = SCAN_G * SCAN_T + gt;
            for (; k + 3 * SCAN_T < cnt_c; k += 4 * SCAN_T) {          // four independent loads in flight
                const u64 a0 = colc[kept[k]], a1 = colc[kept[k + SCAN_T]], a2 = colc[kept[k + 2 * SCAN_T]],
                          a3 = colc[kept[k + 3 * SCAN_T]];
                part |= (a0 | a1) | (a2 | a3);
            }
            for (; k < cnt_c; k += SCAN_T) part |= colc[kept[k]];
        }
        // wave w carries block c-D+w, whose keep bits are hist[D-1-w]
        const u64 hb = wave == 0 ? hist[2] : wave == 1 ? hist[1] : wave == 2 ? hist[0] : 0ULL;
        if ((hb >> lane) & 1ULL) part |= word(slot, SCAN_G);
        const u64 cur_diag = word(slot, SCAN_G + 1);
        part = wave_or(part);
        if (lane == 0) red[grp * 4 + wave] = part;
        lds_barrier();                                                 // ring slot read back by everyone
        // ---- request block c+D into the slot just consumed (kept list = blocks < c; blocks c.. via `spec`)
        issue(slot, c + SCAN_D, count);
        if (slot == 0) cnt_ring[0] = count; else if (slot == 1) cnt_ring[1] = count; else cnt_ring[2] = count;
        if (wave == 0) {
            u64 kb = 0ULL;
            if (active) {
                u64 cur = red[grp * 4] | red[grp * 4 + 1] | red[grp * 4 + 2] | red[grp * 4 + 3];
                const int in_block = min(64, nv - c * 64);
                if (in_block < 64) cur |= ~0ULL << in_block;
                const unsigned cl = __builtin_amdgcn_readfirstlane((unsigned)(cur & 0xFFFFFFFFULL));
                const unsigned ch = __builtin_amdgcn_readfirstlane((unsigned)(cur >> 32));
                cur = ((u64)ch << 32) | cl;
                // Peel in parallel first: a box none of whose still-undecided block mates can suppress it is kept
                // at once (typically most of a block); two wave-wide ORs per round instead of one scalar iteration
                // per kept box.  What is left (boxes inside overlap chains) is walked in order on the scalar unit.
                u64 avail = ~cur;
#pragma unroll 1
                for (int round = 0; round < 3 && avail; ++round) {
                    const u64 col_or = wave_or(((avail >> lane) & 1ULL) ? cur_diag : 0ULL);
                    const u64 free_now = avail & ~col_or;
                    if (!free_now) break;
                    kb |= free_now;
                    cur |= wave_or(((free_now >> lane) & 1ULL) ? cur_diag : 0ULL);
                    avail = ~cur & ~kb;
                }
                while (avail) {
                    const int t = __ffsll((long long)avail) - 1;
                    kb |= 1ULL << t;
                    cur |= readlane64(cur_diag, t);             // bits > t only (upper triangle)
                    if (t == 63) break;
                    avail = ~cur & ~kb & ~((2ULL << t) - 1ULL);
                }
                if ((kb >> lane) & 1ULL) {
                    const int pos = count + __popcll(kb & ((1ULL << lane) - 1ULL));
                    keep[pos] = c * 64 + lane;
                    kept[pos] = c * 64 + lane;
                }
            }
            if (lane == 0) kbits[grp] = kb;
        }
        lds_barrier();
        const u64 kb = kbits[grp];
        hist[2] = hist[1];
        hist[1] = hist[0];
        hist[0] = kb;
        count += __popcll(kb);
        slot = slot + 1 == SCAN_D ? 0 : slot + 1;
        if (PAIR == 2) {
            both += __popcll(kbits[0] & kbits[1]);
            if (stop_after > 0 && both >= stop_after) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no DMA may outlive the workgroup's LDS
    if (gt == 0) num_out[prob] = count;
}

template <int PAIR>
static void launch_scan(const u64 *mask, const u64 *zero_word, int nb, int n, int cb, const int *n_valid, int *keep_out,
                        int *num_out, int stop_after, hipStream_t st)
{
    static size_t configured = 0;
    const size_t lds = (size_t)PAIR * 5 * sizeof(u64) + (size_t)PAIR * SCAN_RING * sizeof(unsigned) +
                       (size_t)PAIR * n * sizeof(int);
    auto *k = greedy_scan_kernel<PAIR>;
    if (lds > 48 * 1024 && lds > configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        configured = lds;
    }
    SRCNN_LAUNCH(k, dim3(nb / PAIR), dim3(SCAN_T * PAIR), lds, st, mask, zero_word, n, cb, n_valid, keep_out,
                       num_out, stop_after);
}

// paired != 0: problems (2b, 2b+1) are scanned together and stop once their keep lists share `stop_after` boxes
static int launch_nms(int *keep_out, const float *dets, int *num_out, const int *n_valid, int nb, int n,
                      int dim, float thresh, void *ws, size_t ws_bytes, hipStream_t st, int paired = 0,
                      int stop_after = 0)
{
    SRCNN_REQUIRE(nb >= 0 && n >= 0 && dim >= 4, "bad sizes");
    SRCNN_REQUIRE(n <= 64 * 64 * 4, "n > 16384 boxes per problem not supported");
    SRCNN_REQUIRE(!paired || nb % 2 == 0, "paired NMS needs an even number of problems");
    if (nb == 0) return SRCNN_OK;
    if (n == 0) {
        SRCNN_HIP_TRY(memset_async(num_out, 0, sizeof(int) * nb, st));
        return SRCNN_OK;
    }
    const int cb = cdiv(n, 64);
    const size_t words = (size_t)nb * n * cb;                       // + one zero word behind the mask
    const size_t need = (words + 1) * sizeof(unsigned long long);
    if (ws == nullptr || ws_bytes < need) {
        set_error("nms: workspace too small (%zu < %zu)", ws_bytes, need);
        return SRCNN_ERR_WORKSPACE;
    }
    SRCNN_REQUIRE((size_t)(paired ? 2 : 1) * (SCAN_RING + (size_t)n) * 4 + 128 <= 160 * 1024, "n too large for the LDS kept list");
    auto *mask = static_cast<unsigned long long *>(ws);
    const int skip = debug_skip_mask();
    if (!(skip & 4))
        SRCNN_LAUNCH(pair_mask_kernel, dim3(cb, cdiv(cb, PM_ROWS), nb), dim3(64 * PM_ROWS), 0, st, dets, n, dim, thresh, mask, cb, mask + words);
    if (skip & 8) return check_launch("nms");
    if (paired) launch_scan<2>(mask, mask + words, nb, n, cb, n_valid, keep_out, num_out, stop_after, st);
    else launch_scan<1>(mask, mask + words, nb, n, cb, n_valid, keep_out, num_out, 0, st);
    return check_launch("nms");
}

int nms_pairs_until(int *keep_out, const float *dets, int *num_out, const int *n_valid, int nb, int n, int dim,
                    float thresh, void *ws, size_t ws_bytes, int stop_after, hipStream_t st)
{
    return launch_nms(keep_out, dets, num_out, n_valid, nb, n, dim, thresh, ws, ws_bytes, st, 1, stop_after);
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
    return srcnn::align_up((size_t)n * srcnn::cdiv(n, 64) * 8 + 8, 256);
}

size_t srcnn_nms_batched_workspace_bytes(int nb, int n)
{
    if (n <= 0 || nb <= 0) return 256;
    return srcnn::align_up((size_t)nb * n * srcnn::cdiv(n, 64) * 8 + 8, 256);
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
