// Stereo RPN scoring and the proposal layer, fully on the device (gfx950).
//
// Reference: rpn/stereo_rpn.py:81-91 (pair softmax quirk + NHWC flatten),
// rpn/proposal_layer.py:42-145, rpn/generate_anchors.py:112-173,
// rpn/bbox_transform.py:79-104,177-185.
//
// What the reference does on the host (anchors in numpy every call + H2D, two NMS
// round-trips with a 4.5 MB mask D2H each, np.intersect1d on the CPU) is here five
// device launches with no host synchronisation:
//   1. tk_* kernels         exact top-K by (score desc, index asc): chip-wide radix select on the
//                           order-preserving score key, tie-break select on the index, compaction
//                           and a chip-wide rank sort of the K survivors.
//                           == torch.sort(descending, stable)[:K]  (proposal_layer.py:96,111-115)
//   2. gather_decode_kernel anchors recomputed analytically in float64 (bit-equal to the
//                           numpy anchors cast to float32), left/right decode + clip.
//   3-4. batched NMS        (nms.hip) over 2*B problems, left/right in lockstep with joint early exit.
//   5. intersect_pad_kernel sorted-set intersection, first `post` rows, zero padding.
#include "common.h"

namespace srcnn {

// ------------------------------------------------------------------ RPN scoring
// head rows: [0,6) cls logits, [6,24) deltas.  probs index (b, off + loc*3 + a, {0,1}).
__global__ void rpn_score_kernel(const float *__restrict__ head, int B, int hw, int hcs, float *__restrict__ probs,
                                 float *__restrict__ deltas, int level_off, int a_total)
{
    const int total = B * hw;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += blockDim.x * gridDim.x) {
        const int b = idx / hw, loc = idx - b * hw;
        const float *h = head + (size_t)idx * hcs;
        float pr[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // softmax over the channel pair (c, c+3)  (stereo_rpn.py:81-83)
            const float s0 = h[c], s1 = h[c + 3];
            const float m = fmaxf(s0, s1);
            const float e0 = expf(s0 - m), e1 = expf(s1 - m);
            const float sum = e0 + e1;
            pr[c] = e0 / sum;
            pr[c + 3] = e1 / sum;
        }
        const size_t base = (size_t)b * a_total + level_off + (size_t)loc * 3;
        float *po = probs + base * 2;      // NHWC flatten pairs CONSECUTIVE channels (:89-90)
#pragma unroll
        for (int c = 0; c < 6; ++c) po[c] = pr[c];
        float *dl = deltas + base * 6;
#pragma unroll
        for (int c = 0; c < 18; ++c) dl[c] = h[6 + c];
    }
}

// All pyramid levels in ONE launch (the five per-level launches sat at the ~5 us launch floor): the level of a location is found
// from the cumulative location counts; per location the arithmetic is rpn_score_kernel's, expression for expression.
struct RpnLevels {
    const float *head[5];
    int cum[6];        // cumulative locations per image: level l owns [cum[l], cum[l + 1])
    int n;
};

__global__ void rpn_score_levels_kernel(RpnLevels lv, int B, int hcs, float *__restrict__ probs, float *__restrict__ deltas,
                                        int a_total)
{
    const int per = lv.cum[lv.n];
    const int total = B * per;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += blockDim.x * gridDim.x) {
        const int b = idx / per, g = idx - b * per;
        int l = 0;
        while (l + 1 < lv.n && g >= lv.cum[l + 1]) ++l;
        const int hw = lv.cum[l + 1] - lv.cum[l], loc = g - lv.cum[l];
        const float *h = lv.head[l] + ((size_t)b * hw + loc) * hcs;
        float pr[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s0 = h[c], s1 = h[c + 3];
            const float m = fmaxf(s0, s1);
            const float e0 = expf(s0 - m), e1 = expf(s1 - m);
            const float sum = e0 + e1;
            pr[c] = e0 / sum;
            pr[c + 3] = e1 / sum;
        }
        const size_t base = (size_t)b * a_total + ((size_t)lv.cum[l] + loc) * 3;     // anchors of a level start at 3 x its first location
        float *po = probs + base * 2;
#pragma unroll
        for (int c = 0; c < 6; ++c) po[c] = pr[c];
        float *dl = deltas + base * 6;
#pragma unroll
        for (int c = 0; c < 18; ++c) dl[c] = h[6 + c];
    }
}

// The same from the PARTIAL sums the RPN conv's fused head leaves (conv_f16s.hip, HEAD 2; srcnn_conv_desc.head_wf): level l has
// nparts[l] planes of (B * hw_l, 24) floats -- one per (eye, N tile) of its launch --, added here in plane order, then the bias.
struct RpnParts {
    const float *part[5];
    long long plane[5];     // floats per plane
    int nparts[5];
    int cum[6];
    int n;
};

__global__ void rpn_score_parts_kernel(RpnParts lv, int B, const float *__restrict__ bias, float *__restrict__ probs,
                                       float *__restrict__ deltas, int a_total)
{
    const int per = lv.cum[lv.n];
    const int total = B * per;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += blockDim.x * gridDim.x) {
        const int b = idx / per, g = idx - b * per;
        int l = 0;
        while (l + 1 < lv.n && g >= lv.cum[l + 1]) ++l;
        const int hw = lv.cum[l + 1] - lv.cum[l], loc = g - lv.cum[l];
        const float *p0 = lv.part[l] + ((size_t)b * hw + loc) * 24;
        float h[24];
#pragma unroll
        for (int c = 0; c < 24; c += 4) {
            float4 s = *reinterpret_cast<const float4 *>(p0 + c);
            for (int q = 1; q < lv.nparts[l]; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(p0 + (size_t)q * lv.plane[l] + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            h[c] = s.x + bias[c]; h[c + 1] = s.y + bias[c + 1]; h[c + 2] = s.z + bias[c + 2]; h[c + 3] = s.w + bias[c + 3];
        }
        float pr[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s0 = h[c], s1 = h[c + 3];
            const float m = fmaxf(s0, s1);
            const float e0 = expf(s0 - m), e1 = expf(s1 - m);
            const float sum = e0 + e1;
            pr[c] = e0 / sum;
            pr[c + 3] = e1 / sum;
        }
        const size_t base = (size_t)b * a_total + ((size_t)lv.cum[l] + loc) * 3;
        float *po = probs + base * 2;
#pragma unroll
        for (int c = 0; c < 6; ++c) po[c] = pr[c];
        float *dl = deltas + base * 6;
#pragma unroll
        for (int c = 0; c < 18; ++c) dl[c] = h[6 + c];
    }
}

// ------------------------------------------------------------------ top-K + sort
__device__ __forceinline__ unsigned score_key(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float -> larger key
}

constexpr int TK_MAXK = 8192;
constexpr int TK_BINS = 2048;      // 11-bit radix digits

// Exact top-K by (score desc, index asc) == torch.sort(descending, stable)[:K], as a short chain of
// chip-wide launches (the first version ran the whole selection in ONE workgroup: 0.48 ms):
//   3 x { tk_hist (grid-wide LDS histograms of an 11-bit digit of the order-preserving score key)
//         -> tk_pick (one wave walks the 2048 bins from the top) }     -> key T of the K-th score
//   2 x { tk_hist on the INDEX digits of the elements with key == T -> tk_pick from the bottom }
//         (skipped on the device when every tie is taken)              -> largest tie index taken
//   tk_compact (grid-wide, wave-aggregated append)  ->  tk_rank (chip-wide rank sort of the <= 8192 unique keys)
struct TkState {
    unsigned prefix, remaining, ties, T, iprefix, idx_limit, need_tb, count;
};

__device__ void tk_pick_wave(TkState *state, unsigned *__restrict__ hist, int b, int lane, int kind, int shift, int last,
                             int first, unsigned K);

// LDS histogram increment of the lanes that `take`.  The scores of a frame cluster: every pass but the first looks at ONE digit
// prefix, and a freshly initialised RPN puts all of them into one bin of the first pass as well -- 64 lanes adding to one LDS
// address serialise (25 us per pass on the synthetic frames of bench.py).  The lanes that agree with the first taker's bin add
// once, together; the others use plain atomics.  The histogram is the same either way.
__device__ __forceinline__ void tk_hist_add(unsigned *lh, bool take, unsigned bin)
{
    const unsigned long long todo = __ballot(take);
    if (!todo) return;
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned lb = (unsigned)__shfl((int)bin, leader);
    const bool same = take && bin == lb;
    const unsigned long long sm = __ballot(same);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[lb], (unsigned)__popcll(sm));
    if (take && !same) atomicAdd(&lh[bin], 1u);
}

// One radix pass = ONE launch: grid-wide LDS histograms of a digit, and the workgroup that finishes last (arrival counter behind a
// device-scope fence) walks the 2048 bins with its first wave -- what used to be the separate one-wave tk_pick launch.  `first`:
// the state's `remaining` is still the memset's zero and stands for K (the tk_set_remaining launch is gone as well).
__global__ __launch_bounds__(256) void tk_hist_kernel(const float *__restrict__ probs, int A, TkState *state,
                                                      unsigned *__restrict__ hist, unsigned *__restrict__ arrivals, int kind,
                                                      int shift, int width, unsigned mask_above, int last, int first, unsigned K)
{
    __shared__ unsigned lh[TK_BINS];
    __shared__ int s_last;
    const int b = blockIdx.y;
    const TkState st = state[b];
    if (kind == 1 && !st.need_tb) return;
    for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const float *sc = probs + (size_t)b * A * 2 + 1;
    const unsigned dmask = (1u << width) - 1u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A; i += gridDim.x * blockDim.x) {
        const unsigned k = score_key(sc[(size_t)i * 2]);
        if (kind == 0) tk_hist_add(lh, (k & mask_above) == st.prefix, (k >> shift) & dmask);
        else tk_hist_add(lh, k == st.T && (((unsigned)i) & mask_above) == st.iprefix, ((unsigned)i >> shift) & dmask);
    }
    __syncthreads();
    unsigned *gh = hist + (size_t)b * TK_BINS;
    // Merge + arrival WITHOUT a device-scope fence: __threadfence() is a release at agent scope, which on this chip writes the
    // XCD's whole L2 back (buffer_wbl2) -- with other forwards in flight that is megabytes of THEIR dirty conv outputs, twice per
    // workgroup, 256 workgroups, five passes: 116 us per forward in the four-in-flight mix (profiles/skip_probe_r05.txt).
    // Everything the workgroups exchange goes through agent-scope atomics instead, which are performed at the coherence point:
    // a wave waits until its own merges have been performed (they return a value: vmcnt(0)), then the workgroup arrives; the
    // workgroup that arrives last reads the bins with agent-scope atomic loads.
    unsigned sink = 0;
    for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x)
        if (lh[i]) sink += __hip_atomic_fetch_add(&gh[i], lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink) : "memory");        // (the operand keeps the returning form of the atomics alive)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
    // under the HIP memory model relaxed atomics order nothing; that this hand-off needs no device-scope release is a
    // property of gfx950 (agent-scope atomics and sc1 loads are performed at the cross-XCD coherence point; an atomic that has
    // returned has been performed).  The library builds for gfx950 only -- any other target must not compile this silently.
#error "tk_hist_kernel's fence-free grid hand-off is validated for gfx950 only (tests/test_host_logic.py checks its ISA)"
#endif
    __syncthreads();
    // the last workgroup to arrive picks the digit (every image of the batch has its own arrival counter)
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&arrivals[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 64) tk_pick_wave(state, hist, b, threadIdx.x, kind, shift, last, first, K);
    if (threadIdx.x == 0) arrivals[b] = 0;             // ready for the next pass (next launch on this stream)
}

// one wave per image: lane l owns bins [32l, 32l+32)
__device__ void tk_pick_wave(TkState *state, unsigned *__restrict__ hist, int b, int lane, int kind, int shift, int last,
                             int first, unsigned K)
{
    TkState st = state[b];
    unsigned *gh = hist + (size_t)b * TK_BINS;
    if (first) st.remaining = K;
    unsigned mine[32], sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {       // agent-scope loads: the other workgroups' merges were atomics performed at the coherence point
        mine[i] = __hip_atomic_load(&gh[lane * 32 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += mine[i];
        gh[lane * 32 + i] = 0;
    }
    // position of lane in walking order: kind 0 walks from the top bin down, kind 1 from the bottom up
    const int ord = kind == 0 ? 63 - lane : lane;
    unsigned before = 0;   // elements in lanes walked before mine
    for (int o = 0; o < 64; ++o) {
        const unsigned s_o = __shfl(sum, kind == 0 ? 63 - o : o);
        if (o < ord) before += s_o;
    }
    const unsigned rem = st.remaining;
    const bool here = before < rem && rem <= before + sum;
    if (here) {
        unsigned cum = before;
        int d = 0;
        unsigned h = 0;
        // statically indexed walk (no dynamically indexed register array: hipcc lowers that to GPR-index mode, the form most
        // exposed to the lane-quarter effect of profiles/dense_align_repeatability_r02.txt -- and this lane may be one of 48-63)
        bool done = false;
        if (kind == 0) {
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                const unsigned hv = mine[i];
                const bool stop = !done && cum + hv >= rem;
                if (stop) { d = i; h = hv; }
                if (!done && !stop) cum += hv;
                done = done || stop;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const unsigned hv = mine[i];
                const bool stop = !done && cum + hv >= rem;
                if (stop) { d = i; h = hv; }
                if (!done && !stop) cum += hv;
                done = done || stop;
            }
        }
        const unsigned digit = (unsigned)(lane * 32 + d);
        st.remaining = rem - cum;
        st.ties = h;
        if (kind == 0) {
            st.prefix |= digit << shift;
            if (last) {
                st.T = st.prefix;
                st.need_tb = st.remaining < st.ties ? 1u : 0u;
                st.iprefix = 0;
                st.idx_limit = 0xFFFFFFFFu;
            }
        } else {
            st.iprefix |= digit << shift;
            if (last) st.idx_limit = st.iprefix;
        }
        state[b] = st;
    }
}

__global__ __launch_bounds__(256) void tk_compact_kernel(const float *__restrict__ probs, int A, TkState *state,
                                                         unsigned long long *__restrict__ cand, int cap)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const TkState st = state[b];
    const float *sc = probs + (size_t)b * A * 2 + 1;
    unsigned long long *out = cand + (size_t)b * cap;
    const int stride = gridDim.x * blockDim.x;
    for (int i0 = blockIdx.x * blockDim.x; i0 < A; i0 += stride) {
        const int i = i0 + threadIdx.x;
        bool take = false;
        unsigned k = 0;
        if (i < A) {
            k = score_key(sc[(size_t)i * 2]);
            take = (k > st.T) || (k == st.T && (unsigned)i <= st.idx_limit);
        }
        const unsigned long long bal = __ballot(take);
        unsigned base = 0;
        if (lane == 0 && bal) base = atomicAdd(&state[b].count, (unsigned)__popcll(bal));
        base = __shfl(base, 0);
        if (take) {
            const unsigned pos = base + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (pos < (unsigned)cap) out[pos] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
}

// Rank sort of the selected candidates on the whole chip: keys are unique (the low word holds the anchor index), so
// the rank of a key = the number of keys greater than it.  A workgroup stages all keys in LDS and ranks 64 of them:
// 16 threads per key, each counting over an interleaved 1/16 of the keys (conflict-free 16 x 8-byte LDS reads,
// broadcast across the four keys of a wavefront), then a 4-step shuffle reduction.  Descending key order =
// descending score, ascending index among ties (the stable order of proposal_layer.py:107).
__global__ __launch_bounds__(1024) void tk_rank_kernel(const unsigned long long *__restrict__ cand_in, int cap, int ksel,
                                                       int K, int *__restrict__ order_out)
{
    __shared__ unsigned long long keys[TK_MAXK];
    const int b = blockIdx.y, tid = threadIdx.x;
    const unsigned long long *in = cand_in + (size_t)b * cap;
    for (int i = tid; i < ksel; i += 1024) keys[i] = in[i];
    __syncthreads();
    const int sub = tid & 15, mi = blockIdx.x * 64 + (tid >> 4);
    const unsigned long long mine = mi < ksel ? keys[mi] : ~0ULL;
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    int j = sub;
    for (; j + 48 < ksel; j += 64) {
        r0 += keys[j] > mine;
        r1 += keys[j + 16] > mine;
        r2 += keys[j + 32] > mine;
        r3 += keys[j + 48] > mine;
    }
    for (; j < ksel; j += 16) r0 += keys[j] > mine;
    int rank = r0 + r1 + r2 + r3;
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    rank += __shfl_xor(rank, 4);
    rank += __shfl_xor(rank, 8);
    if (sub == 0 && mi < ksel) order_out[(size_t)b * K + rank] = (int)(0xFFFFFFFFu - (unsigned)(mine & 0xFFFFFFFFULL));
}

// ------------------------------------------------------------------ anchors + decode + clip
struct LevelTable {
    int off[6];        // anchor offset of each level (off[nl] = A)
    int w[5];          // feature width per level
    double stride[5];
    double aw[5][3], ah[5][3];   // anchor widths/heights per (level, ratio) in float64
    int nl;
};

__device__ __forceinline__ void decode_one(const float ax1, const float ay1, const float ax2, const float ay2,
                                           float dx, float dy, float dw, float dh, float wmax, float hmax, float *o)
{
    // bbox_transform.py:81-104, float32, one rounding per operation (no contraction)
    const float widths = ax2 - ax1 + 1.0f;
    const float heights = ay2 - ay1 + 1.0f;
    const float ctr_x = ax1 + 0.5f * widths;
    const float ctr_y = ay1 + 0.5f * heights;
    const float pcx = dx * widths + ctr_x;
    const float pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths;
    const float ph = expf(dh) * heights;
    o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.f), wmax);   // clip_boxes :177-185
    o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.f), hmax);
    o[2] = fminf(fmaxf(pcx + 0.5f * pw, 0.f), wmax);
    o[3] = fminf(fmaxf(pcy + 0.5f * ph, 0.f), hmax);
}

// dets: (B, 2, n, 5): [b][0] = left, [b][1] = right
__global__ void gather_decode_kernel(const float *__restrict__ probs, const float *__restrict__ deltas, int A, int B,
                                     int n, int K, const int *__restrict__ order, LevelTable lt,
                                     const float *__restrict__ im_info, float *__restrict__ dets)
{
    const int total = B * n;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += blockDim.x * gridDim.x) {
        const int b = idx / n, r = idx - b * n;
        const int ai = order[(size_t)b * K + r];
        int l = 0;
        while (l + 1 < lt.nl && ai >= lt.off[l + 1]) ++l;
        const int local = ai - lt.off[l];
        const int a = local % 3, loc = local / 3;
        const int x = loc % lt.w[l], y = loc / lt.w[l];
        const double cx = (double)x * lt.stride[l], cy = (double)y * lt.stride[l];
        const float ax1 = (float)(cx - 0.5 * lt.aw[l][a]), ay1 = (float)(cy - 0.5 * lt.ah[l][a]);
        const float ax2 = (float)(cx + 0.5 * lt.aw[l][a]), ay2 = (float)(cy + 0.5 * lt.ah[l][a]);
        const float *d = deltas + ((size_t)b * A + ai) * 6;
        const float wmax = im_info[b * 3 + 1] - 1.0f, hmax = im_info[b * 3 + 0] - 1.0f;
        const float score = probs[((size_t)b * A + ai) * 2 + 1];
        float *ol = dets + (((size_t)b * 2 + 0) * n + r) * 5;
        float *orr = dets + (((size_t)b * 2 + 1) * n + r) * 5;
        decode_one(ax1, ay1, ax2, ay2, d[0], d[1], d[2], d[3], wmax, hmax, ol);     // left : (dx, dy, dw, dh)
        decode_one(ax1, ay1, ax2, ay2, d[4], d[1], d[5], d[3], wmax, hmax, orr);    // right: (dx_r, dy, dw_r, dh)
        ol[4] = score;
        orr[4] = score;
    }
}

// ------------------------------------------------------------------ intersect + top `post` + pad
__global__ __launch_bounds__(1024) void intersect_pad_kernel(const int *__restrict__ keep, const int *__restrict__ num,
                                                             const float *__restrict__ dets, int n, int post,
                                                             float *__restrict__ rois_l, float *__restrict__ rois_r,
                                                             int *__restrict__ num_valid)
{
    __shared__ int wave_sum[16];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int *kl = keep + (size_t)(2 * b) * n, *kr = keep + (size_t)(2 * b + 1) * n;
    const int nl = num[2 * b], nr = num[2 * b + 1];
    const float *dl = dets + (size_t)(2 * b) * n * 5, *dr = dets + (size_t)(2 * b + 1) * n * 5;
    float *ol = rois_l + (size_t)b * post * 5, *orr = rois_r + (size_t)b * post * 5;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nl; i0 += 1024) {
        const int base = s_base;
        if (base >= post) break;
        const int i = i0 + tid;
        bool hit = false;
        int v = -1;
        if (i < nl) {
            v = kl[i];
            int lo = 0, hi = nr;   // lower_bound in the ascending right keep list
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (kr[mid] < v) lo = mid + 1; else hi = mid;
            }
            hit = lo < nr && kr[lo] == v;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wave_sum[wv] = __popcll(bal);
        __syncthreads();
        int pre = 0;
        for (int w = 0; w < wv; ++w) pre += wave_sum[w];
        const int pos = base + pre + __popcll(bal & ((1ULL << lane) - 1ULL));
        if (hit && pos < post) {
            const float *sl = dl + (size_t)v * 5, *sr = dr + (size_t)v * 5;
            ol[pos * 5 + 0] = (float)b; orr[pos * 5 + 0] = (float)b;
#pragma unroll
            for (int c = 0; c < 4; ++c) { ol[pos * 5 + 1 + c] = sl[c]; orr[pos * 5 + 1 + c] = sr[c]; }
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_sum[w];
            s_base = base + tot;
        }
        __syncthreads();
    }
    const int count = min(s_base, post);
    for (int p = count + tid; p < post; p += 1024) {   // proposal_layer.py:98-99,139-143
        ol[p * 5 + 0] = (float)b; orr[p * 5 + 0] = (float)b;
#pragma unroll
        for (int c = 1; c < 5; ++c) { ol[p * 5 + c] = 0.f; orr[p * 5 + c] = 0.f; }
    }
    if (tid == 0 && num_valid) num_valid[b] = count;
}

struct ProposalLayout {
    size_t state, hist, cand, order, dets, keep, num, nms, total;
};

static ProposalLayout proposal_layout(int B, int n, int K)
{
    ProposalLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    L.state = take((size_t)B * sizeof(TkState));
    L.hist = take((size_t)B * (TK_BINS + 64) * sizeof(unsigned));     // contiguous with state: zeroed by one memset; the words behind the
                                                                       // B histograms are the radix passes' arrival counters
    L.cand = take((size_t)B * TK_MAXK * sizeof(unsigned long long));
    L.order = take((size_t)B * K * sizeof(int));
    L.dets = take((size_t)B * 2 * n * 5 * sizeof(float));
    L.keep = take((size_t)B * 2 * n * sizeof(int));
    L.num = take((size_t)B * 2 * sizeof(int));
    L.nms = take(srcnn_nms_batched_workspace_bytes(2 * B, n));
    L.total = off;
    return L;
}

}  // namespace srcnn

extern "C" {

int srcnn_rpn_score(const float *head, int B, int hw, int head_cstride, float *probs, float *deltas,
                    int level_offset, int num_anchors_total, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(head && probs && deltas && B > 0 && hw > 0 && head_cstride >= 24, "bad args");
    const int total = B * hw;
    SRCNN_LAUNCH(rpn_score_kernel, dim3(std::min(cdiv(total, 256), 4096)), dim3(256), 0, as_stream(stream), head,
                       B, hw, head_cstride, probs, deltas, level_offset, num_anchors_total);
    return check_launch("srcnn_rpn_score");
}

int srcnn_rpn_score_levels(const float *const *heads, const int *level_hw, int nlevels, int B, int head_cstride, float *probs,
                           float *deltas, int num_anchors_total, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(heads && level_hw && probs && deltas && B > 0 && nlevels >= 1 && nlevels <= 5 && head_cstride >= 24, "bad args");
    RpnLevels lv;
    lv.n = nlevels;
    lv.cum[0] = 0;
    for (int l = 0; l < 5; ++l) {
        lv.head[l] = l < nlevels ? heads[l] : nullptr;
        SRCNN_REQUIRE(l >= nlevels || (heads[l] && level_hw[l] > 0), "null head / empty level");
        lv.cum[l + 1] = lv.cum[l] + (l < nlevels ? level_hw[l] : 0);
    }
    SRCNN_REQUIRE(3 * lv.cum[nlevels] == num_anchors_total, "num_anchors_total must be 3 x the locations of all levels");
    const int total = B * lv.cum[nlevels];
    SRCNN_LAUNCH(rpn_score_levels_kernel, dim3(std::min(cdiv(total, 256), 4096)), dim3(256), 0, as_stream(stream), lv, B,
                       head_cstride, probs, deltas, num_anchors_total);
    return check_launch("srcnn_rpn_score_levels");
}

int srcnn_rpn_score_parts(const float *const *parts, const int *nparts, const long long *plane_floats, const int *level_hw,
                          int nlevels, int B, const float *bias24, float *probs, float *deltas, int num_anchors_total,
                          srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(parts && nparts && plane_floats && level_hw && bias24 && probs && deltas && B > 0 && nlevels >= 1 && nlevels <= 5,
                  "bad args");
    RpnParts lv;
    lv.n = nlevels;
    lv.cum[0] = 0;
    for (int l = 0; l < 5; ++l) {
        const bool on = l < nlevels;
        SRCNN_REQUIRE(!on || (parts[l] && nparts[l] >= 1 && level_hw[l] > 0 && plane_floats[l] >= (long long)B * level_hw[l] * 24 &&
                              (reinterpret_cast<size_t>(parts[l]) & 15) == 0 && plane_floats[l] % 4 == 0),
                      "null / misaligned partial planes, empty level, or planes smaller than B x hw x 24 floats");
        lv.part[l] = on ? parts[l] : nullptr;
        lv.nparts[l] = on ? nparts[l] : 0;
        lv.plane[l] = on ? plane_floats[l] : 0;
        lv.cum[l + 1] = lv.cum[l] + (on ? level_hw[l] : 0);
    }
    SRCNN_REQUIRE(3 * lv.cum[nlevels] == num_anchors_total, "num_anchors_total must be 3 x the locations of all levels");
    const int total = B * lv.cum[nlevels];
    SRCNN_LAUNCH(rpn_score_parts_kernel, dim3(std::min(cdiv(total, 128), 4096)), dim3(128), 0, as_stream(stream), lv, B, bias24, probs,
                       deltas, num_anchors_total);
    return check_launch("srcnn_rpn_score_parts");
}

size_t srcnn_proposal_workspace_bytes(int B, int num_anchors, int pre_nms, int post_nms)
{
    (void)post_nms;
    const int n = pre_nms > 0 && pre_nms < num_anchors ? pre_nms : num_anchors;
    return srcnn::proposal_layout(B, n, n).total;
}

int srcnn_proposal_workspace_layout(int B, int num_anchors, int pre_nms, size_t *offsets, int n_offsets)
{
    using namespace srcnn;
    SRCNN_REQUIRE(offsets && n_offsets >= 5 && B > 0 && num_anchors > 0, "bad args");
    const int n = pre_nms > 0 && pre_nms < num_anchors ? pre_nms : num_anchors;
    const ProposalLayout L = proposal_layout(B, n, n);
    offsets[0] = L.order; offsets[1] = L.dets; offsets[2] = L.keep; offsets[3] = L.num; offsets[4] = L.total;
    return SRCNN_OK;
}

int srcnn_proposal_layer(const float *probs, const float *deltas, int B, int num_anchors, const int *level_hw_host,
                         int nlevels, const float *im_info, int pre_nms, int post_nms, float nms_thresh,
                         float *rois_left, float *rois_right, int *num_valid, void *workspace,
                         size_t workspace_bytes, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(probs && deltas && im_info && rois_left && rois_right, "null pointer");
    SRCNN_REQUIRE(nlevels >= 1 && nlevels <= 5 && B > 0 && post_nms > 0, "bad sizes");
    static const double scales[5] = {32, 64, 128, 256, 512};      // cfg.FPN_ANCHOR_SCALES (config.py:216)
    static const double strides[5] = {4, 8, 16, 32, 64};          // cfg.FPN_FEAT_STRIDES  (config.py:219)
    static const double ratios[3] = {0.5, 1, 2};                  // cfg.ANCHOR_RATIOS     (config.py:210)
    LevelTable lt;
    lt.nl = nlevels;
    int off = 0;
    for (int l = 0; l < nlevels; ++l) {
        lt.off[l] = off;
        lt.w[l] = level_hw_host[2 * l + 1];
        lt.stride[l] = strides[l];
        for (int r = 0; r < 3; ++r) {   // generate_anchors.py:128-129
            lt.ah[l][r] = scales[l] / std::sqrt(ratios[r]);
            lt.aw[l][r] = scales[l] * std::sqrt(ratios[r]);
        }
        off += level_hw_host[2 * l] * level_hw_host[2 * l + 1] * 3;
    }
    for (int l = nlevels; l <= 5; ++l) lt.off[l] = off;
    SRCNN_REQUIRE(off == num_anchors, "num_anchors does not match the level shapes");
    // proposal_layer.py:111: keep everything when pre_nms >= numel
    const int n = pre_nms > 0 && pre_nms < num_anchors ? pre_nms : num_anchors;
    SRCNN_REQUIRE(n <= TK_MAXK, "pre_nms_topN > 8192 not supported");
    ProposalLayout L = proposal_layout(B, n, n);
    if (!workspace || workspace_bytes < L.total) {
        set_error("srcnn_proposal_layer: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return SRCNN_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(workspace);
    int *order = reinterpret_cast<int *>(ws + L.order);
    float *dets = reinterpret_cast<float *>(ws + L.dets);
    int *keep = reinterpret_cast<int *>(ws + L.keep);
    int *num = reinterpret_cast<int *>(ws + L.num);
    hipStream_t st = as_stream(stream);
    SRCNN_REQUIRE(num_anchors < (1 << 22), "more than 4M anchors not supported");
    {
        TkState *state = reinterpret_cast<TkState *>(ws + L.state);
        unsigned *hist = reinterpret_cast<unsigned *>(ws + L.hist);
        unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + L.cand);
        const int skip = debug_skip_mask();       // measurement hook (srcnn_debug_skip_mask): 0 in production
        if (!(skip & (1 | 128))) SRCNN_HIP_TRY(memset_async(ws + L.state, 0, L.cand - L.state, st));
        const int ksel = n;
        const int G = 256;
        static const int sshift[3] = {21, 10, 0}, swidth[3] = {11, 11, 10};
        static const unsigned smask[3] = {0u, 0xFFE00000u, 0xFFFFFC00u};
        unsigned *arrivals = hist + (size_t)B * TK_BINS;
        for (int p = 0; p < 3 && !(skip & 1); ++p)
            SRCNN_LAUNCH(tk_hist_kernel, dim3(G, B), dim3(256), 0, st, probs, num_anchors, state, hist, arrivals, 0,
                               sshift[p], swidth[p], smask[p], p == 2 ? 1 : 0, p == 0 ? 1 : 0, (unsigned)ksel);
        static const int ishift[2] = {11, 0};
        static const unsigned imask[2] = {0u, 0xFFFFF800u};
        for (int p = 0; p < 2 && !(skip & 1); ++p)
            SRCNN_LAUNCH(tk_hist_kernel, dim3(G, B), dim3(256), 0, st, probs, num_anchors, state, hist, arrivals, 1,
                               ishift[p], 11, imask[p], p == 1 ? 1 : 0, 0, (unsigned)ksel);
        if (!(skip & 32)) SRCNN_LAUNCH(tk_compact_kernel, dim3(G, B), dim3(256), 0, st, probs, num_anchors, state, cand, TK_MAXK);
        if (!(skip & 64)) SRCNN_LAUNCH(tk_rank_kernel, dim3(cdiv(ksel, 64), B), dim3(1024), 0, st, cand, TK_MAXK, ksel, n, order);
    }
    if (!(debug_skip_mask() & 2))
        SRCNN_LAUNCH(gather_decode_kernel, dim3(cdiv(B * n, 256)), dim3(256), 0, st, probs, deltas, num_anchors, B,
                           n, n, order, lt, im_info, dets);
    int rc = check_launch("proposal: select/decode");
    if (rc != SRCNN_OK) return rc;
    // left/right problems of an image scanned in lockstep; stop once `post_nms` boxes survive in both
    rc = nms_pairs_until(keep, dets, num, nullptr, 2 * B, n, 5, nms_thresh, ws + L.nms, workspace_bytes - L.nms,
                         post_nms, st);
    if (rc != SRCNN_OK) return rc;
    if (!(debug_skip_mask() & 16))
        SRCNN_LAUNCH(intersect_pad_kernel, dim3(B), dim3(1024), 0, st, keep, num, dets, n, post_nms, rois_left,
                           rois_right, num_valid);
    return check_launch("proposal: intersect");
}

}  // extern "C"
