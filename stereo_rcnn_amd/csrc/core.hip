// Library core: error text, version, conv-engine profiling hooks.
#include "conv_common.h"
#include <atomic>
#include <mutex>
#include <vector>

namespace srcnn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> pool;   // event pairs [2*i, 2*i+1]
    std::vector<double> flops;      // per recorded pair
    size_t used = 0;                // pairs in use
    double total_flops = 0;
    std::mutex mu;
};
static ProfState g_prof;

bool prof_enabled() { return g_prof.on; }

void prof_begin(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.used * 2 + 2 > g_prof.pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof.pool.push_back(a);
        g_prof.pool.push_back(b);
    }
    (void)hipEventRecord(g_prof.pool[g_prof.used * 2], s);
}

void prof_end(hipStream_t s, double flops)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.used * 2 + 2 > g_prof.pool.size()) return;
    (void)hipEventRecord(g_prof.pool[g_prof.used * 2 + 1], s);
    g_prof.flops.resize(g_prof.used + 1);
    g_prof.flops[g_prof.used] = flops;
    g_prof.used++;
}

// ------------------------------------------------------------------ recorded launch programs
struct ProgNode {
    enum Kind { KERNEL, MEMSET, RECORD, WAIT } kind;
    const void *fn = nullptr;
    dim3 grid, block;
    size_t lds = 0;
    int stream = 0;                 // index into Program::streams (0 = the main stream, substituted at run time)
    std::vector<void *> argv;       // pointers into `blob`
    void *blob = nullptr;
    void (*destroy)(void *) = nullptr;
    void *dst = nullptr;            // MEMSET
    int value = 0;
    size_t bytes = 0;
    int event = -1;                 // RECORD / WAIT
};

struct Program {
    std::vector<ProgNode> nodes;
    std::vector<hipStream_t> streams;      // as recorded; [0] = main
    std::vector<hipEvent_t> events;
    bool recording = false;
    ~Program()
    {
        for (auto &n : nodes)
            if (n.blob && n.destroy) n.destroy(n.blob);
        for (auto e : events) (void)hipEventDestroy(e);
    }
    int stream_index(hipStream_t s)
    {
        for (size_t i = 0; i < streams.size(); ++i)
            if (streams[i] == s) return (int)i;
        streams.push_back(s);
        return (int)streams.size() - 1;
    }
};

static thread_local Program *t_recording = nullptr;

Program *recording_program() { return t_recording; }

void program_add_kernel(Program *p, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, void *blob,
                        void (*destroy)(void *), void **argv, int argc)
{
    ProgNode n;
    n.kind = ProgNode::KERNEL;
    n.fn = fn;
    n.grid = grid;
    n.block = block;
    n.lds = lds;
    n.stream = p->stream_index(st);
    n.argv.assign(argv, argv + argc);
    n.blob = blob;
    n.destroy = destroy;
    p->nodes.push_back(std::move(n));
}

void program_add_memset(Program *p, void *dst, int value, size_t bytes, hipStream_t st)
{
    ProgNode n;
    n.kind = ProgNode::MEMSET;
    n.stream = p->stream_index(st);
    n.dst = dst;
    n.value = value;
    n.bytes = bytes;
    p->nodes.push_back(std::move(n));
}

static thread_local unsigned *t_bound_flag = nullptr;

unsigned *range_flag_word()
{
    if (t_bound_flag) return t_bound_flag;          // the caller's own word (srcnn_range_flag_bind): one per forward in flight
    static unsigned *word = nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!word) {
        if (hipMalloc(reinterpret_cast<void **>(&word), 256) != hipSuccess) return nullptr;
        (void)hipMemset(word, 0, 256);
    }
    return word;
}

static unsigned long long *g_stamp = nullptr;
unsigned long long *debug_stamp_buffer() { return g_stamp; }

// Stamp ARENA (debug): every conv_f16s launch issued -- or recorded into a launch program -- while it is set gets a region of
// its own (16 x u64 per workgroup), so that the launches of SEVERAL forwards in flight can be told apart after the run: the
// per-layer residency table of the regime the headline is measured in (tools/mix_layers.py).  A replayed program writes the
// regions its launches got at record time, every replay anew: after a run the arena holds the LAST execution of every launch.
__global__ void null_kernel(int) {}

std::atomic<int> g_debug_skip{0};       // srcnn_debug_skip_mask (measurement hook; 0 in production)
int debug_skip_mask() { return g_debug_skip.load(std::memory_order_relaxed); }

struct StampLogRow { long long off_words; int wgs, tag, lds_bytes, threads, M, N, K; };
static unsigned long long *g_arena = nullptr;
static size_t g_arena_words = 0, g_arena_cursor = 0;
static std::vector<StampLogRow> g_arena_log;
static std::mutex g_arena_mu;

unsigned long long *debug_stamp_region(int wgs, int tag, int lds_bytes, int threads, int M, int N, int K)
{
    if (!g_arena) return g_stamp;                               // legacy single-launch hook (tools/stamp_conv.py) or nullptr
    std::lock_guard<std::mutex> lk(g_arena_mu);
    const size_t need = (size_t)wgs * 16;
    if (g_arena_cursor + need > g_arena_words) return nullptr;  // arena full: this launch runs unstamped
    unsigned long long *r = g_arena + g_arena_cursor;
    g_arena_log.push_back({(long long)g_arena_cursor, wgs, tag, lds_bytes, threads, M, N, K});
    g_arena_cursor += need;
    return r;
}

}  // namespace srcnn

extern "C" {

// debug hook (not part of include/srcnn_hip.h): device buffer of 16 x u64 per workgroup, or NULL to switch off
SRCNN_API void srcnn_debug_set_stamp_buffer(void *buf) { srcnn::g_stamp = static_cast<unsigned long long *>(buf); }

// debug hooks of the stamp arena (see above): buf = device buffer of `words` u64 (zero it first), or NULL to switch off.
// srcnn_debug_stamp_log copies up to max_rows rows of 8 x int64 {offset in words, workgroups, layer tag, LDS bytes, threads, M, N, K}
// and returns the number of launches logged since the arena was set.
// debug hook (tools/skip_probe.py): launches of the proposal layer left out while the mask is set -- 1: top-K selection chain,
// 2: gather + decode, 4: pair mask, 8: greedy scan, 16: intersect + pad.  Their outputs keep the previous call's values; the
// step-time difference is what the launch costs inside the several-forwards-in-flight mix.  0 in production.
SRCNN_API void srcnn_debug_skip_mask(int mask) { srcnn::g_debug_skip.store(mask); }

// debug hook (tools/skip_probe.py): n empty one-wave kernels on `stream` -- what does a kernel boundary (dispatch, the release at
// its end, the acquire of the next launch) cost the OTHER forwards in flight?
SRCNN_API int srcnn_debug_null_launches(int n, srcnn_stream_t stream)
{
    for (int i = 0; i < n; ++i) SRCNN_LAUNCH(srcnn::null_kernel, dim3(1), dim3(64), 0, srcnn::as_stream(stream), i);
    return srcnn::check_launch("srcnn_debug_null_launches");
}

SRCNN_API void srcnn_debug_set_stamp_arena(void *buf, size_t words)
{
    std::lock_guard<std::mutex> lk(srcnn::g_arena_mu);
    srcnn::g_arena = static_cast<unsigned long long *>(buf);
    srcnn::g_arena_words = buf ? words : 0;
    srcnn::g_arena_cursor = 0;
    srcnn::g_arena_log.clear();
}

SRCNN_API int srcnn_debug_stamp_log(long long *rows, int max_rows)
{
    std::lock_guard<std::mutex> lk(srcnn::g_arena_mu);
    const int n = (int)srcnn::g_arena_log.size();
    for (int i = 0; i < n && i < max_rows; ++i) {
        const auto &r = srcnn::g_arena_log[i];
        long long *o = rows + (size_t)i * 8;
        o[0] = r.off_words; o[1] = r.wgs; o[2] = r.tag; o[3] = r.lds_bytes; o[4] = r.threads; o[5] = r.M; o[6] = r.N; o[7] = r.K;
    }
    return n;
}

int srcnn_version(void) { return 240; }   // 210: srcnn_stream_create*, srcnn_probe_placement, srcnn_conv_desc.head_* (appended fields)

int srcnn_range_flag_read(int reset)
{
    unsigned *w = srcnn::range_flag_word();
    if (!w) return SRCNN_ERR_HIP;
    unsigned v = 0;
    if (hipDeviceSynchronize() != hipSuccess) return SRCNN_ERR_HIP;     // non-blocking streams do not order with the copy below
    if (hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return SRCNN_ERR_HIP;
    if (reset && v) (void)hipMemset(w, 0, sizeof(v));
    return (int)v;
}

const void *srcnn_range_flag_device_word(void) { return srcnn::range_flag_word(); }

int srcnn_range_flag_bind(void *device_word)
{
    srcnn::t_bound_flag = static_cast<unsigned *>(device_word);
    return SRCNN_OK;
}

// ---- streams that own a hardware queue (include/srcnn_hip.h)
int srcnn_stream_create(int dedicated_queue, srcnn_stream_t *stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(stream, "null stream pointer");
    hipStream_t s = nullptr;
    hipError_t e;
    if (dedicated_queue) {
        // every CU enabled: the mask restricts nothing, it only takes the stream out of the pooled queues
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SRCNN_ERR_HIP;
        const unsigned words = ((unsigned)prop.multiProcessorCount + 31u) / 32u;
        unsigned mask[64];
        SRCNN_REQUIRE(words >= 1 && words <= 64, "unexpected CU count");
        for (unsigned i = 0; i < words; ++i) mask[i] = 0xffffffffu;
        e = hipExtStreamCreateWithCUMask(&s, words, mask);
    } else {
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
        set_error("srcnn_stream_create: stream creation failed");
        return SRCNN_ERR_HIP;
    }
    *stream = reinterpret_cast<srcnn_stream_t>(s);
    return SRCNN_OK;
}

int srcnn_stream_create_cu_mask(int words, const unsigned *mask, srcnn_stream_t *stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(stream && mask && words >= 1 && words <= 64, "null pointer or bad word count");
    unsigned any = 0;
    for (int i = 0; i < words; ++i) any |= mask[i];
    SRCNN_REQUIRE(any != 0, "empty CU mask");
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) {
        set_error("srcnn_stream_create_cu_mask: hipExtStreamCreateWithCUMask failed");
        return SRCNN_ERR_HIP;
    }
    *stream = reinterpret_cast<srcnn_stream_t>(s);
    return SRCNN_OK;
}

namespace srcnn {
__global__ void placement_probe_kernel(int *xcc, int *hw_id)
{
    if (threadIdx.x == 0) {
        xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // XCC_ID, bits 3:0
        hw_id[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID, all 32 bits
    }
    // stay resident for a moment so that the blocks spread over the CUs instead of reusing the first free one
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {
    }
}
}  // namespace srcnn

int srcnn_probe_placement(int blocks, int *xcc, int *hw_id, srcnn_stream_t stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(blocks > 0 && xcc && hw_id, "bad arguments");
    SRCNN_LAUNCH(placement_probe_kernel, blocks, 64, 0, as_stream(stream), xcc, hw_id);
    return check_launch("srcnn_probe_placement");
}

int srcnn_stream_destroy(srcnn_stream_t stream)
{
    if (!stream) return SRCNN_OK;
    return hipStreamDestroy(srcnn::as_stream(stream)) == hipSuccess ? SRCNN_OK : SRCNN_ERR_HIP;
}

// ---- launch programs (include/srcnn_hip.h)
void *srcnn_program_create(void) { return new srcnn::Program(); }

void srcnn_program_destroy(void *prog) { delete static_cast<srcnn::Program *>(prog); }

int srcnn_program_begin(void *prog, srcnn_stream_t main_stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(prog && !t_recording, "null program, or another program is recording on this thread");
    Program *p = static_cast<Program *>(prog);
    SRCNN_REQUIRE(p->nodes.empty(), "program already holds a recording");
    p->streams.assign(1, as_stream(main_stream));
    p->recording = true;
    t_recording = p;
    return SRCNN_OK;
}

int srcnn_program_end(void *prog)
{
    using namespace srcnn;
    SRCNN_REQUIRE(prog && t_recording == prog, "this program is not recording on this thread");
    t_recording = nullptr;
    static_cast<Program *>(prog)->recording = false;
    return SRCNN_OK;
}

int srcnn_program_recording(void) { return srcnn::t_recording != nullptr; }

int srcnn_program_record_event(void *prog, srcnn_stream_t stream)
{
    using namespace srcnn;
    if (!prog || t_recording != prog) {
        set_error("srcnn_program_record_event: this program is not recording on this thread");
        return SRCNN_ERR_ARG;
    }
    Program *p = static_cast<Program *>(prog);
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        set_error("srcnn_program_record_event: hipEventCreate failed");
        return SRCNN_ERR_HIP;
    }
    p->events.push_back(e);
    ProgNode n;
    n.kind = ProgNode::RECORD;
    n.stream = p->stream_index(as_stream(stream));
    n.event = (int)p->events.size() - 1;
    p->nodes.push_back(std::move(n));
    return (int)p->events.size() - 1;
}

int srcnn_program_wait_event(void *prog, srcnn_stream_t stream, int event_id)
{
    using namespace srcnn;
    SRCNN_REQUIRE(prog && t_recording == prog, "this program is not recording on this thread");
    Program *p = static_cast<Program *>(prog);
    SRCNN_REQUIRE(event_id >= 0 && event_id < (int)p->events.size(), "unknown event");
    ProgNode n;
    n.kind = ProgNode::WAIT;
    n.stream = p->stream_index(as_stream(stream));
    n.event = event_id;
    p->nodes.push_back(std::move(n));
    return SRCNN_OK;
}

int srcnn_program_size(void *prog) { return prog ? (int)static_cast<srcnn::Program *>(prog)->nodes.size() : 0; }

int srcnn_program_run(void *prog, srcnn_stream_t main_stream)
{
    using namespace srcnn;
    SRCNN_REQUIRE(prog, "null program");
    Program *p = static_cast<Program *>(prog);
    SRCNN_REQUIRE(!p->recording && !p->nodes.empty(), "program is empty or still recording");
    hipStream_t main_s = as_stream(main_stream);
    for (auto &n : p->nodes) {
        hipStream_t st = n.stream == 0 ? main_s : p->streams[n.stream];
        hipError_t e = hipSuccess;
        switch (n.kind) {
        case ProgNode::KERNEL: e = hipLaunchKernel(n.fn, n.grid, n.block, n.argv.data(), n.lds, st); break;
        case ProgNode::MEMSET: e = hipMemsetAsync(n.dst, n.value, n.bytes, st); break;
        case ProgNode::RECORD: e = hipEventRecord(p->events[n.event], st); break;
        case ProgNode::WAIT: e = hipStreamWaitEvent(st, p->events[n.event], 0); break;
        }
        if (e != hipSuccess) {
            set_error("srcnn_program_run: node failed: %s", hipGetErrorString(e));
            return SRCNN_ERR_HIP;
        }
    }
    return SRCNN_OK;
}

const char *srcnn_last_error(void) { return srcnn::g_err; }

int srcnn_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(srcnn::g_prof.mu);
    srcnn::g_prof.on = on != 0;
    srcnn::g_prof.used = 0;
    srcnn::g_prof.flops.clear();
    return SRCNN_OK;
}

int srcnn_prof_read_launches(float *ms, int max)
{
    using namespace srcnn;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (size_t i = 0; i < g_prof.used && (int)i < max; ++i) {
        SRCNN_HIP_TRY(hipEventSynchronize(g_prof.pool[2 * i + 1]));
        SRCNN_HIP_TRY(hipEventElapsedTime(&ms[i], g_prof.pool[2 * i], g_prof.pool[2 * i + 1]));
    }
    return (int)g_prof.used;
}

int srcnn_prof_read(double *conv_ms, double *conv_flops, long long *conv_launches)
{
    using namespace srcnn;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0, fl = 0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        SRCNN_HIP_TRY(hipEventSynchronize(g_prof.pool[2 * i + 1]));
        float t = 0;
        SRCNN_HIP_TRY(hipEventElapsedTime(&t, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]));
        ms += t;
        fl += g_prof.flops[i];
    }
    if (conv_ms) *conv_ms = ms;
    if (conv_flops) *conv_flops = fl;
    if (conv_launches) *conv_launches = (long long)g_prof.used;
    g_prof.used = 0;
    g_prof.flops.clear();
    return SRCNN_OK;
}

}  // extern "C"
