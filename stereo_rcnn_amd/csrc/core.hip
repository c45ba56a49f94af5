// Library core: error text, version, conv-engine profiling hooks.
#include "conv_common.h"
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

// library-owned, never freed: 4 KB of zeros that padded conv taps DMA from (conv_f16s.hip)
const void *zero_page()
{
    static void *page = nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!page) {
        if (hipMalloc(&page, 4096) != hipSuccess) return nullptr;
        (void)hipMemset(page, 0, 4096);
    }
    return page;
}

static unsigned long long *g_stamp = nullptr;
unsigned long long *debug_stamp_buffer() { return g_stamp; }

}  // namespace srcnn

extern "C" {

// debug hook (not part of include/srcnn_hip.h): device buffer of 16 x u64 per workgroup, or NULL to switch off
SRCNN_API void srcnn_debug_set_stamp_buffer(void *buf) { srcnn::g_stamp = static_cast<unsigned long long *>(buf); }

int srcnn_version(void) { return 100; }

const char *srcnn_last_error(void) { return srcnn::g_err; }

int srcnn_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(srcnn::g_prof.mu);
    srcnn::g_prof.on = on != 0;
    srcnn::g_prof.used = 0;
    srcnn::g_prof.flops.clear();
    return SRCNN_OK;
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
