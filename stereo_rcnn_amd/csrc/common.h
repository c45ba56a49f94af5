// Shared helpers for libsrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <tuple>
#include <type_traits>
#include <utility>
#include "../../include/srcnn_hip.h"

namespace srcnn {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(srcnn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SRCNN_ERR_HIP;
    }
    return SRCNN_OK;
}

#define SRCNN_HIP_TRY(expr)                                                     \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            ::srcnn::set_error("%s failed: %s", #expr, hipGetErrorString(_e));  \
            return SRCNN_ERR_HIP;                                               \
        }                                                                       \
    } while (0)

#define SRCNN_REQUIRE(cond, msg)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            ::srcnn::set_error("%s: %s", __func__, msg); \
            return SRCNN_ERR_ARG;                        \
        }                                                \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// carve 256-byte aligned chunks out of a caller-provided workspace
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base(static_cast<char *>(p)), off(0) {}
    template <typename T>
    T *take(size_t count)
    {
        T *r = reinterpret_cast<T *>(base + off);
        off += align_up(count * sizeof(T), 256);
        return r;
    }
};

// ---- launch recording (core.hip): every kernel launch and async memset of the library goes through launch_kernel() /
// memset_async().  Normally they launch; while a srcnn_program is recording on the calling thread they append a node
// (function, geometry, stream, a private copy of the by-value arguments) to it instead, and srcnn_program_run() replays
// the whole list from C -- the forward's ~230 launches without a Python frame or a ctypes call in between.
struct Program;
Program *recording_program();
void program_add_kernel(Program *p, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, void *blob,
                        void (*destroy)(void *), void **argv, int argc);
void program_add_memset(Program *p, void *dst, int value, size_t bytes, hipStream_t st);

template <typename Tup, size_t... I>
inline void tuple_arg_pointers(Tup &t, void **argv, std::index_sequence<I...>)
{
    ((argv[I] = const_cast<void *>(static_cast<const void *>(&std::get<I>(t)))), ...);
}

template <typename... P, typename... A>
inline void launch_kernel(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A &&...a)
{
    static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
    using Tup = std::tuple<std::remove_cv_t<std::remove_reference_t<P>>...>;
    constexpr int N = (int)sizeof...(P);
    void *argv[N > 0 ? N : 1];
    Program *prog = recording_program();
    if (!prog) {
        Tup t(static_cast<std::remove_cv_t<std::remove_reference_t<P>>>(a)...);
        tuple_arg_pointers(t, argv, std::index_sequence_for<P...>{});
        (void)hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, argv, lds, st);
        return;
    }
    Tup *t = new Tup(static_cast<std::remove_cv_t<std::remove_reference_t<P>>>(a)...);
    tuple_arg_pointers(*t, argv, std::index_sequence_for<P...>{});
    program_add_kernel(prog, reinterpret_cast<const void *>(kernel), grid, block, lds, st, t,
                       [](void *q) { delete static_cast<Tup *>(q); }, argv, N);
}

inline hipError_t memset_async(void *dst, int value, size_t bytes, hipStream_t st)
{
    Program *prog = recording_program();
    if (!prog) return hipMemsetAsync(dst, value, bytes, st);
    program_add_memset(prog, dst, value, bytes, st);
    return hipSuccess;
}

#define SRCNN_LAUNCH(kernel, grid, block, lds, st, ...) ::srcnn::launch_kernel(kernel, dim3(grid), dim3(block), lds, st, __VA_ARGS__)

// nms.hip: batched NMS where problems (2b, 2b+1) are scanned in lockstep and stop once the intersection of their
// keep lists has `stop_after` entries (keep lists are then complete only up to that point).  Used by the proposal layer.
int nms_pairs_until(int *keep_out, const float *dets, int *num_out, const int *n_valid, int nb, int n, int dim,
                    float thresh, void *ws, size_t ws_bytes, int stop_after, hipStream_t st);

// debug hook: when set, conv_f16s workgroups write 16 x u64 {t_entry, t_setup, t_first_tile, t_loop, t_tile_in_lds,
// t_end (s_memtime, per-CU shader clocks), hw_id, 1 | xcc_id << 8, realtime_entry, realtime_end (100 MHz, chip-wide)}
// at stamp[16 * workgroup]; see tools/stamp_conv.py
unsigned long long *debug_stamp_buffer();
// ... or, while a stamp ARENA is set (core.hip), a region of this launch's own; nullptr = do not stamp
unsigned long long *debug_stamp_region(int wgs, int tag, int lds_bytes, int threads, int M, int N, int K);

int debug_skip_mask();          // core.hip: srcnn_debug_skip_mask (measurement hook of tools/skip_probe.py; 0 in production)

// profiling hooks (conv engine)
bool prof_enabled();
void prof_begin(hipStream_t s);
void prof_end(hipStream_t s, double flops);

}  // namespace srcnn
