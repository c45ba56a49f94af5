// dev probe (not product code): is the result of a double-precision-rate VALU instruction visible to the NEXT VALU instruction
// of the same wave in all 64 lanes (hardware interlock -- the compiler inserts nothing here), also while waves of another kernel
// keep the SIMD's matrix pipe busy?  Background: profiles/dense_align_repeatability_r02.txt
//   hipcc --offload-arch=gfx950 -O2 tools/probes/fp64_use_probe.hip -o /tmp/f64p && /tmp/f64p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: v_cvt_f32_f64 -> v_add_f32     1: v_fma_f64 -> v_cvt_f32_f64 -> v_mul_f32     2: v_mul_f64 -> v_add_f64 -> v_cvt_f32_f64
//      3: v_cvt_f64_f32 -> v_add_f64 -> v_cvt_f32_f64 -> v_mul_f32 (the shape of `(float)((double)x - 0.01)` in the kernel)
template <int MODE>
__global__ void probe_kernel(int iters, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    double x = 1.0 + 1e-3 * (blockIdx.x & 255);
    for (int i = 0; i < iters; ++i) {
        float out = 777.0f + i, want;                 // destination pre-loaded with a sentinel: a stale read is visible
        double t = 555.0 + i;
        const double y = 0.5 + 1e-4 * (i & 1023);
        if (MODE == 0) {
            float r = out;
            asm volatile("s_nop 4\n\tv_cvt_f32_f64 %0, %2\n\tv_add_f32 %1, %0, %0" : "+v"(r), "=v"(out) : "v"(x));
            want = (float)x + (float)x;
        } else if (MODE == 1) {
            float r = out;
            asm volatile("s_nop 4\n\tv_fma_f64 %2, %3, %4, %3\n\tv_cvt_f32_f64 %0, %2\n\tv_mul_f32 %1, %0, %0" : "+v"(r), "=v"(out), "+v"(t) : "v"(x), "v"(y));
            const float f = (float)__builtin_fma(x, y, x);
            want = f * f;
        } else if (MODE == 2) {
            double u = 333.0 + i;
            float r = out;
            asm volatile("s_nop 4\n\tv_mul_f64 %2, %4, %5\n\tv_add_f64 %3, %2, %4\n\tv_cvt_f32_f64 %0, %3\n\tv_add_f32 %1, %0, %0"
                         : "+v"(r), "=v"(out), "+v"(t), "+v"(u) : "v"(x), "v"(y));
            const float f = (float)(x * y + x);
            want = f + f;
        } else {
            float xs = (float)x, r = out;
            double u = 333.0 + i;
            asm volatile("s_nop 4\n\tv_cvt_f64_f32 %2, %4\n\tv_add_f64 %3, %2, %5\n\tv_cvt_f32_f64 %0, %3\n\tv_mul_f32 %1, %0, %0"
                         : "+v"(r), "=v"(out), "+v"(t), "+v"(u) : "v"(xs), "v"(y));
            const float f = (float)((double)xs + y);
            want = f * f;
        }
        asm volatile("" : "+v"(want));
        if (__float_as_uint(out) != __float_as_uint(want)) ++bad;
        x += 1e-6; if (x > 3.0) x = 1.0;
    }
    if (bad) atomicAdd(bad_lanes + lane, bad);
    if (threadIdx.x == 0) atomicAdd(counts, (unsigned long long)iters * (blockDim.x / 64));
}

__global__ __launch_bounds__(512) void mfma_load_kernel(int iters, float *sink)
{
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(bool with_mfma)
{
    unsigned long long *bad, *counts; float *sink;
    CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&counts, 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(counts, 0, 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (with_mfma) for (int k = 0; k < 80; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe_kernel<MODE>), dim3(2048), dim3(256), 0, s1, 20000, bad, counts);
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long c[2];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0}, sum = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    static const char *names[4] = {"v_cvt_f32_f64 -> v_add_f32", "v_fma_f64 -> v_cvt_f32_f64 -> v_mul_f32", "v_mul_f64 -> v_add_f64 -> v_cvt_f32_f64 -> v_add_f32",
                                   "v_cvt_f64_f32 -> v_add_f64 -> v_cvt_f32_f64 -> v_mul_f32"};
    printf("%-58s %-22s: %llu wrong lane results in %.3g wave executions; lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           names[MODE], with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)c[0], q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(counts)); CK(hipFree(sink));
}

int main()
{
    for (int m = 0; m < 2; ++m) { run<0>(m == 1); run<1>(m == 1); run<2>(m == 1); run<3>(m == 1); }
    return 0;
}
