// dev probe (not product code): packed-FP32 VALU ops (v_pk_mul_f32 / v_pk_add_f32, what hipcc emits for adjacent float math on
// gfx950) checked against the plain v_mul_f32 / v_add_f32 results in every lane, alone on the chip and beside a kernel that
// keeps the matrix pipes busy.  Background: profiles/dense_align_repeatability_r02.txt
//   hipcc --offload-arch=gfx950 -O2 tools/probes/packed_f32_probe.hip -o /tmp/pfp && /tmp/pfp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: pk_mul -> pk_add (dependent, back to back);  1: pk_mul -> plain add of both halves;  2: plain mul -> pk_add;
//      3: pk_mul with op_sel / neg modifiers as the compiler uses them -> pk_add
template <int MODE>
__global__ void probe_kernel(int iters, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    float s0 = 1.0f + 0.001f * threadIdx.x, s1 = 0.5f + 0.002f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        floatx2 a = {s0 + 0.25f * (i & 7), s1 - 0.125f * (i & 3)}, b = {1.5f + 0.0625f * (i & 15), s0 * 0.5f}, c = {s1, 3.0f};
        floatx2 r;
        float w0, w1;
        if (MODE == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(c));
            w0 = a.x * b.x + c.x; w1 = a.y * b.y + c.y;
        } else if (MODE == 1) {
            floatx2 t;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
            r.x = t.x + c.x; r.y = t.y + c.y;                       // compiler-scheduled plain consumers
            w0 = a.x * b.x + c.x; w1 = a.y * b.y + c.y;
        } else if (MODE == 2) {
            float m0, m1;
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(m0), "=&v"(m1) : "v"(a.x), "v"(b.x), "v"(a.y), "v"(b.y));
            floatx2 m = {m0, m1};
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(m), "v"(c));
            w0 = a.x * b.x + c.x; w1 = a.y * b.y + c.y;
        } else {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b), "v"(c));
            w0 = a.x * b.x - c.x; w1 = a.x * b.y - c.y;
        }
        // the reference arithmetic must not be contracted or packed behind our back
        asm volatile("" : "+v"(w0), "+v"(w1));
        if (__float_as_uint(r.x) != __float_as_uint(w0) || __float_as_uint(r.y) != __float_as_uint(w1)) ++bad;
        s0 += 0.0078125f; if (s0 > 64.f) s0 = 1.0f;
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
    if (with_mfma) for (int k = 0; k < 60; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe_kernel<MODE>), dim3(2048), dim3(256), 0, s1, 20000, bad, counts);
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long c[2];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0}, sum = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    static const char *names[4] = {"v_pk_mul_f32 -> v_pk_add_f32", "v_pk_mul_f32 -> v_add_f32 x2", "v_mul_f32 x2 -> v_pk_add_f32", "v_pk_mul_f32 op_sel -> v_pk_add_f32 neg"};
    printf("%-42s %-22s: %llu wrong lane results in %.3g wave executions; lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           names[MODE], with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)c[0], q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(counts)); CK(hipFree(sink));
}

int main()
{
    for (int m = 0; m < 2; ++m) { run<0>(m == 1); run<1>(m == 1); run<2>(m == 1); run<3>(m == 1); }
    return 0;
}
