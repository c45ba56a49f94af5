// dev probe (not product code): does the result of a transcendental-unit instruction (v_sqrt_f32 / v_rcp_f32) reach a dependent
// VALU instruction issued one wait state later -- the distance the compiler leaves -- in ALL 64 lanes, also while waves of
// another kernel keep the SIMD's matrix pipe busy?   Background: profiles/dense_align_repeatability_r02.txt.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/trans_hazard_probe.hip -o /tmp/thp && /tmp/thp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// GAP: the instruction(s) between the trans op and its consumer
template <int GAP>
__device__ __forceinline__ float sqrt_then_use(float x, float sentinel)
{
    float r = sentinel, y;
    if (GAP == 0)
        asm volatile("s_nop 4\n\tv_sqrt_f32 %0, %2\n\ts_nop 0\n\tv_add_f32 %1, %0, %0" : "+v"(r), "=v"(y) : "v"(x));
    else if (GAP == 1)
        asm volatile("s_nop 4\n\tv_sqrt_f32 %0, %2\n\tv_mov_b32 %1, %2\n\tv_add_f32 %1, %0, %0" : "+v"(r), "=&v"(y) : "v"(x));
    else if (GAP == 2)
        asm volatile("s_nop 4\n\tv_sqrt_f32 %0, %2\n\ts_nop 1\n\tv_add_f32 %1, %0, %0" : "+v"(r), "=v"(y) : "v"(x));
    else
        asm volatile("s_nop 4\n\tv_sqrt_f32 %0, %2\n\ts_nop 7\n\tv_add_f32 %1, %0, %0" : "+v"(r), "=v"(y) : "v"(x));
    return y;
}

template <int GAP>
__global__ void probe_kernel(int iters, unsigned long long *bad_lanes /* [64] */, unsigned long long *total)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    float x = 2.0f + (float)(blockIdx.x & 1023);
    for (int i = 0; i < iters; ++i) {
        const float sentinel = 1.0e6f + (float)i;             // what a stale read of the destination register would see
        const float y = sqrt_then_use<GAP>(x, sentinel);
        float r;                                                // the same value with a long gap
        asm volatile("v_sqrt_f32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "=v"(r) : "v"(x));
        if (y != r + r) ++bad;
        x += 1.0f;
        if (x > 4.0e6f) x = 2.0f;
    }
    if (bad) atomicAdd(bad_lanes + lane, bad);
    if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * blockDim.x);
}

// keeps every SIMD's matrix pipe busy: 8 waves per workgroup, one workgroup per CU and more
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

template <int GAP>
static void run(const char *what, bool with_mfma, int blocks, int iters)
{
    unsigned long long *bad, *total;
    float *sink;
    CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&total, 8)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(total, 0, 8));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (with_mfma)
        for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe_kernel<GAP>), dim3(blocks), dim3(256), 0, s1, iters, bad, total);
    CK(hipStreamSynchronize(s1));
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64);
    unsigned long long tot = 0, sum = 0;
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    printf("%-44s %-22s: %llu wrong of %.3g uses; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n", what,
           with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)tot, q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(total)); CK(hipFree(sink));
}

int main()
{
    const int blocks = 2048, iters = 20000;
    for (int m = 0; m < 2; ++m) {
        run<0>("v_sqrt_f32 ; s_nop 0 ; use   (compiler's gap)", m == 1, blocks, iters);
        run<1>("v_sqrt_f32 ; 1 VALU ; use    (compiler's gap)", m == 1, blocks, iters);
        run<2>("v_sqrt_f32 ; s_nop 1 ; use", m == 1, blocks, iters);
        run<3>("v_sqrt_f32 ; s_nop 7 ; use", m == 1, blocks, iters);
    }
    return 0;
}
