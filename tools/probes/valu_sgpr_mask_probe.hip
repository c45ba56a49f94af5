// dev probe (not product code): a VALU compare writes a 64-bit lane mask to an SGPR pair (or VCC), a v_cndmask reads it two
// wait states later (the distance hipcc leaves on gfx950: "s_nop 1").  Does the consumer see the NEW mask in all 64 lanes,
// also while waves of another kernel keep the SIMD's matrix pipe busy?  Background: profiles/dense_align_repeatability_r02.txt
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_sgpr_mask_probe.hip -o /tmp/vsp && /tmp/vsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define SEQ_SGPR(NOPS) asm volatile("v_cmp_gt_f32_e64 s[40:41], %1, 0\n\t" NOPS "v_cndmask_b32_e64 %0, %2, %3, s[40:41]" \
                                    : "=v"(out) : "v"(x), "v"(a), "v"(b) : "s40", "s41")
#define SEQ_VCC(NOPS) asm volatile("v_cmp_gt_f32_e32 vcc, 0, %1\n\t" NOPS "v_cndmask_b32_e32 %0, %2, %3, vcc" \
                                   : "=v"(out) : "v"(x), "v"(a), "v"(b) : "vcc")

// MODE 0: SGPR pair, MODE 1: VCC.  WAIT: wait states between producer and consumer (1, 2 = the compiler's, 4)
template <int MODE, int WAIT>
__global__ void probe_kernel(int iters, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        // the mask flips in every lane from one iteration to the next: a stale bit is always a wrong bit
        const float x = ((lane ^ i) & 1) ? 1.0f : -1.0f;
        const float a = 100.f + i, b = 200.f + i;
        float out;
        if (MODE == 0) {
            if (WAIT == 1) SEQ_SGPR("s_nop 0\n\t"); else if (WAIT == 2) SEQ_SGPR("s_nop 1\n\t"); else SEQ_SGPR("s_nop 3\n\t");
            if (out != (x > 0.f ? b : a)) ++bad;
        } else {
            if (WAIT == 1) SEQ_VCC("s_nop 0\n\t"); else if (WAIT == 2) SEQ_VCC("s_nop 1\n\t"); else SEQ_VCC("s_nop 3\n\t");
            if (out != (0.f > x ? b : a)) ++bad;
        }
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

template <int MODE, int WAIT>
static void run(bool with_mfma)
{
    unsigned long long *bad, *counts; float *sink;
    CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&counts, 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(counts, 0, 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (with_mfma) for (int k = 0; k < 60; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe_kernel<MODE, WAIT>), dim3(2048), dim3(256), 0, s1, 40000, bad, counts);
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long c[2];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0}, sum = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    printf("v_cmp -> %s, %d wait state(s), %-22s: %llu wrong lane results in %.3g wave executions; lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           MODE == 0 ? "s[40:41] -> v_cndmask_e64" : "vcc      -> v_cndmask_e32", WAIT, with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)c[0], q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(counts)); CK(hipFree(sink));
}

int main()
{
    for (int m = 0; m < 2; ++m) {
        run<0, 1>(m == 1); run<0, 2>(m == 1); run<0, 4>(m == 1);
        run<1, 1>(m == 1); run<1, 2>(m == 1); run<1, 4>(m == 1);
    }
    return 0;
}
