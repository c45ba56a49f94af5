// dev probe (not product code): MINIMAL form of the hazard behind profiles/dense_align_repeatability_r02.txt.
// hipcc lowers a dynamically indexed read of a register-resident array on gfx950 to
//        s_set_gpr_idx_on sN, gpr_idx(SRC0) ; v_mov_b32 vD, vBASE ; s_set_gpr_idx_off
// This probe issues exactly that sequence (inline asm, known array contents) and checks every lane, alone on the chip and
// beside a kernel that keeps the matrix pipes busy, with 0..3 wait states inserted before s_set_gpr_idx_off.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/gpr_index_hazard_probe.hip -o /tmp/gip && /tmp/gip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define SEQ(NOPS)                                                                                                     \
    asm volatile("v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %4\n\tv_mov_b32 v203, %5\n\ts_nop 4\n\t"  \
                 "s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v200\n\t" NOPS "s_set_gpr_idx_off"                \
                 : "=v"(out) : "s"(idx), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "v200", "v201", "v202", "v203", "m0")

// the compiler's real pattern: several indexed reads back to back, each with its own index (off ; on ; v_mov ; off ; on ...)
#define SEQ4(NOPS)                                                                                                    \
    asm volatile("v_mov_b32 v200, %8\n\tv_mov_b32 v201, %9\n\tv_mov_b32 v202, %10\n\tv_mov_b32 v203, %11\n\ts_nop 4\n\t" \
                 "s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\tv_mov_b32 %0, v200\n\t" NOPS "s_set_gpr_idx_off\n\t"             \
                 "s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\tv_mov_b32 %1, v200\n\t" NOPS "s_set_gpr_idx_off\n\t"             \
                 "s_set_gpr_idx_on %6, gpr_idx(SRC0)\n\tv_mov_b32 %2, v200\n\t" NOPS "s_set_gpr_idx_off\n\t"             \
                 "s_set_gpr_idx_on %7, gpr_idx(SRC0)\n\tv_mov_b32 %3, v200\n\t" NOPS "s_set_gpr_idx_off"                  \
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "s"(i0), "s"(i1), "s"(i2), "s"(i3), "v"(a0), "v"(a1), "v"(a2), "v"(a3) \
                 : "v200", "v201", "v202", "v203", "m0")

template <int WAIT>
__global__ void chain_kernel(int iters, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0, as_other = 0;
    for (int i = 0; i < iters; ++i) {
        const float a0 = 10.f + i, a1 = 20.f + i, a2 = 30.f + i, a3 = 40.f + i;
        const int r = (i + blockIdx.x) & 3;
        const int i0 = __builtin_amdgcn_readfirstlane((r + 1) & 3), i1 = __builtin_amdgcn_readfirstlane((r + 3) & 3);
        const int i2 = __builtin_amdgcn_readfirstlane((r + 2) & 3), i3 = __builtin_amdgcn_readfirstlane(r);
        float o0, o1, o2, o3;
        if (WAIT == 0) SEQ4("");
        else if (WAIT == 1) SEQ4("s_nop 0\n\t");
        else if (WAIT == 2) SEQ4("s_nop 1\n\t");
        else SEQ4("s_nop 3\n\t");
        const float arr[4] = {a0, a1, a2, a3};
        const float w0 = arr[(r + 1) & 3], w1 = arr[(r + 3) & 3], w2 = arr[(r + 2) & 3], w3 = arr[r];
        if (o0 != w0 || o1 != w1 || o2 != w2 || o3 != w3) {
            ++bad;
            if ((o0 != w0 && (o0 == w1 || o0 == a0)) || (o1 != w1 && (o1 == w2 || o1 == w0 || o1 == a0)) || (o2 != w2 && (o2 == w3 || o2 == w1 || o2 == a0)) || (o3 != w3 && (o3 == w2 || o3 == a0))) ++as_other;
        }
    }
    if (bad) { atomicAdd(bad_lanes + lane, bad); atomicAdd(counts + 1, as_other); }
    if (threadIdx.x == 0) atomicAdd(counts, (unsigned long long)iters * (blockDim.x / 64));
}

template <int WAIT>
__global__ void probe_kernel(int iters, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0, as_unindexed = 0;
    for (int i = 0; i < iters; ++i) {
        const float a0 = 10.f + i, a1 = 20.f + i, a2 = 30.f + i, a3 = 40.f + i;
        const int idx = __builtin_amdgcn_readfirstlane(1 + (i + blockIdx.x) % 3);      // never 0: an un-indexed read is visible
        float out;
        if (WAIT == 0) SEQ("");
        else if (WAIT == 1) SEQ("s_nop 0\n\t");
        else if (WAIT == 2) SEQ("s_nop 1\n\t");
        else SEQ("s_nop 3\n\t");
        const float want = idx == 1 ? a1 : (idx == 2 ? a2 : a3);
        if (out != want) { ++bad; if (out == a0) ++as_unindexed; }
    }
    if (bad) { atomicAdd(bad_lanes + lane, bad); atomicAdd(counts + 1, as_unindexed); }
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

template <int WAIT, bool CHAIN>
static void run(bool with_mfma)
{
    unsigned long long *bad, *counts; float *sink;
    CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&counts, 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(counts, 0, 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (with_mfma) for (int k = 0; k < 60; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) {
        if (CHAIN) hipLaunchKernelGGL((chain_kernel<WAIT>), dim3(2048), dim3(256), 0, s1, 20000, bad, counts);
        else hipLaunchKernelGGL((probe_kernel<WAIT>), dim3(2048), dim3(256), 0, s1, 20000, bad, counts);
    }
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long c[2];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0}, sum = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    printf("%s, %d wait state(s) before s_set_gpr_idx_off, %-22s: %llu wrong lane results in %.3g wave executions (%llu of them = the un-indexed / a neighbouring read's element); lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           CHAIN ? "4 indexed reads back to back" : "one indexed read", WAIT == 3 ? 4 : WAIT, with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)c[0], c[1], q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(counts)); CK(hipFree(sink));
}

int main()
{
    for (int m = 0; m < 2; ++m) { run<0, false>(m == 1); run<0, true>(m == 1); run<1, true>(m == 1); run<2, true>(m == 1); run<3, true>(m == 1); }
    return 0;
}
