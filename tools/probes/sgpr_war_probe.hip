// dev probe (not product code): write-after-read on an SGPR that a VALU instruction of the same wave has just read -- as lane
// mask (v_cndmask) or as scalar operand -- by the next SALU / SMEM instruction.  Does the VALU see the OLD value in all 64
// lanes, also beside a kernel that keeps the matrix pipes busy?   Background: profiles/dense_align_repeatability_r02.txt
//   hipcc --offload-arch=gfx950 -O2 tools/probes/sgpr_war_probe.hip -o /tmp/swp && /tmp/swp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: v_cndmask(mask s[40:41]) ; s_mov_b64 s[40:41], ~mask          (SALU overwrites the mask)
//      1: v_cndmask(mask s[40:41]) ; s_load_dwordx2 s[40:41] <- ~mask     (SMEM overwrites the mask)
//      2: v_add_f32 v, s40, v      ; s_mov_b32 s40, other                (SALU overwrites a scalar operand)
//      3: v_add_f32 v, s40, v      ; s_load_dword s40 <- other           (SMEM overwrites a scalar operand)
template <int MODE>
__global__ void probe_kernel(int iters, const unsigned long long *mem, unsigned long long *bad_lanes, unsigned long long *counts)
{
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const float a = 100.f + i, b = 200.f + i;
        float out;
        if (MODE == 0) {
            asm volatile("s_mov_b64 s[40:41], 0\n\ts_nop 4\n\tv_cndmask_b32_e64 %0, %1, %2, s[40:41]\n\ts_mov_b64 s[40:41], -1\n\ts_nop 4"
                         : "=v"(out) : "v"(a), "v"(b) : "s40", "s41");
            if (out != a) ++bad;
        } else if (MODE == 1) {
            asm volatile("s_mov_b64 s[40:41], 0\n\ts_nop 4\n\tv_cndmask_b32_e64 %0, %1, %2, s[40:41]\n\ts_load_dwordx2 s[40:41], %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(out) : "v"(a), "v"(b), "s"(mem) : "s40", "s41", "memory");
            if (out != a) ++bad;
        } else if (MODE == 2) {
            asm volatile("s_mov_b32 s40, 1.0\n\ts_nop 4\n\tv_add_f32 %0, s40, %1\n\ts_mov_b32 s40, 2.0\n\ts_nop 4" : "=v"(out) : "v"(a) : "s40");
            if (out != a + 1.0f) ++bad;
        } else {
            asm volatile("s_mov_b32 s40, 1.0\n\ts_nop 4\n\tv_add_f32 %0, s40, %1\n\ts_load_dword s40, %2, 0x8\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(out) : "v"(a), "s"(mem) : "s40", "memory");
            if (out != a + 1.0f) ++bad;
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

template <int MODE>
static void run(bool with_mfma)
{
    unsigned long long *bad, *counts, *mem; float *sink;
    CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&counts, 16)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&mem, 64));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(counts, 0, 16));
    unsigned long long hm[8] = {~0ULL, 0x4000000040000000ULL /* 2.0f, 2.0f */, 0, 0, 0, 0, 0, 0};
    CK(hipMemcpy(mem, hm, 64, hipMemcpyHostToDevice));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (with_mfma) for (int k = 0; k < 80; ++k) hipLaunchKernelGGL(mfma_load_kernel, dim3(512), dim3(512), 0, s2, 20000, sink);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe_kernel<MODE>), dim3(2048), dim3(256), 0, s1, 20000, mem, bad, counts);
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long c[2];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0}, sum = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += h[l]; sum += h[l]; }
    static const char *names[4] = {"v_cndmask(mask sgpr) ; s_mov_b64 same sgpr", "v_cndmask(mask sgpr) ; s_load_dwordx2 same sgpr", "v_add_f32(s40) ; s_mov_b32 s40", "v_add_f32(s40) ; s_load_dword s40"};
    printf("%-48s %-22s: %llu wrong lane results in %.3g wave executions; lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
           names[MODE], with_mfma ? "beside an MFMA kernel" : "alone on the chip", sum, (double)c[0], q[0], q[1], q[2], q[3]);
    CK(hipFree(bad)); CK(hipFree(counts)); CK(hipFree(sink)); CK(hipFree(mem));
}

int main()
{
    for (int m = 0; m < 2; ++m) { run<0>(m == 1); run<1>(m == 1); run<2>(m == 1); run<3>(m == 1); }
    return 0;
}
