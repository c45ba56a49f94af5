// Sustained-rate probe of the conv engine's MFMA pattern (dev tool; VERDICT r3 item 6: "prove or retract the ceiling").
// Every wave keeps its operand fragments in registers and issues the engine's exact slice -- for each accumulator the three
// products al.bh, ah.bl, ah.bh on v_mfma_f32_32x32x16_f16 -- with NO global traffic in the loop; variant 1 adds the K loop's LDS
// fragment fetches (8 ds_read_b128 per 12 MFMAs, as the 64x64 per-wave tile of conv_f16s.hip does).  One 512-thread workgroup
// per CU (two waves per SIMD, as the engine's 8-wave tiles), launched back to back for the requested time; the wrapper
// (tools/mfma_sustained.py) samples rocm-smi clocks / power meanwhile.
//   usage: mfma_sustained <variant 0|1|2|3> <data zero|rand> <seconds> [workgroups per CU = 1]   (2 / 3: operand-stationary MFMA orders)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <chrono>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ORDER (round 4, energy question: does the matrix pipe draw less when consecutive MFMAs share an operand register?):
//   0 = the engine's order (all al.bh, then all ah.bl, then all ah.bh; A and B both change between most neighbours)
//   1 = A-stationary: per row block i: ah.bh0 ah.bl0 ah.bh1 ah.bl1 (A = ah for four in a row), al.bh0 al.bh1 (A = al for two)
//   2 = B-stationary: per column block j: ah.bh ah'.bh al.bh al'.bh (B = bh_j for four in a row), ah.bl ah'.bl (B = bl_j for two)
template <int LDS, int ORDER = 0>
__global__ __launch_bounds__(512, 2) void mfma_sustained_kernel(const half8 *src, float *out, int iters)
{
    __shared__ half8 frag[8 * 512 + 64];
    const int t = threadIdx.x;
    half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ah[i] = src[(0 + i) * 512 + t];
        al[i] = src[(2 + i) * 512 + t];
        bh[i] = src[(4 + i) * 512 + t];
        bl[i] = src[(6 + i) * 512 + t];
    }
    if (LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) frag[i * 512 + t] = src[i * 512 + t];
        __syncthreads();
    }
    floatx16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            // the K loop's fragment fetch: 8 x ds_read_b128 per slice, address moving with the iteration so that it stays in the loop
            const half8 *f = frag + (it & 1) * 16 + t - (it & 1) * 16;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = f[(0 + i) * 512];
                al[i] = f[(2 + i) * 512];
                bh[i] = f[(4 + i) * 512];
                bl[i] = f[(6 + i) * 512];
            }
        }
        if (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        } else if (ORDER == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e] + accx[i][j][e];
    out[(size_t)blockIdx.x * 512 + t] = s;
}

int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const bool rnd = argc > 2 && !strcmp(argv[2], "rand");
    const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
    const int per_cu = argc > 4 ? atoi(argv[4]) : 1;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * per_cu;
    std::vector<_Float16> h(8 * 512 * 8);
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int f = 0; f < 8; ++f)
        for (int i = 0; i < 512 * 8; ++i) {
            float v = rnd ? nd(g) : 0.f;
            // fragments 0,1,4,5 = hi halves (values of order 1), 2,3,6,7 = lo halves (the rounding residue of an f16: 2^-11 of it)
            const bool lo = (f & 2) != 0;
            h[(size_t)f * 4096 + i] = (_Float16)(lo ? v * 0.00048828125f : v);
        }
    half8 *src;
    float *out;
    CHECK(hipMalloc(&src, h.size() * 2));
    CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 400000;             // ~0.1-0.2 s per launch
    auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL((mfma_sustained_kernel<0, 0>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
        else if (variant == 1) hipLaunchKernelGGL((mfma_sustained_kernel<1, 0>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
        else if (variant == 2) hipLaunchKernelGGL((mfma_sustained_kernel<0, 1>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
        else hipLaunchKernelGGL((mfma_sustained_kernel<0, 2>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
    };
    launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double total_ms = 0;
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CHECK(hipEventRecord(e0, 0));
        for (int k = 0; k < 4; ++k) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        launches += 4;
    }
    const double flops = (double)launches * blocks * 8.0 * iters * 12.0 * 32768.0;
    printf("variant %d (%s) data %s: %d workgroups of 8 waves on %d CUs, %ld launches, %.2f s of kernel time: %.1f TFLOP/s issued "
           "(= %.1f TF/s algorithmic at 3 products per flop), %.2f ns per 12-MFMA slice per wave\n",
           variant, variant == 1 ? "MFMA + 8 ds_read_b128 per 12 MFMAs" : variant == 2 ? "register-resident, A-stationary order" : variant == 3 ? "register-resident, B-stationary order" : "register-resident MFMA only", rnd ? "random" : "zeros", blocks, cus,
           launches, total_ms / 1e3, flops / (total_ms * 1e-3) / 1e12, flops / (total_ms * 1e-3) / 1e12 / 3.0,
           total_ms * 1e6 / ((double)launches * iters));
    return 0;
}
