// What do the access patterns of the SPLIT16 helpers and of the conv epilogue cost against a plain copy?  (stand-alone probe, not
// product code)     hipcc --offload-arch=gfx950 -O3 tools/probes/stream_pattern_probe.hip -o /tmp/spp && /tmp/spp [MB = 512]
// Every variant copies the same buffer (read + write counted) and prints TB/s:
//   0  float4 copy, lane-contiguous 16 B                         (the guide's 6.3 TB/s reference)
//   1  one 32-byte group per lane as TWO 16-byte accesses        (act_load8 / act_store8: hi chunk, lo chunk; lanes 32 B apart)
//   2  lane-contiguous 16 B, two per lane 1 KiB apart            (the lane-pair form the groups could be moved in)
//   3  conv-epilogue walk, 64-column tiles: a wave covers 8 rows x 256 B of a 4 KiB-pitch tensor, tiles N-fast (layer3.conv3: BN = 64)
//   4  the same with 128-column tiles (512 B per row)            5  with 256-column tiles (1 KiB per row)
//   6  variant 3 with non-temporal stores                        7  variant 1 with non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void copy16(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}

template <bool NT>
__global__ void copy_group(const f4 *__restrict__ x, f4 *__restrict__ y, size_t groups)
{
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        const f4 a = x[2 * g], b = x[2 * g + 1];
        if (NT) {
            __builtin_nontemporal_store(a, &y[2 * g]);
            __builtin_nontemporal_store(b, &y[2 * g + 1]);
        } else {
            y[2 * g] = a;
            y[2 * g + 1] = b;
        }
    }
}

__global__ void copy16x2(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n)
{
    // a wave moves 2 KiB per iteration as two lane-contiguous 1 KiB accesses
    const size_t lane = threadIdx.x & 63, wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t base = wave * 128; base + 127 < n; base += waves * 128) {
        const f4 a = x[base + lane], b = x[base + 64 + lane];
        y[base + lane] = a;
        y[base + 64 + lane] = b;
    }
}

// rows x 1024 channels x 4 B (pitch 4 KiB); workgroup tile = 128 rows x BN columns, 256 threads: thread -> (group g = t % (BN/8), row r = t / (BN/8)),
// NG = 128 * (BN/8) / 256 iterations: the conv engine's epilogue mapping; tiles walk N inside an M tile
template <int BN, bool NT>
__global__ __launch_bounds__(256) void copy_tiles(const char *__restrict__ x, char *__restrict__ y, int rows)
{
    constexpr int GROUPS = BN / 8, RSTEP = 256 / GROUPS, NG = 128 / RSTEP, NT_TILES = 1024 / BN;
    const int mtiles = (rows + 127) / 128;
    for (int tile = blockIdx.x; tile < mtiles * NT_TILES; tile += gridDim.x) {
        const int mt = tile / NT_TILES, nt = tile - mt * NT_TILES;
        const int g = threadIdx.x % GROUPS, r0 = threadIdx.x / GROUPS;
        f4 a[NG], b[NG];
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int row = mt * 128 + r0 + it * RSTEP;
            const size_t off = (size_t)row * 4096 + (size_t)(nt * BN + g * 8) * 4;
            a[it] = f4{0, 0, 0, 0};
            b[it] = a[it];
            if (row < rows) {
                a[it] = *reinterpret_cast<const f4 *>(x + off);
                b[it] = *reinterpret_cast<const f4 *>(x + off + 16);
            }
        }
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int row = mt * 128 + r0 + it * RSTEP;
            const size_t off = (size_t)row * 4096 + (size_t)(nt * BN + g * 8) * 4;
            if (row < rows) {
                if (NT) {
                    __builtin_nontemporal_store(a[it], reinterpret_cast<f4 *>(y + off));
                    __builtin_nontemporal_store(b[it], reinterpret_cast<f4 *>(y + off + 16));
                } else {
                    *reinterpret_cast<f4 *>(y + off) = a[it];
                    *reinterpret_cast<f4 *>(y + off + 16) = b[it];
                }
            }
        }
    }
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 512;
    const size_t bytes = mb << 20;
    char *x = nullptr, *y = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(x, 1, bytes);
    (void)hipMemset(y, 0, bytes);
    const size_t n16 = bytes / 16, groups = bytes / 32;
    const int rows = (int)(bytes / 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char *names[8] = {"float4 copy, lane-contiguous", "32-B group per lane as 2 x 16 B (act_load8/store8)", "lane-contiguous 16 B, two per lane 1 KiB apart",
                            "epilogue walk, 64-column tiles (256 B per row)", "epilogue walk, 128-column tiles (512 B per row)",
                            "epilogue walk, 256-column tiles (1 KiB per row)", "epilogue walk, 64-column tiles, non-temporal stores",
                            "32-B group per lane, non-temporal stores"};
    printf("stream pattern probe: %zu MB in, %zu MB out per pass (TB/s counts both); 5 timed passes after 2 warm-ups\n", mb, mb);
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 8; ++v) {
            auto launch = [&]() {
                const int G = 256 * 16;
                switch (v) {
                case 0: hipLaunchKernelGGL(copy16, dim3(G), dim3(256), 0, 0, (const f4 *)x, (f4 *)y, n16); break;
                case 1: hipLaunchKernelGGL(copy_group<false>, dim3(G), dim3(256), 0, 0, (const f4 *)x, (f4 *)y, groups); break;
                case 2: hipLaunchKernelGGL(copy16x2, dim3(G), dim3(256), 0, 0, (const f4 *)x, (f4 *)y, n16); break;
                case 3: hipLaunchKernelGGL((copy_tiles<64, false>), dim3(G), dim3(256), 0, 0, (const char *)x, y, rows); break;
                case 4: hipLaunchKernelGGL((copy_tiles<128, false>), dim3(G), dim3(256), 0, 0, (const char *)x, y, rows); break;
                case 5: hipLaunchKernelGGL((copy_tiles<256, false>), dim3(G), dim3(256), 0, 0, (const char *)x, y, rows); break;
                case 6: hipLaunchKernelGGL((copy_tiles<64, true>), dim3(G), dim3(256), 0, 0, (const char *)x, y, rows); break;
                default: hipLaunchKernelGGL(copy_group<true>, dim3(G), dim3(256), 0, 0, (const f4 *)x, (f4 *)y, groups); break;
                }
            };
            launch();
            launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %d  %-58s %7.1f us per pass  %5.2f TB/s\n", v, names[v], ms / 5 * 1e3, 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
