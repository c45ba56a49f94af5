// dev tool: runs geom_kernel<6> of tools/probes/geometry_lane_probe.hip from a code object built out of EDITED assembly, beside the
// MFMA load kernel, and prints the lane-quarter histogram of deviating results.   usage: co_runner <file.co> [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const char *kname = argc > 3 ? argv[3] : "_Z11geom_kernelILi6EEvPKfS1_iiPyS2_Pi";
    hipModule_t mod; CK(hipModuleLoad(&mod, argv[1]));
    hipFunction_t geom, mfma;
    CK(hipModuleGetFunction(&geom, mod, kname));
    CK(hipModuleGetFunction(&mfma, mod, "_Z16mfma_load_kerneliPf"));
    std::vector<float> poses = {-28.314176559448242f, 7.484988689422607f, 43.75969314575195f, 1.4719889163970947f, 1.581013798713684f, 4.676165580749512f, 2.70131254196167f,
                                -31.009122848510742f, 7.616531848907471f, 45.70008850097656f, 1.4833614826202393f, 1.5263630151748657f, 4.672759056091309f, 2.677232027053833f};
    srand(7);
    for (int k = 0; k < 62; ++k) {
        float z = 8.f + 40.f * (rand() / (float)RAND_MAX), x = (rand() / (float)RAND_MAX - 0.5f) * z, th = 6.28f * (rand() / (float)RAND_MAX) - 3.14f;
        float p[7] = {x, 1.5f + 6.f * (rand() / (float)RAND_MAX), z, 1.6f, 1.5f, 4.2f, th};
        poses.insert(poses.end(), p, p + 7);
    }
    int nposes = (int)poses.size() / 7;
    std::vector<float> cs;
    for (int k = 0; k < nposes; ++k) { cs.push_back((float)cos((double)poses[k * 7 + 6])); cs.push_back((float)sin((double)poses[k * 7 + 6])); }
    float *dposes, *dcs, *sink; unsigned long long *bad, *total; int *first_bad;
    CK(hipMalloc(&dposes, poses.size() * 4)); CK(hipMemcpy(dposes, poses.data(), poses.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dcs, cs.size() * 4)); CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bad, 64 * 8)); CK(hipMalloc(&total, 64)); CK(hipMalloc(&first_bad, 64));
    CK(hipMemset(bad, 0, 64 * 8)); CK(hipMemset(total, 0, 64)); CK(hipMemset(first_bad, 0xff, 64));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    int launches = argc > 2 ? atoi(argv[2]) : 40, iters = 4000, miters = 20000;
    void *margs[] = {&miters, &sink};
    for (int k = 0; k < launches * 5; ++k) CK(hipModuleLaunchKernel(mfma, 512, 1, 1, 512, 1, 1, 0, s2, margs, nullptr));
    void *gargs[] = {&dposes, &dcs, &nposes, &iters, &bad, &total, &first_bad};
    for (int k = 0; k < launches; ++k) CK(hipModuleLaunchKernel(geom, 1200, 1, 1, 256, 1, 1, 0, s1, gargs, nullptr));
    CK(hipStreamSynchronize(s1)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(64); unsigned long long tot[8];
    CK(hipMemcpy(h.data(), bad, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tot, total, 64, hipMemcpyDeviceToHost));
    unsigned long long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) q[l / 16] += h[l];
    printf("%llu %llu %llu %llu deviating (lane quarters) in %.3g wave executions; fields: dist %llu nearest %llu planes %llu\n", q[0], q[1], q[2], q[3], (double)tot[0], tot[2], tot[3], tot[5]);
    return 0;
}
