/* Force-included in front of the REFERENCE'S OWN CUDA translation units (lib/model/nms/src/nms_cuda_kernel.cu,
 * lib/model/roi_align/src/roi_align_kernel.cu, compiled where they lie under /root/reference by oracle/build.py:build_ref)
 * so that hipcc can build them for gfx950: the kernels and their launchers are untouched, only the dozen CUDA runtime
 * names they use are spelled the HIP way.  TEST INFRASTRUCTURE ONLY (oracle/_ref/): this is how the restatements in
 * oracle/csrc/oracle_ops.c and the product kernels are checked against the reference's real kernels on the MI355X.
 * cudaMemcpyHostToDevice / DeviceToHost -> hipMemcpyDefault: nms_cuda.c:14-16 hands DEVICE pointers to the
 * "boxes_host" parameter (unified addressing sorts it out under CUDA; hipMemcpyDefault does the same). */
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyDefault
#define cudaMemcpyDeviceToHost hipMemcpyDefault
#define cudaStream_t hipStream_t
