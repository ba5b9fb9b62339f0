// TEST INFRASTRUCTURE ONLY (oracle/): lets the reference's CUDA op sources compile as plain host C++ so that their
// device functions and host algebra RUN (on the CPU) and pin the oracle's restatements by execution
// (VERDICT r1 #6).  Used only by oracle/build_ref.py, force-included in front of the reference's own, unmodified
// source files, which are compiled from where they lie under /root/reference; nothing of them is copied here.
//
// What the shim supplies:
//   * __global__/__device__/... as empty qualifiers; blockIdx/threadIdx = 0, blockDim/gridDim = 1, so that the sources'
//     own grid-stride loops (CUDA_KERNEL_LOOP, CUDA_1D_KERNEL_LOOP) run every index in ONE call of the "kernel";
//     the <<<grid, block>>> launch syntax is cut out by build_ref.py while piping the source into the compiler;
//   * atomicAdd as a plain add (one thread), cudaGetLastError & co. as successes;
//   * the PyTorch-1.1 spellings the sources use and PyTorch 2.x dropped: AT_CHECK, at::IntList, THCCeilDiv,
//     THCudaCheck, the THC headers (empty files beside this one);
//   * `.is_cuda()` answered "yes" for the CPU tensors this build feeds (the sources refuse CPU tensors);
//   * the half-precision dispatch narrowed to float/double (at::Half arithmetic of the device code is CUDA-only).
#pragma once
#include <torch/extension.h>
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>
#include <ATen/DeviceGuard.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct ref_shim_dim3 {
  unsigned x, y, z;
  ref_shim_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef ref_shim_dim3 dim3;
static const ref_shim_dim3 threadIdx(0, 0, 0), blockIdx(0, 0, 0), blockDim(1, 1, 1), gridDim(1, 1, 1);

template <typename T>
static inline T atomicAdd(T* p, T v) {
  T old = *p;
  *p += v;
  return old;
}

typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
static inline cudaError_t cudaGetLastError() { return 0; }
static inline const char* cudaGetErrorString(cudaError_t) { return "no error (host shim)"; }
#define THCudaCheck(x) ((void)(x))
template <typename T>
static inline T THCCeilDiv(T a, T b) { return (a + b - 1) / b; }

// CUDA's overloaded device max/min (mixed float/double arguments promote to double)
template <typename T> static inline T max(T a, T b) { return a < b ? b : a; }
template <typename T> static inline T min(T a, T b) { return b < a ? b : a; }
static inline double max(double a, float b) { return a < (double)b ? (double)b : a; }
static inline double max(float a, double b) { return (double)a < b ? b : (double)a; }
static inline double min(double a, float b) { return (double)b < a ? (double)b : a; }
static inline double min(float a, double b) { return b < (double)a ? b : (double)a; }

#ifdef AT_CHECK
#undef AT_CHECK
#endif
#define AT_CHECK(cond, ...) TORCH_CHECK(cond, "reference check failed (host shim)")
namespace at {
using IntList = IntArrayRef;
}

// AT_DISPATCH_* took `tensor.type()` (DeprecatedTypeProperties) in PyTorch 1.1; 2.x wants a ScalarType
static inline at::ScalarType ref_shim_scalar_type(at::ScalarType s) { return s; }
static inline at::ScalarType ref_shim_scalar_type(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
#undef AT_DISPATCH_FLOATING_TYPES_AND_HALF
#define AT_DISPATCH_FLOATING_TYPES_AND_HALF(TYPE, NAME, ...) \
  AT_DISPATCH_FLOATING_TYPES(ref_shim_scalar_type(TYPE), NAME, __VA_ARGS__)

// the sources insist on CUDA tensors: `x.type().is_cuda()` / `x.is_cuda()` -> true for what this build feeds.
// Defined AFTER every torch header above has been parsed (include guards keep them from being parsed again).
#define is_cuda() is_sparse() == false
