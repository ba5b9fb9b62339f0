// Shared device helpers for libsipmask_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/sipmask_hip.h"

#define SM_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return SM_ERR_LAUNCH;  \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native 16-byte register (see conv_igemm.hip)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// round-to-nearest-even f32 -> bf16 bits (NaN preserved as quiet NaN)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

// two f32 -> packed bf16x2 (round to nearest even): one v_cvt_pk_bf16_f32 on gfx950 (no clang
// builtin; NaN stays NaN) instead of ~8 integer VALU ops per pair
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = bf16_bits_to_f32(v.x & 0xffffu);
  f[1] = bf16_bits_to_f32(v.x >> 16);
  f[2] = bf16_bits_to_f32(v.y & 0xffffu);
  f[3] = bf16_bits_to_f32(v.y >> 16);
  f[4] = bf16_bits_to_f32(v.z & 0xffffu);
  f[5] = bf16_bits_to_f32(v.z >> 16);
  f[6] = bf16_bits_to_f32(v.w & 0xffffu);
  f[7] = bf16_bits_to_f32(v.w >> 16);
}

__device__ __forceinline__ void unpack_bf16x8(const u32x4& v, float* f) {
  f[0] = bf16_bits_to_f32(v.x & 0xffffu);
  f[1] = bf16_bits_to_f32(v.x >> 16);
  f[2] = bf16_bits_to_f32(v.y & 0xffffu);
  f[3] = bf16_bits_to_f32(v.y >> 16);
  f[4] = bf16_bits_to_f32(v.z & 0xffffu);
  f[5] = bf16_bits_to_f32(v.z >> 16);
  f[6] = bf16_bits_to_f32(v.w & 0xffffu);
  f[7] = bf16_bits_to_f32(v.w >> 16);
}

__device__ __forceinline__ u32x4 pack_bf16x8_v(const float* f) {
  u32x4 v;
  v.x = pack_bf16x2(f[0], f[1]);
  v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]);
  v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]);
  v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]);
  v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

// order-preserving map float -> uint32 (larger float => larger key)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// Ranking sigmoid of the detection path: evaluated in DOUBLE and rounded once to f32.  Every integer decision of
// get_bboxes_single (top-k order, score > score_thr, NMS order; sipmask_head.py:563-605) hangs off these values, and
// f32 expf differs by an ulp between libraries (torch-CPU / CUDA / ocml), which is enough to swap two near-equal keys.
// A double evaluation agrees between any two <1-ulp double exp() implementations after the rounding to f32 except
// when the double result lies within ~2^-52 of an f32 rounding boundary (p ~ 2^-28 per value), so the oracle
// (oracle/ops.py: sigmoid_ref) and this kernel produce the same f32 bits and the comparisons downstream are exact.
__device__ __forceinline__ float sigmoid_rank(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

// GroupNorm statistics that several blocks contribute to (the conv epilogues, gn_stats_kernel) are accumulated as
// 64-bit FIXED POINT in units of 2^-SM_GN_FIX_SHIFT: integer addition is associative, so the sums -- and everything
// normalised with them, down to the NMS keep indices -- do not depend on the order in which tiles / waves arrive
// (the reference's GroupNorm is deterministic: M/mmdet/ops/norm.py:12-55 -> ATen).  A contribution is a partial sum
// that was itself formed in a fixed order (one lane's 8 couts, a fixed shuffle tree) and is rounded ONCE to the grid:
// 6e-8 absolute per partial of >= 256 elements, i.e. below 1e-9 of a sum of squares of O(1) activations; the range is
// |sum| < 2^39 = 5.5e11 (rms of a group below ~2000 at 134 400 elements per (image, level, group)).
#define SM_GN_FIX_SHIFT 24
__device__ __forceinline__ unsigned long long gn_fix(float v) {
  // clamp far inside the int64 range so a stray inf/NaN cannot hit the undefined float->int conversion
  const float s = fminf(fmaxf(v * 16777216.f, -4.0e18f), 4.0e18f);
  return (unsigned long long)(long long)rintf(s == s ? s : 0.f);
}
// Sum NV per-lane floats over the 32 lanes of a half-wave and leave total k on the lanes with (lane & 31) >> (5 - log2 NV)
// == k ("transposing" butterfly: at each of the first log2 NV levels a lane keeps one half of its values and hands the
// other half to its partner, so NV values cost NV - 1 + (5 - log2 NV) shuffles instead of 5 NV).  Every total is formed by
// the SAME tree as the plain xor butterfly (pairs at distance 16, then 8, 4, 2, 1; a + b == b + a bitwise), so the bits
// do not depend on NV or on which lane ends up holding the value.
template <int NV>
__device__ __forceinline__ float gn_half_wave_totals(float (&v)[NV], int l31) {
  static_assert(NV == 2 || NV == 4 || NV == 8 || NV == 16 || NV == 32, "NV");
  int d = 16;
#pragma unroll
  for (int n = NV; n > 1; n >>= 1, d >>= 1) {
    const bool up = (l31 & d) != 0;
#pragma unroll
    for (int j = 0; j < n / 2; ++j) {
      float lo = v[j], hi = v[j + n / 2];
      asm("" : "+v"(lo), "+v"(hi));                 // keep the two values in registers: a select of ADDRESSES would put v[] in scratch
      v[j] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, d, 64);
    }
  }
#pragma unroll
  for (; d > 0; d >>= 1) v[0] += __shfl_xor(v[0], d, 64);
  return v[0];
}
__device__ __forceinline__ double gn_unfix(unsigned long long q) { return (double)(long long)q * (1.0 / 16777216.0); }
// mean / reciprocal standard deviation of one group from its fixed-point (sum, sum of squares); double arithmetic (a
// handful of operations per thread) so that E[x^2] - E[x]^2 does not cancel in f32
__device__ __forceinline__ void gn_mean_rstd(const unsigned long long* st, double cnt, float eps, float* mean, float* rstd) {
  const double m = gn_unfix(st[0]) / cnt;
  const double var = fmax(gn_unfix(st[1]) / cnt - m * m, 0.0);
  *mean = (float)m;
  *rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// Zero-fill as a KERNEL.  hipMemsetAsync inside a captured hipGraph becomes a memset node, and on ROCm 7.2 the GroupNorm
// statistics cleared that way came back from graph replays with a 16-byte garbage pattern (two pointer-like 64-bit values
// alternating over the whole buffer, i.e. the fill kernel behind the node ran with a stale value argument) -- one replay in
// three at small shapes, never in eager mode (round 3, tools/debug: replay of [conv + GN apply] on changing inputs).
// Everything a launch plan clears per step goes through this instead.  `bytes` and `p` multiples of 4.
static __global__ void sm_zero_u32_kernel(unsigned int* __restrict__ p, long long n, int vec) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {                                       // 16-byte aligned base: uint4 stores + a scalar tail
    const long long n4 = n >> 2;
    if (i < n4) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i < (n & 3)) p[(n4 << 2) + i] = 0u;
  } else if (i < n) {
    p[i] = 0u;
  }
}
static inline hipError_t sm_zero_async(void* p, size_t bytes, hipStream_t s) {
  if ((bytes & 3) != 0 || ((uintptr_t)p & 3) != 0) return hipErrorInvalidValue;
  const long long n = (long long)(bytes >> 2);
  if (n == 0) return hipSuccess;
  const int vec = ((uintptr_t)p & 15) == 0 ? 1 : 0;
  const long long nthr = vec ? ((n >> 2) > 4 ? (n >> 2) : 4) : n;
  hipLaunchKernelGGL(sm_zero_u32_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (unsigned int*)p, n, vec);
  return hipGetLastError();
}

// __syncthreads() for code that reaches LDS through GENERIC pointers (a working set that lives in LDS or in global scratch
// depending on its size: detect.hip's NMS).  hipcc (ROCm 7.2) compiles such stores to flat_store and -- observed in
// nms_class_kernel's bitonic sort -- leaves the s_barrier behind a loop BARE when the only pending LDS stores are flat ones:
// the last compare-exchange of a step could still be in flight when the other waves passed the barrier and read the old
// value.  Alone on a CU the window is a few cycles and never hit; beside another kernel's LDS traffic (the steps in flight of
// a PipelinedPlan) a sort came out with a padding key inside the first n entries about once in 1 000 launches, and the index
// 0xffffffff it decodes to sent a load 64 GB past the boxes: "Memory access fault by GPU" (round 6).
__device__ __forceinline__ void sm_syncthreads_flat() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE, size): the opt-in for launches with more than
// 64 KB of dynamic LDS is a property of the function on one device, so a process that drives a second GPU (or launches from
// a second host thread) must not inherit "done" from the first (VERDICT r4 #10 / ADVICE r4: the per-process `static bool`
// guards).  One table per shared library (inline function: one instance), guarded by a mutex; the grant only grows.  The
// call itself is a driver round trip (two orders of magnitude slower under rocprofv3's API interception), hence the cache.
inline hipError_t sm_lds_optin(const void* kern, int bytes) {
  struct Entry {
    const void* kern;
    int dev, bytes;
  };
  static std::mutex mu;
  static Entry tab[512];
  static int n = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  int at = -1;
  for (int i = 0; i < n; ++i)
    if (tab[i].kern == kern && tab[i].dev == dev) {
      if (tab[i].bytes >= bytes) return hipSuccess;
      at = i;
      break;
    }
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  if (at < 0 && n < 512) at = n++;
  if (at >= 0) tab[at] = Entry{kern, dev, bytes};   // (a full table only costs the cached answer: the call is repeated)
  return hipSuccess;
}

static inline hipStream_t sm_hip_stream(sm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int sm_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
