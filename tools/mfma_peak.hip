// MFMA issue-rate probe (gfx950): W waves per block of independent v_mfma_f32_32x32x16_bf16 chains, no memory traffic.
// usage: mfma_peak [waves_per_block] [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(blockIdx.x + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
int main(int argc, char** argv) {
  int waves = argc > 1 ? atoi(argv[1]) : 8, bpc = argc > 2 ? atoi(argv[2]) : 1;
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, nacc = 8;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<8>, dim3(256 * bpc), dim3(64 * waves), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 2.0 * 32 * 32 * 16 * (double)iters * nacc * waves * 256 * bpc;
    double per_simd = (double)iters * nacc * waves * bpc / 4.0;   // MFMAs per SIMD
    printf("waves/block %d blocks/CU %d: %.3f ms  %.0f TFLOP/s  (%.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz)\n", waves, bpc, ms,
           flop / ms / 1e9, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
  return 0;
}
