// MFMA issue-rate probe (gfx950): W waves per block of independent v_mfma_f32_32x32x16_bf16 chains, no memory traffic.
// usage: mfma_peak [waves_per_block] [blocks_per_cu] [iters] [zero_operands]
// Round 3: the kernel also reads the SHADER clock counter (s_memtime via clock64()) and the constant 100 MHz real-time
// counter (s_memrealtime via wall_clock64()) around its loop, so the clock the MFMAs actually ran at is part of the
// measurement: cycles per MFMA per SIMD = (shader cycles) / (MFMAs per SIMD), and TFLOP/s scales with the clock the
// part sustains under this load, not with a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, int zero, unsigned long long* clk) {
  f32x16 acc[NACC];
  const unsigned long long c0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(zero ? 0.f : (float)(threadIdx.x + e)); b[e] = (__bf16)(zero ? 0.f : (float)(blockIdx.x + e)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
  const unsigned long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    clk[(blockIdx.x ? 2 : 0) + 0] = c1 - c0;
    clk[(blockIdx.x ? 2 : 0) + 1] = r1 - r0;
  }
}
int main(int argc, char** argv) {
  int waves = argc > 1 ? atoi(argv[1]) : 8, bpc = argc > 2 ? atoi(argv[2]) : 1;
  const int iters = argc > 3 ? atoi(argv[3]) : 4000, zero = argc > 4 ? atoi(argv[4]) : 0;
  float* out; hipMalloc(&out, 4);
  unsigned long long* clk; hipMalloc(&clk, 32); unsigned long long h[4];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nacc = 8;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<8>, dim3(256 * bpc), dim3(64 * waves), 0, 0, out, iters, zero, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 2.0 * 32 * 32 * 16 * (double)iters * nacc * waves * 256 * bpc;
    double per_simd = (double)iters * nacc * waves * bpc / 4.0;   // MFMAs per SIMD
    hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    // one wave issues iters * nacc MFMAs; the SIMD it sits on also serves (waves * bpc / 4 - 1) other waves of this launch
    const double mhz0 = 100.0 * (double)h[0] / (double)h[1], mhz1 = 100.0 * (double)h[2] / (double)h[3];
    printf("waves/block %d blocks/CU %d iters %d %s operands: %.3f ms  %.0f TFLOP/s  (%.2f ns per MFMA per SIMD); shader clock "
           "%.0f MHz (block 0) / %.0f MHz (last block); one MFMA per 32 cycles per SIMD at that clock = %.0f TFLOP/s\n", waves,
           bpc, iters, zero ? "zero" : "non-zero", ms, flop / ms / 1e9, ms * 1e6 / per_simd, mhz0, mhz1,
           2.0 * 32 * 32 * 16 / 32.0 * mhz0 * 1e6 * 256 * 4 / 1e12);
  }
  return 0;
}
