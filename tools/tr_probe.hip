// probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 value = element index; lane l supplies byte address 8*l (its own
// 4-element chunk).  Prints, for every lane, the 4 element indices it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t* out, int mode) {
  __shared__ uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned addr;
  const int l = threadIdx.x;
  if (mode == 0) addr = 8 * l;                                  // chunk l
  else addr = (unsigned)(((l >> 2) * 32 + (l & 3) * 8));        // row (l>>2) of a 16-element-pitch matrix, cols 4*(l&3)..+4
  addr += (unsigned)(uintptr_t)lds;                             // LDS byte address
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 2) ? "\n" : "   |   ");
  }
  return 0;
}
