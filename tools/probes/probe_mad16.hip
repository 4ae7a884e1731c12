#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned meta = 0x12340000u | (unsigned)threadIdx.x | 0xFF80u;   // hi16 = 0x1234, low bits junk
  unsigned base = 1000u + threadIdx.x;
  unsigned r0, r1;
  asm volatile("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[1,0,0,0]" : "=v"(r0) : "v"(meta), "v"(base));
  asm volatile("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[0,0,0,0]" : "=v"(r1) : "v"(meta), "v"(base));
  out[threadIdx.x * 2] = r0;
  out[threadIdx.x * 2 + 1] = r1;
}
int main() {
  unsigned *d; hipMalloc(&d, 64 * 2 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane0: hi-sel %u (want %u)  lo-sel %u (want %u)\n", h[0], 0x1234u + 1000u, h[1], (0xFF80u | 0u) + 1000u);
  printf("lane5: hi-sel %u (want %u)  lo-sel %u (want %u)\n", h[10], 0x1234u + 1005u, h[11], (0xFF80u | 5u) + 1005u);
  return 0;
}
