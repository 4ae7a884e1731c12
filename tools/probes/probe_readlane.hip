// Hardware probe: issue cost of v_readlane_b32 / v_readfirstlane_b32 / broadcast ds_read vs plain VALU,
// 2 waves per SIMD (240-VGPR kernel), gfx950.  Reports ns per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define R8(x) x x x x x x x x
template <int V>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) k(float* out, int iters) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned addr = 64;
  asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 2.0\n s_mov_b32 s30, 5\n v_mov_b32 v239, 0\n" ::: "v40", "v41", "s30", "v239");
  for (int it = 0; it < iters; ++it) {
    if (V == 0) asm volatile(R8("v_mov_b32 v42, v40\n") ::: "v42");
    if (V == 1) asm volatile("v_readlane_b32 s20, v40, s30\n v_readlane_b32 s21, v41, s30\n v_readlane_b32 s22, v40, s30\n v_readlane_b32 s23, v41, s30\n"
                             "v_readlane_b32 s24, v40, s30\n v_readlane_b32 s25, v41, s30\n v_readlane_b32 s26, v40, s30\n v_readlane_b32 s27, v41, s30\n"
                             ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (V == 2) asm volatile("v_readfirstlane_b32 s20, v40\n v_readfirstlane_b32 s21, v41\n v_readfirstlane_b32 s22, v40\n v_readfirstlane_b32 s23, v41\n"
                             "v_readfirstlane_b32 s24, v40\n v_readfirstlane_b32 s25, v41\n v_readfirstlane_b32 s26, v40\n v_readfirstlane_b32 s27, v41\n"
                             ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (V == 3) asm volatile("ds_read_b128 v[44:47], %0\n ds_read_b128 v[48:51], %0 offset:16\n ds_read_b128 v[52:55], %0 offset:32\n ds_read_b128 v[56:59], %0 offset:48\n"
                             "ds_read_b128 v[44:47], %0 offset:64\n ds_read_b128 v[48:51], %0 offset:80\n ds_read_b128 v[52:55], %0 offset:96\n ds_read_b128 v[56:59], %0 offset:112\n s_waitcnt lgkmcnt(0)\n"
                             :: "v"(addr) : "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
    if (V == 4) asm volatile(R8("s_mov_b32 s20, s30\n") ::: "s20");
    if (V == 5) asm volatile("v_pk_fma_f32 v[60:61], v[40:41], v[40:41], v[60:61]\n v_pk_fma_f32 v[62:63], v[40:41], v[40:41], v[62:63]\n v_pk_fma_f32 v[64:65], v[40:41], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], v[40:41], v[40:41], v[66:67]\n"
                             "v_pk_fma_f32 v[68:69], v[40:41], v[40:41], v[68:69]\n v_pk_fma_f32 v[70:71], v[40:41], v[40:41], v[70:71]\n v_pk_fma_f32 v[72:73], v[40:41], v[40:41], v[72:73]\n v_pk_fma_f32 v[74:75], v[40:41], v[40:41], v[74:75]\n"
                             ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
  }
  if (iters < 0) out[threadIdx.x] = lds[0];
}
template <int V> static double run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(512), dim3(256), 4096, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(512), dim3(256), 4096, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 8);   // ns per instruction per wave (all waves run concurrently)
}
int main() {
  float* d; CK(hipMalloc(&d, 4096));
  const int it = 20000;
  printf("ns per instruction per wave at 2 waves/SIMD (includes C++ loop overhead, ~3 instr per 8):\n");
  printf("  v_mov_b32            %.2f\n", run<0>(d, it));
  printf("  v_readlane_b32       %.2f\n", run<1>(d, it));
  printf("  v_readfirstlane_b32  %.2f\n", run<2>(d, it));
  printf("  ds_read_b128 (bcast) %.2f\n", run<3>(d, it));
  printf("  s_mov_b32            %.2f\n", run<4>(d, it));
  printf("  v_pk_fma_f32         %.2f\n", run<5>(d, it));
  return 0;
}
