#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, int iters) {
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      asm volatile(REP16("v_pk_fma_f32 v[40:41], v[60:61], v[62:63], v[40:41]\n v_pk_fma_f32 v[42:43], v[60:61], v[62:63], v[42:43]\n v_pk_fma_f32 v[44:45], v[60:61], v[62:63], v[44:45]\n v_pk_fma_f32 v[46:47], v[60:61], v[62:63], v[46:47]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    } else if (KIND == 1) {
      asm volatile(REP16("v_fma_f32 v40, v60, v62, v40\n v_fma_f32 v42, v60, v62, v42\n v_fma_f32 v44, v60, v62, v44\n v_fma_f32 v46, v60, v62, v46\n") ::: "v40","v42","v44","v46");
    } else {
      asm volatile(REP16("v_pk_fma_f32 v[40:41], s[40:41], v[62:63], v[40:41] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[42:43], s[40:41], v[62:63], v[42:43] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[44:45], s[40:41], v[62:63], v[44:45] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[46:47], s[40:41], v[62:63], v[46:47] op_sel_hi:[0,1,1]\n") ::: "v40","v41","v42","v43","v44","v45","v46","v47");
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 4] = c1 - c0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + 1] = r1 - r0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + 2] = r0; }
}
template <int KIND> int run(const char *name, unsigned long long *d) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  for (int wgs : {1, 256, 512}) for (int waves : {1, 4, 8, 16}) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(64 * waves), 0, 0, d, 10); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(64 * waves), 0, 0, d, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    static unsigned long long h[512 * 16 * 4]; CK(hipMemcpy(h, d, sizeof(unsigned long long) * wgs * 16 * 4, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; unsigned long long s0 = ~0ull, s1 = 0;
    for (int w = 0; w < wgs; ++w) { cyc += h[w * 64]; rt += h[w * 64 + 1]; if (h[w*64+2] < s0) s0 = h[w*64+2]; if (h[w*64+2] > s1) s1 = h[w*64+2]; }
    const double ninstr = 64.0 * iters;
    printf("%-10s CUs=%d wgs=%3d waves/wg=%2d  event %.3f ms  in-kernel %.3f ms  %.3f GHz  %.2f cycles/instr/wave  start spread %.3f ms\n", name, prop.multiProcessorCount, wgs, waves, ms,
           rt / wgs * 1e-5, cyc / (rt * 10), cyc / wgs / ninstr, (s1 - s0) * 1e-5);
  }
  return 0;
}
int main() {
  unsigned long long *d; CK(hipMalloc(&d, 512 * 16 * 4 * 8));
  run<0>("pk_fma", d); run<1>("fma", d); run<2>("pk_fma_sgpr", d);
  return 0;
}
