// Hardware probe (not product code): throughput of candidate per-record instruction mixes
// for the stream loop at 2 waves/SIMD (240-VGPR kernels), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REC_SMOV(v, m) "s_set_gpr_idx_idx " m "\n v_pk_fma_f32 v[64:65], " v ", v[40:41], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], " v ", v[42:43], v[66:67] op_sel_hi:[0,1,1]\n"
// T=1 readlane val + packed idx shift
#define REC_RL1(vreg) "v_readlane_b32 s20, " vreg ", s30\n s_lshr_b32 s24, s24, 8\n s_set_gpr_idx_idx s24\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67] op_sel_hi:[0,1,1]\n"
// T=2: same + 2 more pk on second accumulator half (v[160..])
#define REC_RL2(vreg) "v_readlane_b32 s20, " vreg ", s30\n s_lshr_b32 s24, s24, 8\n s_set_gpr_idx_idx s24\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[160:161], s[20:21], v[44:45], v[160:161] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[162:163], s[20:21], v[46:47], v[162:163] op_sel_hi:[0,1,1]\n"
// readlane into m0 + readlane val
#define REC_RLM0(vreg, mreg) "v_readlane_b32 s20, " vreg ", s30\n v_readlane_b32 s22, " mreg ", s30\n s_set_gpr_idx_idx s22\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67] op_sel_hi:[0,1,1]\n"
// T=2 with alternating val SGPRs (s20/s22) to break the dependency on one SGPR
#define REC_RL2B(vreg, sv) "v_readlane_b32 " sv ", " vreg ", s30\n s_lshr_b32 s24, s24, 8\n s_set_gpr_idx_idx s24\n v_pk_fma_f32 v[64:65], " sv "x, v[40:41], v[64:65] op_sel_hi:[0,1,1]\n"

template <int V>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) k_mix(float* out, int iters) {
  float seed = (float)threadIdx.x;
  unsigned m0a = __builtin_amdgcn_readfirstlane(0xC000u | 4);
  asm volatile(
      "v_mov_b32 v40, %[sd]\n v_mov_b32 v41, %[sd]\n v_mov_b32 v42, %[sd]\n v_mov_b32 v43, %[sd]\n"
      "v_mov_b32 v44, %[sd]\n v_mov_b32 v45, %[sd]\n v_mov_b32 v46, %[sd]\n v_mov_b32 v47, %[sd]\n"
      "v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.25\n v_mov_b32 v52, 0.125\n v_mov_b32 v53, 0x0c080400\n"
      "v_mov_b32 v54, 0xC004\n"
      "s_mov_b32 s20, 0.5\n s_mov_b32 s21, 0.5\n s_mov_b32 s22, 0.5\n s_mov_b32 s23, 0.5\n s_mov_b32 s30, 3\n s_mov_b32 s24, 0x0c080400\n"
      "s_mov_b32 s25, 0xC004\n s_mov_b32 s26, 0xC008\n"
      ::[sd] "v"(seed)
      : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s30", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50",
        "v51", "v52", "v53", "v54", "v239");
  for (int i = 64; i < 240; i += 1) { }  // (accumulators start undefined; only timing matters)
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
                   REC_SMOV("s[20:21]", "s25") REC_SMOV("s[22:23]", "s26") REC_SMOV("s[20:21]", "s25") REC_SMOV("s[22:23]", "s26")
                   "s_set_gpr_idx_off\n" ::[a] "s"(m0a) : "memory", "v64", "v65", "v66", "v67");
    } else if (V == 1) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n s_mov_b32 s24, 0x0c080400\n"
                   REC_RL1("v50") REC_RL1("v51") REC_RL1("v52") REC_RL1("v50")
                   "s_set_gpr_idx_off\n" ::[a] "s"(m0a) : "memory", "s20", "s24", "v64", "v65", "v66", "v67");
    } else if (V == 2) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n s_mov_b32 s24, 0x0c080400\n"
                   REC_RL2("v50") REC_RL2("v51") REC_RL2("v52") REC_RL2("v50")
                   "s_set_gpr_idx_off\n" ::[a] "s"(m0a) : "memory", "s20", "s24", "v64", "v65", "v66", "v67", "v160", "v161", "v162", "v163");
    } else if (V == 3) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
                   REC_RLM0("v50", "v54") REC_RLM0("v51", "v54") REC_RLM0("v52", "v54") REC_RLM0("v50", "v54")
                   "s_set_gpr_idx_off\n" ::[a] "s"(m0a) : "memory", "s20", "v64", "v65", "v66", "v67");
    } else if (V == 4) {  // pure pk (reference)
      asm volatile("v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67] op_sel_hi:[0,1,1]\n"
                   "v_pk_fma_f32 v[68:69], s[20:21], v[40:41], v[68:69] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[70:71], s[20:21], v[42:43], v[70:71] op_sel_hi:[0,1,1]\n"
                   "v_pk_fma_f32 v[72:73], s[20:21], v[40:41], v[72:73] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[74:75], s[20:21], v[42:43], v[74:75] op_sel_hi:[0,1,1]\n"
                   "v_pk_fma_f32 v[76:77], s[20:21], v[40:41], v[76:77] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[78:79], s[20:21], v[42:43], v[78:79] op_sel_hi:[0,1,1]\n"
                   ::: "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    }
  }
  float r0;
  asm volatile("v_add_f32 %0, v64, v66" : "=v"(r0)::"v64", "v66");
  if (r0 == 12345.678f) out[threadIdx.x] = r0;
}

template <int V>
static double run(int pk_per_iter, int iters, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, threads = 256;  // 2 blocks of 4 waves per CU -> 2 waves/SIMD
  hipLaunchKernelGGL(k_mix<V>, dim3(blocks), dim3(threads), 0, 0, dout, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mix<V>, dim3(blocks), dim3(threads), 0, 0, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 2 * 64.0 * pk_per_iter * (threads / 64) * blocks * (double)iters;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float* dout; CK(hipMalloc(&dout, 1 << 20));
  const int it = 20000;
  printf("2 waves/SIMD, 240 VGPR kernels (TFLOP/s of useful pk work):\n");
  printf("  V0 s_set_gpr_idx_idx + 2pk         : %.1f\n", run<0>(8, it, dout));
  printf("  V1 readlane+lshr+idx + 2pk  (T=1)  : %.1f\n", run<1>(8, it, dout));
  printf("  V2 readlane+lshr+idx + 4pk  (T=2)  : %.1f\n", run<2>(16, it, dout));
  printf("  V3 readlane val + readlane m0 + 2pk: %.1f\n", run<3>(8, it, dout));
  printf("  V4 pure pk                         : %.1f\n", run<4>(8, it, dout));
  return 0;
}
