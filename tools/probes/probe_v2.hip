// Hardware probe (not product code): per-group cost of the current stream-loop body (values and
// row offset broadcast with v_readlane out of a lane-distributed VGPR window) against a variant
// whose values arrive as a broadcast ds_read_b128 and feed v_pk_fma_f32 as a VGPR pair with
// op_sel.  2 waves/SIMD, GPR-index mode on, 3 records per group, T=2 (4 pk per record).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// X quads: v[36:39] (A), v[40:43] (B); second set v[44:47], v[48:51]
#define PK4S(val, xa, xb) \
  "v_pk_fma_f32 v[64:65], " val ", v[" #xa ":" #xa "+1], v[64:65] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[66:67], " val ", v[" #xa "+2:" #xa "+3], v[66:67] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[160:161], " val ", v[" #xb ":" #xb "+1], v[160:161] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[162:163], " val ", v[" #xb "+2:" #xb "+3], v[162:163] op_sel_hi:[0,1,1]\n"
// VGPR pair source, broadcast low half (op_sel 0) or high half (op_sel 1)
#define PK4V(val, sel, xa, xb) \
  "v_pk_fma_f32 v[64:65], " val ", v[" #xa ":" #xa "+1], v[64:65] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[66:67], " val ", v[" #xa "+2:" #xa "+3], v[66:67] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[160:161], " val ", v[" #xb ":" #xb "+1], v[160:161] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[162:163], " val ", v[" #xb "+2:" #xb "+3], v[162:163] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n"

// one group, current design, phase reading X set (xa,xb), prefetching into (ya,yb)
#define GROUP_CUR(xa, xb, ya, yb) \
  "v_readlane_b32 s36, v53, s30\n v_readlane_b32 s38, v55, s30\n v_readlane_b32 s40, v56, s30\n" \
  "s_waitcnt lgkmcnt(0)\n" \
  "s_set_gpr_idx_idx s36\n s_lshr_b32 s58, s36, 8\n" PK4S("s[38:39]", xa, xb) \
  "v_readlane_b32 s38, v57, s30\n" \
  "v_readlane_b32 s35, v52, s30\n s_set_gpr_idx_idx 0\n v_add_u32 v32, s35, %[lbA]\n v_add_u32 v33, s35, %[lbB]\n" \
  "ds_read_b128 v[" #ya ":" #ya "+3], v32\n ds_read_b128 v[" #yb ":" #yb "+3], v33\n" \
  "s_set_gpr_idx_idx s58\n s_lshr_b32 s59, s36, 16\n" PK4S("s[40:41]", xa, xb) \
  "s_set_gpr_idx_idx s59\n" PK4S("s[38:39]", xa, xb)

// one group, v2: payload quad P (ix, v0, v1, v2) in v[pa:pa+3], next payload prefetched into pb
#define GROUP_V2(xa, xb, ya, yb, pa, pb) \
  "s_waitcnt lgkmcnt(0)\n" \
  "v_readfirstlane_b32 s36, v" #pa "\n" \
  "s_set_gpr_idx_idx s36\n s_lshr_b32 s58, s36, 8\n" PK4V("v[" #pa ":" #pa "+1]", 1, xa, xb) \
  "v_readlane_b32 s35, v52, s30\n s_set_gpr_idx_idx 0\n v_add_u32 v32, s35, %[lbA]\n v_add_u32 v33, s35, %[lbB]\n v_add_u32 v34, 16, v34\n" \
  "ds_read_b128 v[" #ya ":" #ya "+3], v32\n ds_read_b128 v[" #yb ":" #yb "+3], v33\n ds_read_b128 v[" #pb ":" #pb "+3], %[pbase]\n" \
  "s_set_gpr_idx_idx s58\n s_lshr_b32 s59, s36, 16\n" PK4V("v[" #pa "+2:" #pa "+3]", 0, xa, xb) \
  "s_set_gpr_idx_idx s59\n" PK4V("v[" #pa "+2:" #pa "+3]", 1, xa, xb)

template <int V>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(32))) k_v2(float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lbA = (wave * 4096 + lane * 16) & 0xFFFF, lbB = lbA + 1024;
  const unsigned pbase = 32768 + wave * 64;   // uniform per wave: broadcast read
  unsigned m0a = __builtin_amdgcn_readfirstlane(0xC000u);
  asm volatile(
      "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v55, 0.5\n v_mov_b32 v56, 0.25\n v_mov_b32 v57, 0.125\n"
      "v_mov_b32 v34, 0\n s_mov_b32 s30, 5\n"
      "v_mov_b32 v60, 0\n v_mov_b32 v61, 0.5\n v_mov_b32 v62, 0.5\n v_mov_b32 v63, 0.5\n"
      "v_mov_b32 v56, 0\n v_mov_b32 v57, 0.5\n v_mov_b32 v58, 0.5\n v_mov_b32 v59, 0.5\n"
      ::: "v52", "v53", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v34", "s30");
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
                   GROUP_CUR(36, 40, 44, 48) GROUP_CUR(44, 48, 36, 40) GROUP_CUR(36, 40, 44, 48) GROUP_CUR(44, 48, 36, 40)
                   "s_set_gpr_idx_off\n s_waitcnt lgkmcnt(0)\n"
                   ::[a] "s"(m0a), [lbA] "v"(lbA), [lbB] "v"(lbB)
                   : "memory", "scc", "m0", "s35", "s36", "s38", "s39", "s40", "s41", "s58", "s59", "v32", "v33", "v36", "v37", "v38", "v39",
                     "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v64", "v65", "v66", "v67",
                     "v160", "v161", "v162", "v163");
    } else if (V == 1) {
      asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
                   GROUP_V2(36, 40, 44, 48, 56, 60) GROUP_V2(44, 48, 36, 40, 60, 56) GROUP_V2(36, 40, 44, 48, 56, 60) GROUP_V2(44, 48, 36, 40, 60, 56)
                   "s_set_gpr_idx_off\n s_waitcnt lgkmcnt(0)\n"
                   ::[a] "s"(m0a), [lbA] "v"(lbA), [lbB] "v"(lbB), [pbase] "v"(pbase)
                   : "memory", "scc", "m0", "s35", "s36", "s58", "s59", "v32", "v33", "v34", "v36", "v37", "v38", "v39",
                     "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v56", "v57", "v58", "v59",
                     "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v160", "v161", "v162", "v163");
    } else if (V == 2) {   // pk only, SGPR source
      asm volatile(PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40)
                   PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40)
                   PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40)
                   PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40) PK4S("s[38:39]", 36, 40)
                   ::: "memory", "v64", "v65", "v66", "v67", "v160", "v161", "v162", "v163");
    } else if (V == 3) {   // pk only, VGPR pair source with op_sel
      asm volatile(PK4V("v[56:57]", 1, 36, 40) PK4V("v[58:59]", 0, 36, 40) PK4V("v[58:59]", 1, 36, 40)
                   PK4V("v[56:57]", 1, 36, 40) PK4V("v[58:59]", 0, 36, 40) PK4V("v[58:59]", 1, 36, 40)
                   PK4V("v[56:57]", 1, 36, 40) PK4V("v[58:59]", 0, 36, 40) PK4V("v[58:59]", 1, 36, 40)
                   PK4V("v[56:57]", 1, 36, 40) PK4V("v[58:59]", 0, 36, 40) PK4V("v[58:59]", 1, 36, 40)
                   ::: "memory", "v64", "v65", "v66", "v67", "v160", "v161", "v162", "v163");
    }
  }
  float r0;
  asm volatile("v_add_f32 %0, v64, v66" : "=v"(r0)::"v64", "v66");
  if (r0 == 12345.678f) out[threadIdx.x] = r0;
}

template <int V>
static double run(int iters, float *dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)k_v2<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  const int blocks = 256, threads = 512;   // one 8-wave workgroup per CU: 2 waves/SIMD
  hipLaunchKernelGGL(k_v2<V>, dim3(blocks), dim3(threads), 96 * 1024, 0, dout, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_v2<V>, dim3(blocks), dim3(threads), 96 * 1024, 0, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 4);   // ns per group (4 groups per iteration)
}

int main() {
  float *dout; CK(hipMalloc(&dout, 1 << 20));
  const int it = 20000;
  printf("ns per group of 3 records (12 pk), 8 waves/CU:\n");
  printf("  current (readlane window)     : %.1f\n", run<0>(it, dout));
  printf("  v2 (broadcast ds_read payload): %.1f\n", run<1>(it, dout));
  printf("  12 pk only, SGPR src0         : %.1f\n", run<2>(it, dout));
  printf("  12 pk only, VGPR src0 + op_sel: %.1f\n", run<3>(it, dout));
  return 0;
}
