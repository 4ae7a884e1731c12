// Hardware probe (not product code): cost of the format-2 stream-loop body and of its parts.
// One 8-wave workgroup per CU (2 waves/SIMD); each iteration = 4 groups of 3 records.
// Variants drop one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define PK4V(val, sel, xa, xb) \
  "v_pk_fma_f32 v[64:65], " val ", v[" #xa ":" #xa "+1], v[64:65] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[66:67], " val ", v[" #xa "+2:" #xa "+3], v[66:67] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[160:161], " val ", v[" #xb ":" #xb "+1], v[160:161] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n" \
  "v_pk_fma_f32 v[162:163], " val ", v[" #xb "+2:" #xb "+3], v[162:163] op_sel:[" #sel ",0,0] op_sel_hi:[" #sel ",1,1]\n"

// SET0: "s_set_gpr_idx_idx 0\n" or "" ; XRD: X reads or "" ; RFL: readfirstlane+lshr or "" ; SETR: per-record set_idx or ""
#define GROUP(xa, xb, ya, yb, pa, pb, mcur, mprev, SET0, XADDR, XRD, PRD, WAIT, SETI, RFL, IX1, IX2, CTL) \
  SET0 XADDR XRD PRD WAIT SETI(mprev) \
  PK4V("v[" #pa ":" #pa "+1]", 1, xa, xb) \
  RFL(mcur, pa) IX1(mcur) \
  PK4V("v[" #pa "+2:" #pa "+3]", 0, xa, xb) \
  IX2(mcur) \
  PK4V("v[" #pa "+2:" #pa "+3]", 1, xa, xb) CTL

#define S_SET0 "s_set_gpr_idx_idx 0\n"
#define S_XADDR "v_lshl_add_u32 v32, s36, 5, %[lbA]\n"
#define S_XRD(ya, yb) "ds_read_b128 v[" #ya ":" #ya "+3], v32\n ds_read_b128 v[" #yb ":" #yb "+3], v32 offset:1024\n"
#define S_PRD(pb) "ds_read_b128 v[" #pb ":" #pb "+3], v34 offset:16\n v_add_u32 v34, 16, v34\n"
#define S_SETI(m) "s_set_gpr_idx_idx " #m "\n"
#define S_RFL(m, pa) "v_readfirstlane_b32 " #m ", v" #pa "\n s_lshr_b32 s36, " #m ", 21\n"
#define S_IX1(m) "s_lshr_b32 s59, " #m ", 7\n s_set_gpr_idx_idx s59\n"
#define S_IX2(m) "s_bfe_u32 s58, " #m ", 0x7000e\n s_set_gpr_idx_idx s58\n"
#define S_CTL "s_add_u32 s32, s32, -1\n"
#define N_RFL(m, pa) ""
#define N_IX(m) ""
#define N_SETI(m) ""

template <int V>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(32))) k_v3(float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) lds[i] = 0.f;   // zeros: meta = 0 -> offsets 0
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lbA = (wave * 4096 + lane * 16) & 0xFFFF;
  const unsigned pbase = 65536 + wave * 2048;
  unsigned m0a = __builtin_amdgcn_readfirstlane(0xC000u);
  asm volatile("v_mov_b32 v34, %[pb]\n s_mov_b32 s36, 0\n s_mov_b32 s34, 0\n s_mov_b32 s37, 0\n s_mov_b32 s32, 100\n"
               "v_mov_b32 v52, 0\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 0.5\n v_mov_b32 v55, 0.5\n"
               "v_mov_b32 v56, 0\n v_mov_b32 v57, 0.5\n v_mov_b32 v58, 0.5\n v_mov_b32 v59, 0.5\n"
               ::[pb] "v"(pbase) : "v34", "s36", "s34", "s37", "s32", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
#define CLOB "memory", "scc", "m0", "s32", "s34", "s36", "s37", "s58", "s59", "v32", "v34", "v36", "v37", "v38", "v39", \
             "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
             "v56", "v57", "v58", "v59", "v64", "v65", "v66", "v67", "v160", "v161", "v162", "v163"
#define RUN4(SET0, XADDR, XRD0, XRD1, PRD0, PRD1, WAIT, SETI, RFL, IX1, IX2, CTL)                                     \
  asm volatile("s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"                                                      \
               GROUP(36, 40, 44, 48, 52, 56, s34, s37, SET0, XADDR, XRD0, PRD0, WAIT, SETI, RFL, IX1, IX2, CTL)   \
               GROUP(44, 48, 36, 40, 56, 52, s37, s34, SET0, XADDR, XRD1, PRD1, WAIT, SETI, RFL, IX1, IX2, CTL)   \
               GROUP(36, 40, 44, 48, 52, 56, s34, s37, SET0, XADDR, XRD0, PRD0, WAIT, SETI, RFL, IX1, IX2, CTL)   \
               GROUP(44, 48, 36, 40, 56, 52, s37, s34, SET0, XADDR, XRD1, PRD1, WAIT, SETI, RFL, IX1, IX2, CTL)   \
               "s_set_gpr_idx_off\n s_waitcnt lgkmcnt(0)\n v_mov_b32 v34, %[pb]\n"                                 \
               ::[a] "s"(m0a), [lbA] "v"(lbA), [pb] "v"(pbase) : CLOB)
  for (int it = 0; it < iters; ++it) {
    if (V == 0) RUN4(S_SET0, S_XADDR, S_XRD(44, 48), S_XRD(36, 40), S_PRD(56), S_PRD(52), "s_waitcnt lgkmcnt(3)\n", S_SETI, S_RFL, S_IX1, S_IX2, S_CTL);
    if (V == 1) RUN4(S_SET0, S_XADDR, "", "", S_PRD(56), S_PRD(52), "s_waitcnt lgkmcnt(1)\n", S_SETI, S_RFL, S_IX1, S_IX2, S_CTL);          // no X reads
    if (V == 2) RUN4(S_SET0, S_XADDR, S_XRD(44, 48), S_XRD(36, 40), S_PRD(56), S_PRD(52), "s_waitcnt lgkmcnt(3)\n", S_SETI, N_RFL, S_IX1, S_IX2, S_CTL);  // no rfl
    if (V == 3) RUN4(S_SET0, S_XADDR, S_XRD(44, 48), S_XRD(36, 40), S_PRD(56), S_PRD(52), "s_waitcnt lgkmcnt(3)\n", N_SETI, S_RFL, N_IX, N_IX, S_CTL);    // no index switching
    if (V == 4) RUN4("", "", "", "", "", "", "", N_SETI, N_RFL, N_IX, N_IX, "");                                  // FMAs only
    if (V == 5) RUN4(S_SET0, S_XADDR, S_XRD(44, 48), S_XRD(36, 40), S_PRD(56), S_PRD(52), "", S_SETI, S_RFL, S_IX1, S_IX2, S_CTL);            // no wait
    if (V == 6) RUN4("", S_XADDR, S_XRD(44, 48), S_XRD(36, 40), S_PRD(56), S_PRD(52), "s_waitcnt lgkmcnt(3)\n", S_SETI, S_RFL, S_IX1, S_IX2, S_CTL);       // no set 0 (wrong dst)
    if (V == 7) RUN4(S_SET0, S_XADDR, S_XRD(44, 48), S_XRD(36, 40), "", "", "s_waitcnt lgkmcnt(2)\n", S_SETI, S_RFL, S_IX1, S_IX2, S_CTL);                 // no payload read
  }
  float r0;
  asm volatile("v_add_f32 %0, v64, v66" : "=v"(r0)::"v64", "v66");
  if (r0 == 12345.678f) out[threadIdx.x] = r0;
}

template <int V>
static double run(int iters, float *dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)k_v3<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(k_v3<V>, dim3(256), dim3(512), 96 * 1024, 0, dout, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_v3<V>, dim3(256), dim3(512), 96 * 1024, 0, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 4);
}

int main() {
  float *dout; CK(hipMalloc(&dout, 1 << 20));
  const int it = 20000;
  printf("ns per group of 3 records (12 pk), 8 waves/CU:\n");
  printf("  full body                : %.1f\n", run<0>(it, dout));
  printf("  no X reads               : %.1f\n", run<1>(it, dout));
  printf("  no readfirstlane/lshr    : %.1f\n", run<2>(it, dout));
  printf("  no index switching       : %.1f\n", run<3>(it, dout));
  printf("  FMAs only                : %.1f\n", run<4>(it, dout));
  printf("  no lgkmcnt wait          : %.1f\n", run<5>(it, dout));
  printf("  no set_idx 0             : %.1f\n", run<6>(it, dout));
  printf("  no payload read          : %.1f\n", run<7>(it, dout));
  return 0;
}
