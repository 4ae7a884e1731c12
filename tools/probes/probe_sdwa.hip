// Hardware probe: does GPR-index mode (VDST relative) relocate the destination of SDWA / DPP /
// VOP3 encodings of v_add_u32?  Writes distinct values and reports where they landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __attribute__((amdgpu_num_vgpr(32))) k(unsigned* out) {
  unsigned a = threadIdx.x + 1000;
  unsigned r[12];
  asm volatile(
      "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"
      "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
      "s_mov_b32 s20, 4\n"
      "s_mov_b32 s21, 0x00070005\n"
      "s_set_gpr_idx_on s20, gpr_idx(SRC2,DST)\n"
      "v_add_u32_e32 v40, s20, %[a]\n"                                                     // VOP2: expect v44
      "v_add_u32_sdwa v41, s20, %[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"   // SDWA: v41 or v45?
      "v_add_u32_sdwa v42, s21, %[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"  // uses low 16 bits (5)
      "v_add_u32_dpp v43, %[a], %[a] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"     // DPP: v43 or v47?
      "s_set_gpr_idx_off\n"
      ::[a] "v"(a) : "s20", "s21", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  asm volatile("v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v45\n"
               "v_mov_b32 %6, v46\n v_mov_b32 %7, v47\n v_mov_b32 %8, v48\n v_mov_b32 %9, v49\n v_mov_b32 %10, v50\n v_mov_b32 %11, v51\n"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]));
  if (threadIdx.x == 1) for (int i = 0; i < 12; ++i) out[i] = r[i];
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 64));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  unsigned h[12]; CK(hipMemcpy(h, d, 48, hipMemcpyDeviceToHost));
  printf("lane1 a=1001; v40..v51 =");
  for (int i = 0; i < 12; ++i) printf(" %u", h[i]);
  printf("\n(VOP2 add of 4 -> 1005 expected at v44 if relocated; SDWA add -> 1005 at v41 (not relocated) or v45; SDWA WORD_0 of 0x70005 -> 1006 at v42 or v46; DPP a+a=2002 at v43 or v47)\n");
  return 0;
}
