// Hardware probe (not product code): GPR-index mode semantics and VALU issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---------- correctness: relative dst/src2 with VOP3 v_fma_f32 and VOP3P v_pk_fma_f32 ----------
__global__ void __attribute__((amdgpu_num_vgpr(32))) k_sem(float* out, const float* in, int idx4) {
  // accumulators v64..v79 zeroed; x in v40..v43 = in[lane*4+e]
  int lane = threadIdx.x;
  float x0 = in[lane * 4 + 0], x1 = in[lane * 4 + 1], x2 = in[lane * 4 + 2], x3 = in[lane * 4 + 3];
  unsigned m0v = 0xC000u | (unsigned)idx4;
  m0v = __builtin_amdgcn_readfirstlane(m0v);
  float val = 2.0f;
  unsigned long long valpair = 0;  // unused
  asm volatile(
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n"
      "v_mov_b32 v40, %[x0]\n v_mov_b32 v41, %[x1]\n v_mov_b32 v42, %[x2]\n v_mov_b32 v43, %[x3]\n"
      "s_mov_b32 s20, 2.0\n"
      "s_mov_b32 s21, 2.0\n"
      "s_set_gpr_idx_on %[m0v], gpr_idx(SRC2,DST)\n"
      "s_mov_b32 m0, %[m0v]\n"
      "s_nop 0\n"
      // VOP3 fma into acc[idx4 + 0..1]
      "v_fma_f32 v64, s20, v40, v64\n"
      "v_fma_f32 v65, s20, v41, v65\n"
      // VOP3P pk_fma into acc[idx4 + 2..3]: src0 = s[20:21] (2.0, 2.0)
      "v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
      "s_set_gpr_idx_off\n"
      ::[x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m0v] "s"(m0v)
      : "memory", "s20", "s21", "v40", "v41", "v42", "v43", "v64", "v65", "v66", "v67", "v68", "v69", "v70",
        "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  float r[16];
  asm volatile(
      "v_mov_b32 %0, v64\n v_mov_b32 %1, v65\n v_mov_b32 %2, v66\n v_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\n v_mov_b32 %5, v69\n v_mov_b32 %6, v70\n v_mov_b32 %7, v71\n"
      "v_mov_b32 %8, v72\n v_mov_b32 %9, v73\n v_mov_b32 %10, v74\n v_mov_b32 %11, v75\n"
      "v_mov_b32 %12, v76\n v_mov_b32 %13, v77\n v_mov_b32 %14, v78\n v_mov_b32 %15, v79\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]),
        "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
      :: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
         "v78", "v79");
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = r[i];
  (void)val; (void)valpair;
}

// op_sel broadcast form: src0 = s[20:21] with op_sel_hi:[0,1,1] -> both halves use s20
__global__ void __attribute__((amdgpu_num_vgpr(32))) k_sem2(float* out, const float* in, int idx4) {
  int lane = threadIdx.x;
  float x0 = in[lane * 4 + 0], x1 = in[lane * 4 + 1];
  unsigned m0v = __builtin_amdgcn_readfirstlane(0xC000u | (unsigned)idx4);
  asm volatile(
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
      "v_mov_b32 v40, %[x0]\n v_mov_b32 v41, %[x1]\n"
      "s_mov_b32 s20, 3.0\n"
      "s_mov_b32 s21, 100.0\n"
      "s_set_gpr_idx_on %[m0v], gpr_idx(SRC2,DST)\n"
      "s_nop 0\n"
      "v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65] op_sel_hi:[0,1,1]\n"
      // does ds / plain VOP2 get affected? v_add_u32 with dst relative: expect v[72+idx]... skip
      "s_set_gpr_idx_off\n"
      ::[x0] "v"(x0), [x1] "v"(x1), [m0v] "s"(m0v)
      : "memory", "s20", "s21", "v40", "v41", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
  float r[8];
  asm volatile(
      "v_mov_b32 %0, v64\n v_mov_b32 %1, v65\n v_mov_b32 %2, v66\n v_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\n v_mov_b32 %5, v69\n v_mov_b32 %6, v70\n v_mov_b32 %7, v71\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
      :: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
  for (int i = 0; i < 8; ++i) out[lane * 8 + i] = r[i];
}

// ---------- throughput: N iterations of a block of FMAs, variants ----------
template <int VARIANT>
__global__ void __attribute__((amdgpu_num_vgpr(32))) k_rate(float* out, int iters, const unsigned* recs) {
  unsigned m0a = __builtin_amdgcn_readfirstlane(recs[0]);
  unsigned m0b = __builtin_amdgcn_readfirstlane(recs[1]);
  float seed = (float)threadIdx.x;
  asm volatile(
      "v_mov_b32 v40, %[sd]\n v_mov_b32 v41, %[sd]\n v_mov_b32 v42, %[sd]\n v_mov_b32 v43, %[sd]\n"
      "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n"
      "s_mov_b32 s20, 0.5\n s_mov_b32 s21, 0.5\n"
      ::[sd] "v"(seed)
      : "s20", "s21", "v40", "v41", "v42", "v43", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",
        "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0) {  // 16 plain v_fma_f32, no indexing
      asm volatile(
          "v_fma_f32 v64, s20, v40, v64\n v_fma_f32 v65, s20, v41, v65\n v_fma_f32 v66, s20, v42, v66\n v_fma_f32 v67, s20, v43, v67\n"
          "v_fma_f32 v68, s20, v40, v68\n v_fma_f32 v69, s20, v41, v69\n v_fma_f32 v70, s20, v42, v70\n v_fma_f32 v71, s20, v43, v71\n"
          "v_fma_f32 v72, s20, v40, v72\n v_fma_f32 v73, s20, v41, v73\n v_fma_f32 v74, s20, v42, v74\n v_fma_f32 v75, s20, v43, v75\n"
          "v_fma_f32 v76, s20, v40, v76\n v_fma_f32 v77, s20, v41, v77\n v_fma_f32 v78, s20, v42, v78\n v_fma_f32 v79, s20, v43, v79\n"
          ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    } else if (VARIANT == 1) {  // 8 v_pk_fma_f32 (same flops), no indexing
      asm volatile(
          "v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
          "v_pk_fma_f32 v[68:69], s[20:21], v[40:41], v[68:69]\n v_pk_fma_f32 v[70:71], s[20:21], v[42:43], v[70:71]\n"
          "v_pk_fma_f32 v[72:73], s[20:21], v[40:41], v[72:73]\n v_pk_fma_f32 v[74:75], s[20:21], v[42:43], v[74:75]\n"
          "v_pk_fma_f32 v[76:77], s[20:21], v[40:41], v[76:77]\n v_pk_fma_f32 v[78:79], s[20:21], v[42:43], v[78:79]\n"
          ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    } else if (VARIANT == 2) {  // indexed: 4x (s_mov m0 + 4 v_fma) = record pattern
      asm volatile(
          "s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
          "s_mov_b32 m0, %[a]\n s_nop 0\n v_fma_f32 v64, s20, v40, v64\n v_fma_f32 v65, s20, v41, v65\n v_fma_f32 v66, s20, v42, v66\n v_fma_f32 v67, s20, v43, v67\n"
          "s_mov_b32 m0, %[b]\n s_nop 0\n v_fma_f32 v64, s20, v40, v64\n v_fma_f32 v65, s20, v41, v65\n v_fma_f32 v66, s20, v42, v66\n v_fma_f32 v67, s20, v43, v67\n"
          "s_mov_b32 m0, %[a]\n s_nop 0\n v_fma_f32 v64, s21, v40, v64\n v_fma_f32 v65, s21, v41, v65\n v_fma_f32 v66, s21, v42, v66\n v_fma_f32 v67, s21, v43, v67\n"
          "s_mov_b32 m0, %[b]\n s_nop 0\n v_fma_f32 v64, s21, v40, v64\n v_fma_f32 v65, s21, v41, v65\n v_fma_f32 v66, s21, v42, v66\n v_fma_f32 v67, s21, v43, v67\n"
          "s_set_gpr_idx_off\n"
          ::[a] "s"(m0a), [b] "s"(m0b)
          : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    } else if (VARIANT == 3) {  // indexed pk: 4x (s_mov m0 + 2 v_pk_fma)
      asm volatile(
          "s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
          "s_mov_b32 m0, %[a]\n s_nop 0\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
          "s_mov_b32 m0, %[b]\n s_nop 0\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
          "s_mov_b32 m0, %[a]\n s_nop 0\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
          "s_mov_b32 m0, %[b]\n s_nop 0\n v_pk_fma_f32 v[64:65], s[20:21], v[40:41], v[64:65]\n v_pk_fma_f32 v[66:67], s[20:21], v[42:43], v[66:67]\n"
          "s_set_gpr_idx_off\n"
          ::[a] "s"(m0a), [b] "s"(m0b)
          : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    } else if (VARIANT == 4) {  // indexed, no s_nop after s_mov m0 (hazard/perf check)
      asm volatile(
          "s_set_gpr_idx_on %[a], gpr_idx(SRC2,DST)\n"
          "s_mov_b32 m0, %[a]\n v_fma_f32 v64, s20, v40, v64\n v_fma_f32 v65, s20, v41, v65\n v_fma_f32 v66, s20, v42, v66\n v_fma_f32 v67, s20, v43, v67\n"
          "s_mov_b32 m0, %[b]\n v_fma_f32 v64, s20, v40, v64\n v_fma_f32 v65, s20, v41, v65\n v_fma_f32 v66, s20, v42, v66\n v_fma_f32 v67, s20, v43, v67\n"
          "s_mov_b32 m0, %[a]\n v_fma_f32 v64, s21, v40, v64\n v_fma_f32 v65, s21, v41, v65\n v_fma_f32 v66, s21, v42, v66\n v_fma_f32 v67, s21, v43, v67\n"
          "s_mov_b32 m0, %[b]\n v_fma_f32 v64, s21, v40, v64\n v_fma_f32 v65, s21, v41, v65\n v_fma_f32 v66, s21, v42, v66\n v_fma_f32 v67, s21, v43, v67\n"
          "s_set_gpr_idx_off\n"
          ::[a] "s"(m0a), [b] "s"(m0b)
          : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
    }
  }
  float r0, r1;
  asm volatile("v_add_f32 %0, v64, v68\n v_add_f32 %1, v72, v76" : "=v"(r0), "=v"(r1) :: "v64", "v68", "v72", "v76");
  if (r0 + r1 == 12345.678f) out[threadIdx.x] = r0;
}

template <int V>
static double run_rate(int blocks, int threads, int iters, float* dout, unsigned* drecs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(threads), 0, 0, dout, iters, drecs);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(threads), 0, 0, dout, iters, drecs);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 16 * 64.0 * (threads / 64) * blocks * (double)iters;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *din, *dout; unsigned* drecs;
  CK(hipMalloc(&din, 64 * 4 * 4)); CK(hipMalloc(&dout, 1 << 20)); CK(hipMalloc(&drecs, 64));
  std::vector<float> in(256); for (int i = 0; i < 256; ++i) in[i] = 1.0f + i;
  CK(hipMemcpy(din, in.data(), 1024, hipMemcpyHostToDevice));
  unsigned recs[2] = {0xC000u | 4, 0xC000u | 8};
  CK(hipMemcpy(drecs, recs, 8, hipMemcpyHostToDevice));
  for (int idx4 : {0, 4, 8}) {
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dout, din, idx4);
    CK(hipDeviceSynchronize());
    std::vector<float> o(64 * 16); CK(hipMemcpy(o.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost));
    printf("k_sem idx4=%d lane1 acc[0..15]:", idx4);
    for (int i = 0; i < 16; ++i) printf(" %g", o[16 + i]);
    printf("   (x = 5 6 7 8; expect 2x at [idx4..idx4+3])\n");
  }
  for (int idx4 : {0, 2, 4}) {
    hipLaunchKernelGGL(k_sem2, dim3(1), dim3(64), 0, 0, dout, din, idx4);
    CK(hipDeviceSynchronize());
    std::vector<float> o(64 * 8); CK(hipMemcpy(o.data(), dout, 64 * 8 * 4, hipMemcpyDeviceToHost));
    printf("k_sem2 idx=%d lane1 acc[0..7]:", idx4);
    for (int i = 0; i < 8; ++i) printf(" %g", o[8 + i]);
    printf("   (x = 5 6; expect 15 18 at [idx..idx+1] if op_sel_hi broadcasts s20=3)\n");
  }
  const int iters = 20000;
  for (int wpc : {4, 8, 12, 16}) {  // waves per CU (256 CUs): blocks of 64*wpc threads? use blocks=256*wpc/4, 256 thr
    int threads = 256, blocks = 256 * wpc / 4;
    printf("waves/CU=%2d (waves/SIMD=%d): plain fma %.1f TF | pk_fma %.1f TF | idx fma %.1f TF | idx pk %.1f TF | idx fma nonop %.1f TF\n",
           wpc, wpc / 4, run_rate<0>(blocks, threads, iters, dout, drecs), run_rate<1>(blocks, threads, iters, dout, drecs),
           run_rate<2>(blocks, threads, iters, dout, drecs), run_rate<3>(blocks, threads, iters, dout, drecs),
           run_rate<4>(blocks, threads, iters, dout, drecs));
  }
  return 0;
}
