// Probe: what an LDS-DMA instruction (buffer_load_dwordx4 ... lds) costs the wave that issues it,
// and how much of that is the rewrite of M0 between two of them.  Not product code.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_ldsdma tools/probes/probe_ldsdma.hip && /tmp/probe_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: M0 rewritten before every instruction; 1: once per 4 (instruction offsets); 2: never
__global__ void __launch_bounds__(512) k(const float *in, unsigned bytes, unsigned long long *out, int iters, int stride_kib) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned long long q = reinterpret_cast<unsigned long long>(in);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)q);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned voff = (blockIdx.x * 8 + wave) * 65536u + lane * 16u;
  unsigned lbase = wave * 8192u;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    unsigned la = __builtin_amdgcn_readfirstlane(lbase + (unsigned)(it & 1) * 4096u);
    unsigned so = __builtin_amdgcn_readfirstlane((unsigned)it * (unsigned)stride_kib * 1024u);
    if (MODE == 0) {
      asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds\n"
                   "s_add_u32 m0, m0, 0x400\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n"
                   "s_add_u32 m0, m0, 0x400\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n"
                   "s_add_u32 m0, m0, 0x400\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds\n"
                   :: "s"(la), "v"(voff), "s"(r), "s"(so) : "memory", "m0");
    } else if (MODE == 1) {
      asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds\n"
                   :: "s"(la), "v"(voff), "s"(r), "s"(so) : "memory", "m0");
    } else {
      if (it == 0) asm volatile("s_mov_b32 m0, %0\n s_nop 0" :: "s"(la) : "m0");
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n"
                   "buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds\n"
                   "buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds\n"
                   "buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds\n"
                   :: "v"(voff), "s"(r), "s"(so) : "memory");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();   // issue done (not landed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0;
  }
  if (lds[threadIdx.x] == 12345.f) out[0] = 1;
}

// Spread issue: per iteration `ndma` LDS-DMA instructions (0, 1, 2 or 4; one M0 write), then `nfma`
// dependent-free packed FMAs.  Does the DMA cost the wave anything when it is this thin?
__global__ void __launch_bounds__(512) kspread(const float *in, unsigned bytes, unsigned long long *out, int iters,
                                               int ndma, int nfma, int flags) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned long long q = reinterpret_cast<unsigned long long>(in);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)q);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned voff = (blockIdx.x * 8 + wave) * 262144u + lane * 16u;
  unsigned lbase = wave * 8192u;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0, x3 = x0, x4 = x0;
  float a0 = lane, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    unsigned la = __builtin_amdgcn_readfirstlane(lbase + (unsigned)(it & 1) * 4096u);
    unsigned so = __builtin_amdgcn_readfirstlane((unsigned)it * 4096u);
    if (flags & 2) {   // LDS reads in flight across the DMA, then a counted wait (the stream loop's pattern)
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(lane * 16u + 65536u) : "memory");
    }
    if (flags & 1) asm volatile("s_set_gpr_idx_off" ::: "m0");
    if (ndma >= 1) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(la), "v"(voff), "s"(r), "s"(so) : "memory", "m0");
    if (ndma >= 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" :: "v"(voff), "s"(r), "s"(so) : "memory");
    if (ndma >= 4) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds\n buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" :: "v"(voff), "s"(r), "s"(so) : "memory");
    if (flags & 1) asm volatile("s_set_gpr_idx_on %0, gpr_idx(SRC2,DST)" :: "s"(0) : "m0");
    if (flags & 2) {
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(x4) : "v"(lane * 16u + 65536u) : "memory");
      asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");     // the four older reads
      a0 += x0[0] + x1[1] + x2[2] + x3[3];
    }
    for (int f = 0; f < nfma; f += 8)
      asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                   "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0.5f));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
  if (flags & 1) asm volatile("s_set_gpr_idx_off" ::: "m0");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  a1 += x4[0];
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f || lds[threadIdx.x] == 12345.f) out[0] = 1;
}

template <int MODE>
static void run(const char *name, const float *din, unsigned bytes, unsigned long long *dout, int iters, int stride_kib, int grid) {
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 96 * 1024, 0, din, bytes, dout, iters, stride_kib);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 16);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < grid * 8; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  a /= grid * 8; b /= grid * 8;
  printf("%-34s iters %3d x4 instr, src stride %3d KiB: issue %7.0f cycles = %5.0f / instr; landed %7.0f = %5.0f / instr\n",
         name, iters, stride_kib, a, a / (4.0 * iters), b, b / (4.0 * iters));
}

int main() {
  const size_t bytes = (size_t)3 << 30;
  float *din; hipMalloc(&din, bytes); hipMemset(din, 0, bytes);
  unsigned long long *dout; hipMalloc(&dout, 1 << 20);
  for (int grid : {1, 256}) {
    printf("--- %d workgroup(s) of 8 waves\n", grid);
    for (int stride : {0, 4}) {          // 0: the same 4 KiB again and again (L2 / TCP hits); 4: streaming
      run<0>("M0 rewritten per instruction", din, (unsigned)bytes, dout, 8, stride, grid);
      run<1>("M0 once per 4 (instr. offsets)", din, (unsigned)bytes, dout, 8, stride, grid);
      run<2>("M0 never rewritten", din, (unsigned)bytes, dout, 8, stride, grid);
    }
  }
  printf("--- spread issue, 256 workgroups of 8 waves, 16 iterations\n");
  hipFuncSetAttribute((const void *)kspread, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int flags : {0, 1, 2, 3})
  for (int nfma : {256}) {
    double base = 0;
    for (int ndma : {0, 1, 2, 4}) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kspread, dim3(256), dim3(512), 96 * 1024, 0, din, (unsigned)bytes, dout, 16, ndma, nfma, flags);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * 16);
      hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
      double a = 0;
      for (int i = 0; i < 256 * 8; ++i) a += h[2 * i];
      a /= 256 * 8;
      if (ndma == 0) base = a;
      printf("[idx mode off/on around the DMA: %d, LDS reads in flight + counted wait: %d] %d FMAs + %d DMA per iteration: %7.0f cycles (+%5.0f = %4.0f per DMA instruction)\n",
             flags & 1, (flags >> 1) & 1, nfma, ndma, a / 16, (a - base) / 16, ndma ? (a - base) / 16 / ndma : 0.0);
    }
  }
  return 0;
}
