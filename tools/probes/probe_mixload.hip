// Probe: how fast a CU fills LDS from global memory -- through LDS-DMA (buffer_load ... lds), through
// registers (global_load_dwordx4 + ds_write_b128), or with half of its waves on each path.  One
// 8-wave workgroup per CU, every workgroup streams its own region; data from HBM (1 GiB, read once
// per pass) or from L2 / Infinity Cache (small region re-read).  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mixload tools/probes/probe_mixload.hip && /tmp/probe_mixload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode: 0 all DMA, 1 all register path, 2 waves 0-3 DMA / 4-7 registers, 3 = 2 with 2:6, 4 = 6:2
__global__ void __launch_bounds__(512) k(const float *in, unsigned long long region_bytes, int iters, int mode, float *sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *base = reinterpret_cast<const char *>(in) + (size_t)blockIdx.x * region_bytes;
  const unsigned long long q = reinterpret_cast<unsigned long long>(base);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)q);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)region_bytes);
  r[3] = 0x00020000u;
  const int n_dma = mode == 0 ? 8 : mode == 1 ? 0 : mode == 2 ? 4 : mode == 3 ? 2 : 6;
  const bool dma = (int)wave < n_dma;
  // every wave moves 4 KiB per iteration: 4 instructions of 1 KiB
  const unsigned lbase = wave * 8192u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned off = (unsigned)(((size_t)it * 8 + wave) * 4096u % region_bytes);
    const unsigned la = __builtin_amdgcn_readfirstlane(lbase + (unsigned)(it & 1) * 4096u);
    if (dma) {
      asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds\n"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds\n"
                   :: "s"(la), "v"(lane * 16u), "s"(r), "s"(off) : "memory", "m0");
      if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      const float4 *p = reinterpret_cast<const float4 *>(base + off) + lane;
      const float4 a = p[0], b = p[64], c = p[128], d = p[192];
      float4 *l = reinterpret_cast<float4 *>(reinterpret_cast<char *>(lds) + la) + lane;
      l[0] = a; l[64] = b; l[128] = c; l[192] = d;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc = lds[threadIdx.x];
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

int main() {
  const size_t total = 1ull << 30;
  float *d, *sink;
  CK(hipMalloc(&d, total));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(d, 0, total));
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const char *names[5] = {"all DMA", "all registers", "4 DMA : 4 reg", "2 DMA : 6 reg", "6 DMA : 2 reg"};
  for (int src = 0; src < 2; ++src) {
    const unsigned long long region = src == 0 ? total / 256 : 256 * 1024;   // HBM stream / cache-resident
    printf("%s\n", src == 0 ? "from HBM (4 MiB per workgroup, read once)" : "from cache (256 KiB per workgroup, re-read)");
    for (int mode = 0; mode < 5; ++mode) {
      const int iters = src == 0 ? (int)(region / 32768) : 512;
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d, region, iters, mode, sink);
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int reps = 5;
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, d, region, iters, mode, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = 256.0 * iters * 32768.0 * reps;
      printf("  %-16s %8.3f ms  %6.2f TB/s  %6.1f GB/s per CU\n", names[mode], ms / reps, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 1e9);
    }
  }
  return 0;
}
