// probe_hwid.hip -- which bits of HW_REG_HW_ID tell the two co-resident 256-thread workgroups of a CU apart?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256, 2) k(unsigned *out, unsigned long long *t) {
  __shared__ float pad[12 * 1024];   // 48 KB: two workgroups per CU, like the dense kernel
  pad[threadIdx.x] = 1.f;
  unsigned id, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // stay resident long enough for the whole grid to be placed
  while (__builtin_amdgcn_s_memrealtime() - t0 < 20000) __builtin_amdgcn_s_sleep(10);
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t0;
  }
  if (pad[threadIdx.x] == 2.f) out[0] = 0;
}
int main() {
  const int n = 512;
  unsigned *d; unsigned long long *dt;
  hipMalloc(&d, n * 4 * 2 * 4); hipMalloc(&dt, n * 4 * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, dt);
  std::vector<unsigned> h(n * 8); std::vector<unsigned long long> ht(n * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, ht.size() * 8, hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] tg_id[19:16] vm_id[23:20] queue_id[26:24] state_id[29:27] me_id[31:30]
  std::map<unsigned, std::vector<int>> by_cu;
  for (int b = 0; b < n; ++b) {
    const unsigned id = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 0xF;
    const unsigned key = (xcc << 16) | (id & 0xFF00);    // xcc, se, sh, cu
    by_cu[key].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) for %d workgroups\n", by_cu.size(), n);
  int shown = 0;
  for (auto &kv : by_cu) {
    if (shown++ >= 6) break;
    printf("cu key %05x:", kv.first);
    for (int b : kv.second) {
      printf("  wg %3d [", b);
      for (int w = 0; w < 4; ++w) { const unsigned id = h[(b * 4 + w) * 2]; printf(" simd%u.wave%u", (id >> 4) & 3, id & 15); }
      printf(" ]");
    }
    printf("\n");
  }
  // statistics: parity of wave_id of wave 0 among co-resident pairs; block index difference
  int pairs = 0, diffpar = 0; std::map<int,int> dist;
  for (auto &kv : by_cu) if (kv.second.size() == 2) {
    ++pairs;
    const unsigned a = h[(kv.second[0] * 4) * 2] & 15, b = h[(kv.second[1] * 4) * 2] & 15;
    if ((a ^ b) & 1) ++diffpar;
    dist[kv.second[1] - kv.second[0]]++;
  }
  printf("%d CUs with two workgroups; wave_id parity differs in %d; block index distance histogram:", pairs, diffpar);
  for (auto &kv : dist) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
}
