// probe_rsrc_range.hip -- does the buffer descriptor's range check (raw buffer, stride 0, offen) see the scalar
// offset?  One wave loads through a descriptor of 1024 records (bytes) from a 64 KiB allocation filled with
// i + 1: (a) voffset inside, soffset 0; (b) voffset inside, soffset pushing the address past num_records (still
// inside the allocation); (c) voffset past num_records; and the same three as LDS-DMA (buffer_load ... lds).
//   hipcc --offload-arch=gfx950 -O2 -o probe_rsrc_range probe_rsrc_range.hip && ./probe_rsrc_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned *buf, unsigned *out) {
  __shared__ unsigned lds[1024];
  const unsigned long long q = reinterpret_cast<unsigned long long>(buf);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)q);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32) & 0xFFFFu);
  r[2] = 1024u;
  r[3] = 0x00020000u;
  const unsigned lane = threadIdx.x;
  const unsigned vin = lane * 4u, vout = 2048u + lane * 4u;
  unsigned a, b, c;
  const unsigned s0 = 0u, s1 = 2048u;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n s_waitcnt vmcnt(0)" : "=v"(a) : "v"(vin), "s"(r), "s"(s0) : "memory");
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n s_waitcnt vmcnt(0)" : "=v"(b) : "v"(vin), "s"(r), "s"(s1) : "memory");
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n s_waitcnt vmcnt(0)" : "=v"(c) : "v"(vout), "s"(r), "s"(s0) : "memory");
  out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xDEADu;
  __syncthreads();
  const unsigned l0 = (unsigned)(size_t)&lds[0], l1 = l0 + 256u, l2 = l0 + 512u;
  asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dword %1, %2, %3 offen lds\n s_waitcnt vmcnt(0)" :: "s"(l0), "v"(vin), "s"(r), "s"(s0) : "memory", "m0");
  asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dword %1, %2, %3 offen lds\n s_waitcnt vmcnt(0)" :: "s"(l1), "v"(vin), "s"(r), "s"(s1) : "memory", "m0");
  asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dword %1, %2, %3 offen lds\n s_waitcnt vmcnt(0)" :: "s"(l2), "v"(vout), "s"(r), "s"(s0) : "memory", "m0");
  __syncthreads();
  out[192 + lane] = lds[lane]; out[256 + lane] = lds[64 + lane]; out[320 + lane] = lds[128 + lane];
}

int main() {
  const int n = 16384;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned)i + 1u;
  unsigned *d, *o;
  if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&o, 384 * 4) != hipSuccess) return 2;
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  std::vector<unsigned> r(384);
  if (hipMemcpy(r.data(), o, 384 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  const char *names[6] = {"load  voffset in range, soffset 0          ", "load  voffset in range, soffset past records",
                          "load  voffset past records                 ", "lds   voffset in range, soffset 0          ",
                          "lds   voffset in range, soffset past records", "lds   voffset past records                 "};
  for (int k = 0; k < 6; ++k) printf("%s: lane 0 -> %u, lane 63 -> %u\n", names[k], r[64 * k], r[64 * k + 63]);
  printf("(data = index + 1: in range 1 / 64; at +2048 bytes 513 / 576; dropped by the range check 0)\n");
  return 0;
}
