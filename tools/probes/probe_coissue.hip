// probe_coissue.hip -- VERDICT r5 item 5a: can the matrix pipe take work while the generated-code walk saturates the
// vector issue?  The walk is v_pk_fma_f32 (SGPR-pair weight x VGPR-pair input -> VGPR-pair accumulator) at two waves
// per SIMD (256 registers each); the fp32 matrix instruction is v_mfma_f32_32x32x2_f32 (4096 flops per wave, 64 cycles).
//
//   hipcc --offload-arch=gfx950 -O2 -o probe_coissue probe_coissue.hip && ./probe_coissue
//
// Modes (one 512-thread workgroup per CU, 8 waves = 2 per SIMD, every wave 256 VGPRs like the walk's kernel):
//   valu      every wave: 48 independent v_pk_fma_f32 per loop body
//   mfma      every wave: 4 independent chains of v_mfma_f32_32x32x2_f32
//   split     waves 0-3 (one per SIMD) run the valu body, waves 4-7 the mfma body -- two pipes, two waves
//   mixK      every wave interleaves 1 MFMA per K packed FMAs in ONE instruction stream (K = 48, 24, 12, 6)
// Reported per mode: shader cycles per packed FMA and per MFMA per SIMD, the TFLOP/s of each pipe, the clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kPk = 48;       // packed FMAs per body (independent accumulators: 96 VGPRs)
constexpr int kIter = 4000;

struct Out { unsigned long long cycles, realtime; };

__device__ __forceinline__ void pk_block(v2f (&acc)[kPk], const v2f &x, unsigned long long w, int from, int to) {
#pragma unroll
  for (int i = from; i < to; ++i)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(w), "v"(x));
}

// MODE 0 valu, 1 mfma, 2 split by wave, 3.. mix with K = kPk >> (MODE - 3) packed FMAs per MFMA
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(256))) probe(Out *out, float *sink, float seed) {
  const int wave = threadIdx.x >> 6;
  v2f acc[kPk];
  for (int i = 0; i < kPk; ++i) acc[i] = v2f{seed * i, seed};
  v16f m[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) m[j][e] = seed * (j + e);
  const v2f x = {seed + threadIdx.x, seed * 0.5f};
  const float a = seed * 1.5f, b = seed + 2.f;
  unsigned long long w = __builtin_amdgcn_readfirstlane(__float_as_uint(seed)) | ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(seed * 3.f)) << 32);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  const bool do_valu = MODE == 0 || (MODE == 2 && wave < 4), do_mfma = MODE == 1 || (MODE == 2 && wave >= 4);
  if constexpr (MODE <= 2) {
    if (do_valu)
      for (int it = 0; it < kIter; ++it) pk_block(acc, x, w, 0, kPk);
    if (do_mfma)
      for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m[j], 0, 0, 0);
      }
  } else {
    constexpr int K = kPk >> (MODE >= 3 ? MODE - 3 : 0);          // packed FMAs per MFMA
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
      for (int g = 0; g < kPk / K; ++g) {
        m[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m[g & 3], 0, 0, 0);
        pk_block(acc, x, w, g * K, g * K + K);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < kPk; ++i) s += acc[i][0] + acc[i][1];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) s += m[j][e];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    out[blockIdx.x * 8 + wave].cycles = t1 - t0;
    out[blockIdx.x * 8 + wave].realtime = r1 - r0;
  }
}

template <int MODE>
static void run(const char *name, int pk_per_wave_iter, int mfma_per_wave_iter, int valu_waves, int mfma_waves) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  Out *d_out;
  float *d_sink;
  CHECK(hipMalloc(&d_out, sizeof(Out) * cus * 8));
  CHECK(hipMalloc(&d_sink, 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 96 * 1024, 0, d_out, d_sink, 1.0f + rep);   // (96 KB of LDS: one workgroup per CU)
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Out> h((size_t)cus * 8);
  CHECK(hipMemcpy(h.data(), d_out, sizeof(Out) * h.size(), hipMemcpyDeviceToHost));
  double cyc_v = 0, cyc_m = 0, rt = 0;
  int nv = 0, nm = 0;
  for (int b = 0; b < cus; ++b)
    for (int w = 0; w < 8; ++w) {
      const bool is_m = (MODE == 1) || (MODE == 2 && w >= 4);
      const bool is_v = (MODE == 0) || (MODE == 2 && w < 4) || MODE >= 3;
      if (is_v) { cyc_v += (double)h[b * 8 + w].cycles; ++nv; }
      if (is_m || MODE >= 3) { cyc_m += (double)h[b * 8 + w].cycles; ++nm; }
      rt += (double)h[b * 8 + w].realtime;
    }
  const double mean_cyc = (cyc_v + cyc_m) / std::max(1, nv + nm);
  const double ghz = mean_cyc / (rt / (cus * 8) * 10.0);        // s_memrealtime: 100 MHz
  // per SIMD: waves_on_simd x instructions / cycles
  const double pk_total = (double)pk_per_wave_iter * kIter, mf_total = (double)mfma_per_wave_iter * kIter;
  const double cyc_per_pk = nv ? (cyc_v / nv) / (pk_total * (valu_waves / 4.0)) : 0;      // cycles of a SIMD per packed FMA it issued
  const double cyc_per_mf = nm ? (cyc_m / nm) / (mf_total * (mfma_waves / 4.0)) : 0;
  const double secs = ms * 1e-3;
  const double tf_v = pk_total * valu_waves * cus * 256.0 / secs * 1e-12;      // 64 lanes x 2 x 2 flops
  const double tf_m = mf_total * mfma_waves * cus * 4096.0 / secs * 1e-12;     // 32 x 32 x 2 x 2 flops
  printf("%-7s kernel %.3f ms  clock %.2f GHz | packed FMA: %5.2f cycles of a SIMD each, %6.1f TFLOP/s | MFMA 32x32x2 f32: %6.1f cycles of a SIMD each, %6.1f TFLOP/s | sum %6.1f\n",
         name, ms, ghz, pk_total > 0 ? cyc_per_pk : 0.0, tf_v, mf_total > 0 ? cyc_per_mf : 0.0, tf_m, tf_v + tf_m);
  CHECK(hipFree(d_out));
  CHECK(hipFree(d_sink));
}

int main() {
  printf("# probe_coissue: v_pk_fma_f32 (SGPR-pair weight) against v_mfma_f32_32x32x2_f32, 2 waves per SIMD, one workgroup per CU\n");
  run<0>("valu", kPk, 0, 8, 0);
  run<1>("mfma", 0, 4, 0, 8);
  run<2>("split", kPk, 4, 4, 4);
  run<3>("mix48", kPk, 1, 8, 8);
  run<4>("mix24", kPk, 2, 8, 8);
  run<5>("mix12", kPk, 4, 8, 8);
  run<6>("mix6", kPk, 8, 8, 8);
  return 0;
}
