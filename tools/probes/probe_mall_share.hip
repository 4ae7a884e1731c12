// probe_mall_share.hip -- when all eight XCDs read the SAME bytes within one launch, who serves the re-reads?
//
// The XCD grouping of workgroup columns (sconv_tiled.hip, xcd_q) makes every XCD read the whole bottom blob itself:
// rocprofv3's FETCH_SIZE then shows 4-5x the algorithmic bytes for res5 / AlexNet conv3 (profiles/traffic_*.json).
// FETCH_SIZE counts requests at the L2 / fabric interface (TCC_EA0_RDREQ); the Infinity Cache sits behind the fabric
// and the TCC counters cannot tell a hit in it from a DRAM read (TCC_EA0_RDREQ_DRAM counts the requests' DESTINATION
// type, DRAM vs GMI vs IO).  So this probe times it: 2048 workgroups stream a buffer
//   mode A "disjoint": every byte read by exactly one workgroup (one XCD)            -> B bytes through the L2s
//   mode B "shared"  : the 256 workgroups of each XCD partition the WHOLE buffer      -> 8 B bytes through the L2s
// HBM-cold: a ring of buffers larger than the 256 MiB Infinity Cache, one launch per buffer.  If mode B takes about as
// long as mode A, seven of the eight reads of every line were served on-die (Infinity Cache) and the DRAM traffic of
// such a layer is its algorithmic traffic, whatever FETCH_SIZE says; if it takes ~8x as long, they went to HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 2048 workgroups (eight per CU: enough loads in flight to saturate the memory system -- a first version with one
// 4-wave workgroup per CU was latency-bound at 2 TB/s and said nothing), eight independent 16-byte loads per lane
__global__ void __launch_bounds__(256) stream_kernel(const float4 *__restrict__ buf, size_t n16, int shared, float *sink) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int xcd = wg & 7, k = wg >> 3, per_xcd = nwg >> 3;
  // shared: workgroup k of its XCD takes the k-th part of the whole buffer; disjoint: workgroup wg the wg-th part
  const size_t parts = shared ? (size_t)per_xcd : (size_t)nwg;
  const size_t part = shared ? (size_t)k : (size_t)wg;
  const size_t lo = n16 * part / parts, hi = n16 * (part + 1) / parts;
  float4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = float4{0.f, 0.f, 0.f, 0.f};
  size_t i = lo + threadIdx.x;
  for (; i + 7 * 256 < hi; i += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = buf[i + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; i < hi; i += 256) { const float4 v = buf[i]; acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w; }
  float t = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) t += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
  if (t == 123.456f) sink[wg & 1023] = t + (float)xcd;     // (never: keeps the loads)
}

int main() {
  const size_t sizes_mb[] = {26, 51, 103, 206};
  float *sink = nullptr;
  CHECK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("2048 workgroups x 256 lanes, eight 16-byte loads in flight per lane, ring of buffers > 256 MiB (every launch reads a buffer the Infinity Cache no longer holds)\n");
  printf("%8s %6s | %12s %12s | %12s %12s | %s\n", "MB", "ring", "disjoint us", "GB/s", "shared us", "L2-side GB/s", "shared / disjoint time");
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20, n16 = bytes / 16;
    const int ring = (int)((600u << 20) / bytes) + 1;
    std::vector<float4 *> bufs(ring, nullptr);
    for (auto &b : bufs) {
      CHECK(hipMalloc(&b, bytes));
      CHECK(hipMemset(b, 0, bytes));
    }
    double t[2] = {0, 0};
    for (int shared = 0; shared < 2; ++shared) {
      for (int i = 0; i < ring; ++i) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, bufs[i], n16, shared, sink);   // warm-up round
      CHECK(hipDeviceSynchronize());
      const int rounds = 3;
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < rounds; ++r)
        for (int i = 0; i < ring; ++i) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, bufs[i], n16, shared, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      t[shared] = ms * 1e3 / (rounds * ring);
    }
    printf("%8zu %6d | %12.1f %12.0f | %12.1f %12.0f | %.2f\n", mb, ring, t[0], bytes / t[0] * 1e-3, t[1], 8.0 * bytes / t[1] * 1e-3, t[1] / t[0]);
    for (auto &b : bufs) CHECK(hipFree(b));
  }
  return 0;
}
