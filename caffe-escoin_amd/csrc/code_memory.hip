// code_memory.hip -- see code_memory.h
#include "code_memory.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "escoin_plan.h"

namespace escoin {

namespace {

constexpr int kMaxDevices = 64;
struct DevicePool {
  bool tried = false, ok = false;
  hsa_amd_memory_pool_t pool{};
  size_t granule = 4096;
};
std::mutex g_mu;
DevicePool g_pools[kMaxDevices];
bool g_hsa_tried = false, g_hsa_ok = false;

std::string hsa_error(hsa_status_t s, const char *what) {
  const char *m = nullptr;
  if (hsa_status_string(s, &m) != HSA_STATUS_SUCCESS || !m) m = "?";
  return std::string(what) + ": " + m;
}

// The device-local, coarse-grained, allocatable pool of an agent: where hipMalloc's memory comes from.
hsa_status_t pick_pool(hsa_amd_memory_pool_t pool, void *data) {
  DevicePool *out = static_cast<DevicePool *>(data);
  hsa_amd_segment_t seg;
  if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
    return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  bool can_alloc = false;
  if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags) != HSA_STATUS_SUCCESS ||
      hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &can_alloc) != HSA_STATUS_SUCCESS)
    return HSA_STATUS_SUCCESS;
  if (!can_alloc || !(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) return HSA_STATUS_SUCCESS;
  size_t granule = 0;
  if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_GRANULE, &granule) != HSA_STATUS_SUCCESS || granule == 0)
    granule = 4096;
  out->pool = pool;
  out->granule = granule;
  out->ok = true;
  return HSA_STATUS_INFO_BREAK;
}

// The pool for the current HIP device, looked up once per device: the agent is the owner of a probe allocation made
// through HIP on that device (no enumeration order or bus address to match: HIP_VISIBLE_DEVICES and ROCR_VISIBLE_DEVICES
// renumber the two views independently).
int pool_for_current_device(DevicePool *out) {
  int dev = 0;
  ESCOIN_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(ESCOIN_EINVAL, "code memory: device index out of range");
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_hsa_tried) {
    g_hsa_tried = true;
    const hsa_status_t s = hsa_init();          // (reference-counted: the HIP runtime holds the first one)
    g_hsa_ok = s == HSA_STATUS_SUCCESS;
    if (!g_hsa_ok) return fail(ESCOIN_EHIP, hsa_error(s, "code memory: hsa_init"));
  }
  if (!g_hsa_ok) return fail(ESCOIN_EHIP, "code memory: the ROCm runtime is not available");
  DevicePool &d = g_pools[dev];
  if (!d.tried) {
    d.tried = true;
    void *probe = nullptr;
    if (hipMalloc(&probe, 256) == hipSuccess && probe) {
      hsa_amd_pointer_info_t info;
      info.size = sizeof(info);
      const hsa_status_t s = hsa_amd_pointer_info(probe, &info, nullptr, nullptr, nullptr);
      if (s == HSA_STATUS_SUCCESS && info.type == HSA_EXT_POINTER_TYPE_HSA && info.agentOwner.handle != 0) {
        hsa_device_type_t type;
        if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_GPU)
          (void)hsa_amd_agent_iterate_memory_pools(info.agentOwner, pick_pool, &d);
      }
      (void)hipFree(probe);
    }
    if (getenv("ESCOIN_VERBOSE"))
      fprintf(stderr, "[escoin] code memory: device %d: %s\n", dev, d.ok ? "executable pool found" : "no executable pool (code object loader)");
  }
  if (!d.ok) return fail(ESCOIN_EHIP, "code memory: no allocatable device pool");
  *out = d;
  return ESCOIN_OK;
}

}  // namespace

int code_mem_alloc(size_t bytes, void **ptr) {
  *ptr = nullptr;
  if (bytes == 0) return fail(ESCOIN_EINVAL, "code memory: empty");
  DevicePool d;
  const int rc = pool_for_current_device(&d);
  if (rc != ESCOIN_OK) return rc;
  const size_t rounded = (bytes + d.granule - 1) / d.granule * d.granule;
  void *p = nullptr;
  const hsa_status_t s = hsa_amd_memory_pool_allocate(d.pool, rounded, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, &p);
  if (s != HSA_STATUS_SUCCESS || !p) return fail(ESCOIN_ENOMEM, hsa_error(s, "code memory: hsa_amd_memory_pool_allocate"));
  *ptr = p;
  return ESCOIN_OK;
}

void code_mem_free(void *ptr) {
  if (ptr) (void)hsa_amd_memory_pool_free(ptr);
}

// dst[0 .. code_words) = src, dst[code_words .. total_words) = s_nop: the instruction prefetcher runs ahead of the last
// instruction, and what it finds there must be mapped and harmless (the code object wrapper pads the same way).
__global__ void __launch_bounds__(256) escoin_code_copy_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                                               size_t code_words, size_t total_words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_words; i += stride)
    dst[i] = i < code_words ? src[i] : 0xBF800000u;
}

int code_mem_fill(void *exec, const void *dev_src, size_t code_bytes, size_t total_bytes, hipStream_t stream) {
  if (!exec || !dev_src || (code_bytes & 3) || (total_bytes & 3) || code_bytes > total_bytes)
    return fail(ESCOIN_EINVAL, "code memory: bad fill");
  const size_t words = total_bytes / 4;
  const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 4096);
  hipLaunchKernelGGL(escoin_code_copy_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint32_t *>(exec),
                     static_cast<const uint32_t *>(dev_src), code_bytes / 4, words);
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

}  // namespace escoin
