// code_memory.h -- executable device memory for the code WeightAlign generates.
//
// A plan's generated code is position-independent words the kernel body jumps into (sconv_tiled.hip, code_base): all
// it needs from the system is device memory instructions can be fetched from.  hipMalloc'd memory is mapped without the
// execute permission; the HIP module loader (hipModuleLoadData) provides it, but spends ~0.6-1 ms per megabyte parsing
// and copying a code object (a res5 layer's 11 MB: 8-12 ms; the 16 ResNet layers on a broadcast receiver: 44-48 ms) and
// needs the code wrapped in one first.  The ROCm runtime's own allocator has the permission as a flag
// (hsa_amd_memory_pool_allocate, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG -- what its loader uses underneath): this file
// asks it directly, on the pool of the agent that owns the CURRENT HIP device's memory.
#ifndef ESCOIN_CODE_MEMORY_H_
#define ESCOIN_CODE_MEMORY_H_

#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace escoin {

// `bytes` of executable memory on the current HIP device (a whole number of allocation granules; readable and
// writable by kernels of this process like any device memory).  ESCOIN_* status; on any failure nothing is allocated
// and the caller falls back to the code object loader.
int code_mem_alloc(size_t bytes, void **ptr);
void code_mem_free(void *ptr);
// exec[0 .. code_bytes) = the code at dev_src (ordinary device memory), exec[code_bytes .. total_bytes) = s_nop; a
// kernel on `stream` (instruction caches are invalidated at every dispatch, so the next launch fetches what it wrote).
int code_mem_fill(void *exec, const void *dev_src, size_t code_bytes, size_t total_bytes, hipStream_t stream);

}  // namespace escoin
#endif
