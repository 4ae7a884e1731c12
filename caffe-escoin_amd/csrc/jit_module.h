// jit_module.h -- generated machine code -> a loaded code object whose code a kernel can call.
#ifndef ESCOIN_JIT_MODULE_H_
#define ESCOIN_JIT_MODULE_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

namespace escoin {

struct JitModule {
  hipModule_t module = nullptr;
  unsigned long long code_base = 0;   // device address of the first byte of the generated code
  size_t code_bytes = 0;
};

// False when generated code cannot be used in this process (ESCOIN_JIT=0, or the code object
// manager cannot be reached); the LDS-staged stream kernel is then what KERNEL_AUTO picks.
bool jit_available();

// Wraps `code` (jit_codegen.h) in a code object -- a three-line assembly file that .incbin's the
// bytes behind a locator kernel, assembled and linked in process by the ROCm code object manager
// (libamd_comgr, the library the HIP runtime itself loads kernels with) -- loads it on the
// current device and asks the locator where the code landed.  ESCOIN_* status.
int jit_load(const std::vector<uint32_t> &code, JitModule *out, hipStream_t stream);
void jit_unload(JitModule *m);

}  // namespace escoin
#endif
