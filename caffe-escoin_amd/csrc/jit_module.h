// jit_module.h -- generated machine code -> device memory a kernel can call into.
#ifndef ESCOIN_JIT_MODULE_H_
#define ESCOIN_JIT_MODULE_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

namespace escoin {

struct JitModule {
  hipModule_t module = nullptr;       // code object loader: the loaded module ...
  void *direct = nullptr;             // ... or executable device memory the library filled itself (code_memory.h)
  unsigned long long code_base = 0;   // device address of the first byte of the generated code
  size_t code_bytes = 0;
};

// False when generated code is switched off in this process (ESCOIN_JIT=0); the LDS-staged stream
// kernel is then what KERNEL_AUTO picks.  (The library links libamd_comgr, a part of every ROCm
// install and what the HIP runtime itself loads kernels with; a failure inside it at WeightAlign
// makes KERNEL_AUTO fall back to the stream kernel with a message under ESCOIN_VERBOSE.)
bool jit_available();

// Puts `code` (jit_codegen.h: position-independent, called through code_base + offset) where the current device can
// execute it.  loader 0: executable device memory straight from the ROCm runtime's allocator, filled by a copy kernel
// (code_memory.h: no code object, no assembler, ~0.1 ms per megabyte) -- and if that is not to be had on this system,
// what loader 1 always does: the code wrapped in a code object -- a three-line assembly file that .incbin's the bytes
// behind a locator kernel, built in process (jit_wrap / jit_assemble below) -- loaded by the HIP module loader, the
// locator asked where the code landed (0.6-1 ms per megabyte).  ESCOIN_* status.
int jit_load(const uint32_t *code, size_t words, JitModule *out, hipStream_t stream, int loader = 0);
inline int jit_load(const std::vector<uint32_t> &code, JitModule *out, hipStream_t stream, int loader = 0) {
  return jit_load(code.data(), code.size(), out, stream, loader);
}
// The two ways to the code object's bytes (no device involved; both testable on a CPU-only box):
//   jit_assemble  the assembler and linker run on the wrapper around `code` (19 ms per megabyte of code);
//   jit_wrap      the SAME bytes without them: a template -- the wrapper around 4 KiB of s_nop, assembled once per
//                 process -- whose .text is grown by the code (padded to whole pages with s_nop): everything behind
//                 the insertion point moves by a whole number of pages, so file offsets, addresses and segment
//                 alignments move together, and the headers, section table and symbols are rewritten accordingly.
//                 tests/test_jit_codegen.py holds the two byte-identical.
// The code object path of jit_load uses jit_wrap and falls back to jit_assemble (ESCOIN_JIT_WRAP=0: always the assembler).
int jit_assemble(const std::vector<uint32_t> &code, std::vector<char> *elf);
int jit_wrap(const std::vector<uint32_t> &code, std::vector<char> *elf);
// Loads a code object jit_wrap / jit_assemble produced.
int jit_load_elf(const std::vector<char> &elf, size_t code_bytes, JitModule *out, hipStream_t stream);
void jit_unload(JitModule *m);

}  // namespace escoin
#endif
