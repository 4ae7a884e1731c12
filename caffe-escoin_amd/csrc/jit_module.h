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

// False when generated code is switched off in this process (ESCOIN_JIT=0); the LDS-staged stream
// kernel is then what KERNEL_AUTO picks.  (The library links libamd_comgr, a part of every ROCm
// install and what the HIP runtime itself loads kernels with; a failure inside it at WeightAlign
// makes KERNEL_AUTO fall back to the stream kernel with a message under ESCOIN_VERBOSE.)
bool jit_available();

// Wraps `code` (jit_codegen.h) in a code object -- a three-line assembly file that .incbin's the
// bytes behind a locator kernel, assembled and linked in process by the ROCm code object manager
// (libamd_comgr, the library the HIP runtime itself loads kernels with) -- loads it on the
// current device and asks the locator where the code landed.  ESCOIN_* status.
// keep_elf != nullptr: the code object's bytes are handed back (escoin_plan_export_aligned persists them).
int jit_load(const std::vector<uint32_t> &code, JitModule *out, hipStream_t stream, std::vector<char> *keep_elf = nullptr);
// The two ways to the code object's bytes (no device involved; both testable on a CPU-only box):
//   jit_assemble  the assembler and linker run on the wrapper around `code` (19 ms per megabyte of code);
//   jit_wrap      the SAME bytes without them: a template -- the wrapper around 4 KiB of s_nop, assembled once per
//                 process -- whose .text is grown by the code (padded to whole pages with s_nop): everything behind
//                 the insertion point moves by a whole number of pages, so file offsets, addresses and segment
//                 alignments move together, and the headers, section table and symbols are rewritten accordingly.
//                 tests/test_jit_codegen.py holds the two byte-identical.
// jit_load uses jit_wrap and falls back to jit_assemble (ESCOIN_JIT_WRAP=0: always the assembler).
int jit_assemble(const std::vector<uint32_t> &code, std::vector<char> *elf);
int jit_wrap(const std::vector<uint32_t> &code, std::vector<char> *elf);
// Loads a code object jit_load produced earlier (same library build, same target): no assembler run.
int jit_load_elf(const std::vector<char> &elf, size_t code_bytes, JitModule *out, hipStream_t stream);
void jit_unload(JitModule *m);

}  // namespace escoin
#endif
