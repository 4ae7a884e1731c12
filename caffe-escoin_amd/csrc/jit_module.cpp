// jit_module.cpp -- see jit_module.h
#include "jit_module.h"

#include <amd_comgr/amd_comgr.h>
#include <elf.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "code_memory.h"
#include "escoin_plan.h"
#include "knobs.h"

namespace escoin {

namespace {

// The wrapper: a locator kernel (stores the address of escoin_jit_code through its one pointer
// argument) and the generated code behind it, in the executable segment of a code object the HIP
// runtime loads like any other.  Directives as hipcc emits them for gfx950 (code object v6).
const char *kWrapper = R"(
	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.amdhsa_code_object_version 6
	.text
	.protected	escoin_jit_locator
	.globl	escoin_jit_locator
	.p2align	8
	.type	escoin_jit_locator,@function
escoin_jit_locator:
	s_load_dwordx2 s[0:1], s[0:1], 0x0
	s_getpc_b64 s[2:3]
.Lpc:
	s_add_u32 s2, s2, escoin_jit_code-.Lpc
	s_addc_u32 s3, s3, 0
	v_mov_b32_e32 v2, 0
	v_mov_b32_e32 v0, s2
	v_mov_b32_e32 v1, s3
	s_waitcnt lgkmcnt(0)
	global_store_dwordx2 v2, v[0:1], s[0:1]
	s_endpgm
.Lfunc_end0:
	.size	escoin_jit_locator, .Lfunc_end0-escoin_jit_locator
	.p2align	8
	.globl	escoin_jit_code
escoin_jit_code:
	.incbin "%BLOB%"
	.fill 64, 4, 0xBF800000
	.section	.rodata,"a",@progbits
	.p2align	6, 0x0
	.amdhsa_kernel escoin_jit_locator
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 8
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 3
		.amdhsa_next_free_sgpr 4
		.amdhsa_accum_offset 4
		.amdhsa_reserve_vcc 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     0
    .args:
      - .address_space:  global
        .offset:         0
        .size:           8
        .value_kind:     global_buffer
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 8
    .max_flat_workgroup_size: 1024
    .name:           escoin_jit_locator
    .private_segment_fixed_size: 0
    .sgpr_count:     8
    .sgpr_spill_count: 0
    .symbol:         escoin_jit_locator.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     3
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
)";

std::string comgr_error(amd_comgr_status_t s, const char *what) {
  const char *m = "?";
  (void)amd_comgr_status_string(s, &m);
  return std::string(what) + ": " + m;
}

#define ESCOIN_CG_TRY(expr)                                                             \
  do {                                                                                  \
    amd_comgr_status_t s__ = (expr);                                                    \
    if (s__ != AMD_COMGR_STATUS_SUCCESS) { err = comgr_error(s__, #expr); goto done; }  \
  } while (0)

// source text -> executable code object (ELF bytes); "" on success, else the error
std::string assemble_and_link(const std::string &src, std::vector<char> *elf) {
  std::string err;
  amd_comgr_data_t d{}, out{};
  amd_comgr_data_set_t in{}, rel{}, exe{};
  amd_comgr_action_info_t info{};
  bool have_d = false, have_in = false, have_rel = false, have_exe = false, have_info = false, have_out = false;
  size_t sz = 0;
  ESCOIN_CG_TRY(amd_comgr_create_data(AMD_COMGR_DATA_KIND_SOURCE, &d));
  have_d = true;
  ESCOIN_CG_TRY(amd_comgr_set_data(d, src.size(), src.data()));
  ESCOIN_CG_TRY(amd_comgr_set_data_name(d, "escoin_jit.s"));
  ESCOIN_CG_TRY(amd_comgr_create_data_set(&in));
  have_in = true;
  ESCOIN_CG_TRY(amd_comgr_create_data_set(&rel));
  have_rel = true;
  ESCOIN_CG_TRY(amd_comgr_create_data_set(&exe));
  have_exe = true;
  ESCOIN_CG_TRY(amd_comgr_data_set_add(in, d));
  ESCOIN_CG_TRY(amd_comgr_create_action_info(&info));
  have_info = true;
  ESCOIN_CG_TRY(amd_comgr_action_info_set_isa_name(info, "amdgcn-amd-amdhsa--gfx950"));
  ESCOIN_CG_TRY(amd_comgr_do_action(AMD_COMGR_ACTION_ASSEMBLE_SOURCE_TO_RELOCATABLE, info, in, rel));
  ESCOIN_CG_TRY(amd_comgr_do_action(AMD_COMGR_ACTION_LINK_RELOCATABLE_TO_EXECUTABLE, info, rel, exe));
  ESCOIN_CG_TRY(amd_comgr_action_data_get_data(exe, AMD_COMGR_DATA_KIND_EXECUTABLE, 0, &out));
  have_out = true;
  ESCOIN_CG_TRY(amd_comgr_get_data(out, &sz, nullptr));
  elf->resize(sz);
  ESCOIN_CG_TRY(amd_comgr_get_data(out, &sz, elf->data()));
done:
  if (have_out) (void)amd_comgr_release_data(out);
  if (have_d) (void)amd_comgr_release_data(d);
  if (have_in) (void)amd_comgr_destroy_data_set(in);
  if (have_rel) (void)amd_comgr_destroy_data_set(rel);
  if (have_exe) (void)amd_comgr_destroy_data_set(exe);
  if (have_info) (void)amd_comgr_destroy_action_info(info);
  return err;
}

}  // namespace

bool jit_available() {
  static const bool on = (ESC_KNOB("JIT", 1) != 0);
  return on;
}

// Where the blob's temporary file goes: $TMPDIR unless it holds a character the assembler's string
// syntax would need escaped (the path is spliced into an .incbin directive).
static std::string tmp_dir() {
  const char *t = getenv("TMPDIR");
  if (!t || !*t) return "/tmp";
  for (const char *c = t; *c; ++c)
    if (*c == '"' || *c == '\\' || *c == '\n' || *c == '\r') return "/tmp";
  return t;
}

int jit_load_elf(const std::vector<char> &elf, size_t code_bytes, JitModule *out, hipStream_t stream) {
  if (elf.size() < sizeof(Elf64_Ehdr) || std::memcmp(elf.data(), "\177ELF", 4) != 0) return fail(ESCOIN_EINVAL, "jit: not a code object");
  {
    // hipModuleLoadData takes no length: the header's tables must lie inside the bytes we hold (a truncated or
    // foreign blob reaches here through escoin_plan_import_aligned)
    Elf64_Ehdr eh;
    std::memcpy(&eh, elf.data(), sizeof(eh));
    const uint64_t n = elf.size();
    if (eh.e_ident[EI_CLASS] != ELFCLASS64 || eh.e_machine != 224 /* EM_AMDGPU */ || eh.e_phentsize != sizeof(Elf64_Phdr) ||
        eh.e_shentsize != sizeof(Elf64_Shdr) || eh.e_phoff > n || (uint64_t)eh.e_phnum * sizeof(Elf64_Phdr) > n - eh.e_phoff ||
        eh.e_shoff > n || (uint64_t)eh.e_shnum * sizeof(Elf64_Shdr) > n - eh.e_shoff || code_bytes > n)
      return fail(ESCOIN_EINVAL, "jit: code object header out of bounds");
    for (unsigned i = 0; i < eh.e_phnum; ++i) {
      Elf64_Phdr ph;
      std::memcpy(&ph, elf.data() + eh.e_phoff + (size_t)i * sizeof(ph), sizeof(ph));
      if (ph.p_offset > n || ph.p_filesz > n - ph.p_offset) return fail(ESCOIN_EINVAL, "jit: code object segment out of bounds");
    }
    for (unsigned i = 0; i < eh.e_shnum; ++i) {
      Elf64_Shdr sh;
      std::memcpy(&sh, elf.data() + eh.e_shoff + (size_t)i * sizeof(sh), sizeof(sh));
      if (sh.sh_type != SHT_NOBITS && sh.sh_type != SHT_NULL && (sh.sh_offset > n || sh.sh_size > n - sh.sh_offset))
        return fail(ESCOIN_EINVAL, "jit: code object section out of bounds");
    }
  }
  JitModule m;
  ESCOIN_HIP_TRY(hipModuleLoadData(&m.module, elf.data()));
  hipFunction_t locator = nullptr;
  unsigned long long *d_addr = nullptr;
  hipError_t e = hipModuleGetFunction(&locator, m.module, "escoin_jit_locator");
  if (e == hipSuccess) e = hipMalloc(&d_addr, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(d_addr, 0, sizeof(unsigned long long), stream);
  if (e == hipSuccess) {
    void *args[] = {&d_addr};
    e = hipModuleLaunchKernel(locator, 1, 1, 1, 1, 1, 1, 0, stream, args, nullptr);
  }
  unsigned long long addr = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&addr, d_addr, sizeof(addr), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (d_addr) (void)hipFree(d_addr);
  if (e != hipSuccess || addr == 0) {
    (void)hipModuleUnload(m.module);
    return fail(ESCOIN_EHIP, std::string("jit: locating the generated code failed: ") + hipGetErrorString(e));
  }
  m.code_base = addr;
  m.code_bytes = code_bytes;
  *out = m;
  return ESCOIN_OK;
}

int jit_assemble(const std::vector<uint32_t> &code, std::vector<char> *elf) {
  if (code.empty()) return fail(ESCOIN_EINVAL, "jit: empty program");
  // the bytes reach the assembler through .incbin: a file, gone again before this returns
  std::string path = tmp_dir() + "/escoin_jit_XXXXXX";
  std::vector<char> pbuf(path.begin(), path.end());
  pbuf.push_back(0);
  const int fd = mkstemp(pbuf.data());
  if (fd < 0) return fail(ESCOIN_ENOMEM, "jit: cannot create a temporary file in " + path);
  const size_t bytes = code.size() * 4;
  size_t done = 0;
  while (done < bytes) {
    const ssize_t n = write(fd, reinterpret_cast<const char *>(code.data()) + done, bytes - done);
    if (n <= 0) break;
    done += (size_t)n;
  }
  close(fd);
  std::string src = kWrapper;
  src.replace(src.find("%BLOB%"), 6, pbuf.data());
  elf->clear();
  const std::string err = done == bytes ? assemble_and_link(src, elf) : std::string("short write to the temporary file");
  unlink(pbuf.data());
  if (!err.empty()) return fail(ESCOIN_EHIP, "jit: " + err);
  return ESCOIN_OK;
}

namespace {
constexpr size_t kTemplateCode = 4096;     // bytes of s_nop the template carries where the code goes
struct WrapTemplate {
  std::vector<char> elf;
  size_t code_off = 0;                     // file offset of the first code byte
  uint64_t code_addr = 0;                  // ... and its address
  bool ok = false;
};
// The template, parsed once: where escoin_jit_code lies in the file.
const WrapTemplate &wrap_template() {
  static WrapTemplate t;
  static std::once_flag once;
  std::call_once(once, []() {
    const std::vector<uint32_t> nops(kTemplateCode / 4, 0xBF800000u);
    if (jit_assemble(nops, &t.elf) != ESCOIN_OK || t.elf.size() < sizeof(Elf64_Ehdr)) return;
    const Elf64_Ehdr *eh = reinterpret_cast<const Elf64_Ehdr *>(t.elf.data());
    if (std::memcmp(eh->e_ident, ELFMAG, SELFMAG) != 0 || eh->e_ident[EI_CLASS] != ELFCLASS64 || eh->e_shentsize != sizeof(Elf64_Shdr) ||
        eh->e_phentsize != sizeof(Elf64_Phdr) || eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) > t.elf.size())
      return;
    const Elf64_Shdr *sh = reinterpret_cast<const Elf64_Shdr *>(t.elf.data() + eh->e_shoff);
    for (int i = 0; i < eh->e_shnum; ++i) {
      if (sh[i].sh_type != SHT_SYMTAB) continue;
      const Elf64_Shdr &str = sh[sh[i].sh_link];
      const Elf64_Sym *sym = reinterpret_cast<const Elf64_Sym *>(t.elf.data() + sh[i].sh_offset);
      for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Sym); ++k) {
        if (std::strcmp(t.elf.data() + str.sh_offset + sym[k].st_name, "escoin_jit_code") != 0) continue;
        const Elf64_Shdr &text = sh[sym[k].st_shndx];
        t.code_addr = sym[k].st_value;
        t.code_off = (size_t)(text.sh_offset + (sym[k].st_value - text.sh_addr));
        // the placeholder must be where the symbol says, whole, inside the section
        bool nop = t.code_off + kTemplateCode <= text.sh_offset + text.sh_size;
        for (size_t b = 0; nop && b < kTemplateCode; b += 4)
          nop = *reinterpret_cast<const uint32_t *>(t.elf.data() + t.code_off + b) == 0xBF800000u;
        t.ok = nop;
      }
    }
  });
  return t;
}
}  // namespace

int jit_wrap(const std::vector<uint32_t> &code, std::vector<char> *elf) {
  if (code.empty()) return fail(ESCOIN_EINVAL, "jit: empty program");
  const WrapTemplate &t = wrap_template();
  if (!t.ok) return fail(ESCOIN_EHIP, "jit: no code object template");
  const size_t bytes = code.size() * 4, padded = (bytes + 4095) / 4096 * 4096;
  const uint64_t delta = padded - kTemplateCode;
  const size_t cut = t.code_off + kTemplateCode;            // file offset everything from which moves
  const uint64_t cut_addr = t.code_addr + kTemplateCode;
  elf->resize(t.elf.size() + delta);
  char *o = elf->data();
  std::memcpy(o, t.elf.data(), t.code_off);
  std::memcpy(o + t.code_off, code.data(), bytes);
  for (size_t b = bytes; b < padded; b += 4) { const uint32_t nop = 0xBF800000u; std::memcpy(o + t.code_off + b, &nop, 4); }
  std::memcpy(o + t.code_off + padded, t.elf.data() + cut, t.elf.size() - cut);
  if (delta == 0) return ESCOIN_OK;
  Elf64_Ehdr *eh = reinterpret_cast<Elf64_Ehdr *>(o);
  if (eh->e_shoff >= cut) eh->e_shoff += delta;
  if (eh->e_phoff >= cut) eh->e_phoff += delta;
  Elf64_Phdr *ph = reinterpret_cast<Elf64_Phdr *>(o + eh->e_phoff);
  for (int i = 0; i < eh->e_phnum; ++i) {
    if (ph[i].p_offset >= cut) {
      ph[i].p_offset += delta; ph[i].p_vaddr += delta; ph[i].p_paddr += delta;
    } else if (ph[i].p_offset + ph[i].p_filesz >= cut) {     // the segment that holds the code
      ph[i].p_filesz += delta; ph[i].p_memsz += delta;
    }
  }
  Elf64_Shdr *sh = reinterpret_cast<Elf64_Shdr *>(o + eh->e_shoff);
  for (int i = 0; i < eh->e_shnum; ++i) {
    if (sh[i].sh_type == SHT_NULL) continue;
    if (sh[i].sh_offset >= cut) {
      sh[i].sh_offset += delta;
      if (sh[i].sh_addr >= cut_addr) sh[i].sh_addr += delta;
    } else if (sh[i].sh_offset + sh[i].sh_size >= cut && sh[i].sh_type != SHT_NOBITS) {
      sh[i].sh_size += delta;                                // .text
    }
  }
  for (int i = 0; i < eh->e_shnum; ++i) {
    if (sh[i].sh_type == SHT_SYMTAB || sh[i].sh_type == SHT_DYNSYM) {
      Elf64_Sym *sym = reinterpret_cast<Elf64_Sym *>(o + sh[i].sh_offset);
      for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Sym); ++k)
        if (sym[k].st_shndx != SHN_UNDEF && sym[k].st_shndx < SHN_LORESERVE && sym[k].st_value >= cut_addr) sym[k].st_value += delta;
    } else if (sh[i].sh_type == SHT_DYNAMIC) {
      Elf64_Dyn *dyn = reinterpret_cast<Elf64_Dyn *>(o + sh[i].sh_offset);
      for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Dyn) && dyn[k].d_tag != DT_NULL; ++k)
        switch (dyn[k].d_tag) {
          case DT_HASH: case DT_GNU_HASH: case DT_STRTAB: case DT_SYMTAB: case DT_RELA: case DT_REL: case DT_JMPREL: case DT_PLTGOT:
          case DT_INIT: case DT_FINI: case DT_INIT_ARRAY: case DT_FINI_ARRAY:
            if (dyn[k].d_un.d_ptr >= cut_addr) dyn[k].d_un.d_ptr += delta;
            break;
          default: break;
        }
    }
  }
  return ESCOIN_OK;
}

// code -> executable device memory (code_memory.h): staged through an ordinary device buffer, copied by a kernel
static int jit_load_direct(const uint32_t *code, size_t words, JitModule *out, hipStream_t stream) {
  const size_t bytes = words * 4, total = (bytes + 256 + 4095) / 4096 * 4096;   // >= 64 s_nop behind the code, like the wrapper
  void *exec = nullptr, *stage = nullptr;
  int rc = code_mem_alloc(total, &exec);
  if (rc != ESCOIN_OK) return rc;
  hipError_t e = hipMalloc(&stage, bytes);
  if (e == hipSuccess) e = hipMemcpyAsync(stage, code, bytes, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) rc = code_mem_fill(exec, stage, bytes, total, stream);
  if (e == hipSuccess && rc == ESCOIN_OK) e = hipStreamSynchronize(stream);
  if (stage) (void)hipFree(stage);
  if (e != hipSuccess || rc != ESCOIN_OK) {
    code_mem_free(exec);
    return rc != ESCOIN_OK ? rc : fail(ESCOIN_EHIP, std::string("jit: filling the code memory failed: ") + hipGetErrorString(e));
  }
  JitModule m;
  m.direct = exec;
  m.code_base = (unsigned long long)(uintptr_t)exec;
  m.code_bytes = bytes;
  *out = m;
  return ESCOIN_OK;
}

int jit_load(const uint32_t *code, size_t words, JitModule *out, hipStream_t stream, int loader) {
  if (!code || words == 0) return fail(ESCOIN_EINVAL, "jit: empty program");
  static const bool direct_on = (ESC_KNOB("JIT_DIRECT", 1) != 0);
  if (loader == 0 && direct_on) {
    if (jit_load_direct(code, words, out, stream) == ESCOIN_OK) return ESCOIN_OK;
    if (getenv("ESCOIN_VERBOSE")) fprintf(stderr, "[escoin] jit: no executable device memory (%s): the code object loader instead\n", escoin_last_error());
  }
  static const bool wrap = (ESC_KNOB("JIT_WRAP", 1) != 0);
  const std::vector<uint32_t> words_v(code, code + words);
  std::vector<char> elf;
  int rc = wrap ? jit_wrap(words_v, &elf) : ESCOIN_EHIP;
  if (rc == ESCOIN_OK) rc = jit_load_elf(elf, words * 4, out, stream);
  if (rc != ESCOIN_OK) {       // (no template, or a loader that does not take the grown one: the assembler's own)
    rc = jit_assemble(words_v, &elf);
    if (rc == ESCOIN_OK) rc = jit_load_elf(elf, words * 4, out, stream);
  }
  return rc;
}

void jit_unload(JitModule *m) {
  if (m && m->module) (void)hipModuleUnload(m->module);
  if (m && m->direct) {
    // (hipModuleUnload and hipFree wait for the device themselves; the runtime's allocator underneath does not, and a
    //  launch that still runs this code may be in flight on a stream the caller has not synchronised)
    (void)hipDeviceSynchronize();
    code_mem_free(m->direct);
  }
  if (m) *m = JitModule();
}

}  // namespace escoin
