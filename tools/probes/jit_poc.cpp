// jit_poc.cpp -- proof of concept for run-time generated code (not product code):
//   machine-code bytes -> tiny assembly wrapper (.incbin) -> comgr assemble + link -> code object ->
//   hipModuleLoadData -> a locator kernel reports the blob's device address -> a compiled HIP kernel
//   calls it with s_swappc_b64 and the blob returns with s_setpc_b64 s[30:31].
//   hipcc --offload-arch=gfx950 -O3 -o jit_poc jit_poc.cpp -lamd_comgr
#include <amd_comgr/amd_comgr.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CG(x) do { amd_comgr_status_t s_ = (x); if (s_ != AMD_COMGR_STATUS_SUCCESS) { const char *m_ = "?"; amd_comgr_status_string(s_, &m_); printf("%s: %s\n", #x, m_); exit(2); } } while (0)

static const char *kWrapper = R"(
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
	s_setpc_b64 s[30:31]
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

static std::vector<char> build_code_object(const std::string &blob_path) {
  std::string src = kWrapper;
  src.replace(src.find("%BLOB%"), 6, blob_path);
  amd_comgr_data_t d;
  CG(amd_comgr_create_data(AMD_COMGR_DATA_KIND_SOURCE, &d));
  CG(amd_comgr_set_data(d, src.size(), src.data()));
  CG(amd_comgr_set_data_name(d, "escoin_jit.s"));
  amd_comgr_data_set_t in, rel, exe;
  CG(amd_comgr_create_data_set(&in));
  CG(amd_comgr_create_data_set(&rel));
  CG(amd_comgr_create_data_set(&exe));
  CG(amd_comgr_data_set_add(in, d));
  amd_comgr_action_info_t info;
  CG(amd_comgr_create_action_info(&info));
  CG(amd_comgr_action_info_set_isa_name(info, "amdgcn-amd-amdhsa--gfx950"));
  amd_comgr_status_t s = amd_comgr_do_action(AMD_COMGR_ACTION_ASSEMBLE_SOURCE_TO_RELOCATABLE, info, in, rel);
  if (s != AMD_COMGR_STATUS_SUCCESS) {
    size_t n = 0;
    amd_comgr_action_data_count(rel, AMD_COMGR_DATA_KIND_LOG, &n);
    printf("assemble failed, %zu logs\n", n);
    for (size_t i = 0; i < n; ++i) {
      amd_comgr_data_t l; size_t sz = 0;
      amd_comgr_action_data_get_data(rel, AMD_COMGR_DATA_KIND_LOG, i, &l);
      amd_comgr_get_data(l, &sz, nullptr);
      std::vector<char> b(sz + 1, 0);
      amd_comgr_get_data(l, &sz, b.data());
      printf("%s\n", b.data());
    }
    exit(3);
  }
  CG(amd_comgr_do_action(AMD_COMGR_ACTION_LINK_RELOCATABLE_TO_EXECUTABLE, info, rel, exe));
  amd_comgr_data_t out;
  CG(amd_comgr_action_data_get_data(exe, AMD_COMGR_DATA_KIND_EXECUTABLE, 0, &out));
  size_t sz = 0;
  CG(amd_comgr_get_data(out, &sz, nullptr));
  std::vector<char> elf(sz);
  CG(amd_comgr_get_data(out, &sz, elf.data()));
  amd_comgr_release_data(out);
  amd_comgr_release_data(d);
  amd_comgr_destroy_data_set(in);
  amd_comgr_destroy_data_set(rel);
  amd_comgr_destroy_data_set(exe);
  amd_comgr_destroy_action_info(info);
  return elf;
}

__global__ void call_jit(unsigned long long target, float *out) {
  float r;
  asm volatile("v_mov_b32 v64, 1.0\n"
               "s_swappc_b64 s[30:31], %1\n"
               "v_mov_b32 %0, v64"
               : "=v"(r) : "s"(target) : "v64", "s30", "s31", "scc", "memory");
  out[threadIdx.x] = r;
}

int main(int argc, char **argv) {
  // blob: v_add_f32 v64, v64, v64 ; v_add_f32 v64, v64, v64   (1 -> 4), encodings from llvm-mc
  const unsigned blob[] = {0x02808140u, 0x02808140u};
  const std::string path = "/tmp/escoin_jit_poc.bin";
  FILE *f = fopen(path.c_str(), "wb");
  fwrite(blob, 1, sizeof(blob), f);
  fclose(f);
  std::vector<char> elf = build_code_object(path);
  printf("code object: %zu bytes\n", elf.size());
  if (argc > 1) {   // CPU-only check of the comgr half: write the ELF and stop
    FILE *o = fopen(argv[1], "wb"); fwrite(elf.data(), 1, elf.size(), o); fclose(o);
    return 0;
  }
  hipModule_t mod;
  CK(hipModuleLoadData(&mod, elf.data()));
  hipFunction_t loc;
  CK(hipModuleGetFunction(&loc, mod, "escoin_jit_locator"));
  unsigned long long *d_addr; float *d_out;
  CK(hipMalloc(&d_addr, 8));
  CK(hipMalloc(&d_out, 64 * 4));
  void *args[] = {&d_addr};
  CK(hipModuleLaunchKernel(loc, 1, 1, 1, 1, 1, 1, 0, 0, args, nullptr));
  unsigned long long addr = 0;
  CK(hipMemcpy(&addr, d_addr, 8, hipMemcpyDeviceToHost));
  printf("generated code lives at 0x%llx\n", addr);
  if (!addr) return 4;
  hipLaunchKernelGGL(call_jit, dim3(1), dim3(64), 0, 0, addr, d_out);
  float h[64];
  CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
  printf("v64 after the call: %g (expect 4)\n", h[0]);
  return h[0] == 4.f ? 0 : 5;
}
