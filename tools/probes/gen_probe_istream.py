#!/usr/bin/env python3
"""Generates probe_istream.hip: can a wave be fed STRAIGHT-LINE code that never repeats?

Not product code.  The question behind it: the stream walk of the tiled kernel spends more than
half of its instructions decoding the sparse structure at run time (meta word, row offset,
accumulator index through M0).  A kernel specialised per layer at WeightAlign ("the sparsity
pattern compiled into the instruction stream": static accumulator registers, static LDS offsets,
weights as literals) needs none of that -- but its code is megabytes per layer and every wave of a
workgroup runs a different part of it, so it cannot live in the 64 KiB instruction cache two CUs
share.  This probe runs exactly that instruction mix (per input row: 2 ds_read_b128 with immediate
offsets, per nonzero: s_mov_b32 literal + 4 v_pk_fma_f32 on statically chosen accumulators) as
straight-line code of 32 KiB .. 8 MiB per workgroup, each of the 8 waves looping over its own
eighth, on every CU, and reports the packed-FMA rate against the cache-resident case.

    python gen_probe_istream.py > probe_istream.hip
    hipcc --offload-arch=gfx950 -O3 -o probe_istream probe_istream.hip
"""
import random
import sys

ACC_A, ACC_B = 64, 160
SIZES_KB = [32, 2048]
NO_READS = NO_SMOV = NO_FMA = False     # ablations of the instruction mix (variants, see main)


def group(rng, n, phase):
    """One input row with n nonzeros, the way generated code would look."""
    xa, xb = (36, 40) if phase == 0 else (44, 48)
    off = rng.randrange(0, 56) * 1024 + rng.randrange(0, 4) * 256
    if NO_READS:
        return []
    L = ["ds_read_b128 v[%d:%d], v1 offset:%d" % (xa, xa + 3, off),
         "ds_read_b128 v[%d:%d], v1 offset:%d" % (xb, xb + 3, off + 1024)]
    return L


def records(rng, n, phase):
    xa, xb = (36, 40) if phase == 0 else (44, 48)
    L = []
    for r in range(n):
        a = 4 * rng.randrange(0, 24)
        if not NO_SMOV:
            L.append("s_mov_b32 s40, 0x%08x" % (0x3c000000 + rng.randrange(1, 1 << 20)))
        if NO_FMA:
            continue
        for (acc, x) in ((ACC_A + a, xa), (ACC_A + a + 2, xa + 2), (ACC_B + a, xb), (ACC_B + a + 2, xb + 2)):
            L.append("v_pk_fma_f32 v[%d:%d], s[40:41], v[%d:%d], v[%d:%d] op_sel_hi:[0,1,1]"
                     % (acc, acc + 1, x, x + 1, acc, acc + 1))
    return L


def block(seed, ngroups=16):
    """ngroups rows, software-pipelined one row ahead: reads of row k+1 before the FMAs of row k."""
    rng = random.Random(seed)
    ns = [rng.choice((1, 2, 2, 3, 3, 3, 4)) for _ in range(ngroups)]
    L, nrec = [], 0
    L += group(rng, ns[0], 0)
    for k in range(ngroups):
        if k + 1 < ngroups:
            L += group(rng, ns[k + 1], (k + 1) & 1)
            L.append("s_waitcnt lgkmcnt(%d)" % (0 if NO_READS else 2))
        else:
            L.append("s_waitcnt lgkmcnt(0)")
        L += records(rng, ns[k], k & 1)
        nrec += ns[k]
    return L, nrec


def size_of(lines):
    n = 0
    for ln in lines:
        n += 4 if ln.startswith("s_waitcnt") else 8
    return n


def kernel(size_kb, nseg=8, tag=""):
    blk, nrec = block(7)
    bsz = size_of(blk)
    seg_target = size_kb * 1024 // nseg
    tail = 24
    reps = max(1, (seg_target - tail) // bsz)
    seg_bytes = reps * bsz + tail
    body = "\\n".join(blk)
    asm = []
    A = asm.append
    A("s_getpc_b64 s[46:47]")
    A("ESCP_PC1_%=:")
    A("s_mul_i32 s42, %%[seg], 0x%x" % seg_bytes)
    A("s_add_u32 s46, s46, s42")
    A("s_addc_u32 s47, s47, 0")
    A("s_add_u32 s46, s46, ESCP_SEG0_%=-ESCP_PC1_%=")
    A("s_addc_u32 s47, s47, 0")
    A("s_getpc_b64 s[48:49]")
    A("ESCP_PC2_%=:")
    A("s_add_u32 s48, s48, ESCP_END_%=-ESCP_PC2_%=")
    A("s_addc_u32 s49, s49, 0")
    A("s_mov_b32 s44, %[iters]")
    A("s_setpc_b64 s[46:47]")
    for s in range(nseg):
        A("ESCP_SEG%d_%%=:" % s)
        A(".rept %d" % reps)
        asm.extend(blk)
        A(".endr")
        A("s_sub_u32 s44, s44, 1")
        A("s_cmp_eq_u32 s44, 0")
        A("s_cbranch_scc1 ESCP_DONE%d_%%=" % s)
        A("s_setpc_b64 s[46:47]")
        A("ESCP_DONE%d_%%=:" % s)
        A("s_setpc_b64 s[48:49]")
        A("s_nop 0")
    A("ESCP_END_%=:")
    A("s_waitcnt lgkmcnt(0)")
    text = "".join('      "%s\\n"\n' % ln for ln in asm)
    clob = ", ".join('"v%d"' % i for i in range(36, 256))
    src = """
__global__ void __launch_bounds__(512) k_is%s(float *out, int iters, int own, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)(i & 7) * 1e-3f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int seg = own ? wave : 0;
  const unsigned lb = (threadIdx.x & 63) * 16;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v1, %%[lb]\\n"
%s      :: [seg] "s"(seg), [iters] "s"(iters), [lb] "v"(lb)
      : "memory", "scc", "s40", "s41", "s42", "s44", "s46", "s47", "s48", "s49", "v1", %s);
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  float r;
  asm volatile("v_mov_b32 %%0, v64" : "=v"(r));
  if (r == 12345.f) out[threadIdx.x] = r;
}
""" % ("%d%s" % (size_kb, tag), text, clob)
    return src, reps * nrec, seg_bytes * nseg


HEADER = r"""// GENERATED by gen_probe_istream.py -- see there.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
"""


def main():
    out = sys.stdout
    out.write(HEADER)
    table = []
    global NO_READS, NO_SMOV, NO_FMA
    for tag, (NO_READS, NO_SMOV, NO_FMA) in (("", (False, False, False)), ("_noreads", (True, False, False)),
                                             ("_nosmov", (False, True, False)), ("_fmaonly", (True, True, False)),
                                             ("_nofma", (False, False, True))):
        for kb in SIZES_KB:
            src, recs_per_pass, code_bytes = kernel(kb, tag=tag)
            out.write(src)
            table.append(("%d%s" % (kb, tag), recs_per_pass, code_bytes))
    out.write(r"""
typedef void (*Kern)(float *, int, int, unsigned long long *);
struct Var { const char *kb; Kern fn; double recs; double code; };
int main(int argc, char **argv) {
  float *dout; CK(hipMalloc(&dout, 1 << 20));
  Var vars[] = {
""")
    for (kb, recs, code) in table:
        out.write('    {"%s", k_is%s, %d.0, %d.0},\n' % (kb, kb, recs, code))
    out.write(r"""  };
  unsigned long long *dclk; CK(hipMalloc(&dclk, 256 * 16));
  printf("%-16s %-6s %8s %12s %12s %14s %8s %12s\n", "code/WG KiB", "waves", "ms", "pkFMA TF/s", "ns/record", "ifetch GB/s/CU", "GHz", "cyc/record");
  for (const Var &v : vars) {
    CK(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int own = 0; own < 2; ++own) {
      // about the same number of records per wave in every configuration
      int iters = (int)(4.0e5 / v.recs); if (iters < 2) iters = 2;
      hipLaunchKernelGGL(v.fn, dim3(256), dim3(512), 65536, 0, dout, 2, own, dclk);
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(256), dim3(512), 65536, 0, dout, iters, own, dclk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double recs = v.recs * iters;                 // per wave
      const double flops = recs * 4 * 128 * 2 * 8 * 256;   // 4 pk_fma x 128 FMA, 8 waves, 256 WGs
      const double code_per_wave = v.code / 8 * iters;
      unsigned long long hc[512]; CK(hipMemcpy(hc, dclk, sizeof(hc), hipMemcpyDeviceToHost));
      double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
      printf("%-16s %-6s %8.3f %12.1f %12.2f %14.1f %8.3f %12.1f\n", v.kb, own ? "own" : "same", ms, flops / (ms * 1e-3) / 1e12,
             ms * 1e6 / recs, code_per_wave * (own ? 8 : 1) / (ms * 1e-3) / 1e9, cyc / (rt * 10.0), cyc / 256 / recs);
    }
  }
  return 0;
}
""")


if __name__ == "__main__":
    main()
