#!/usr/bin/env python3
"""Generates probe_issue.hip: instruction-issue model probe for the sparse stream loop (gfx950).

Not product code.  Questions it answers (results: profiles/r02_probe_issue.txt):
  * does a SIMD co-issue one wave's SALU / LDS instructions with another wave's v_pk_fma_f32, and how
    does that change from 1 to 2 to 4 waves per SIMD;
  * what a "one LDS read per nonzero, static accumulators" loop and a "row group, accumulator
    through M0 written by v_readfirstlane" loop sustain;
  * whether a VALU write of M0 is picked up by the next indexed VALU instruction, and whether SDWA
    forms escape GPR indexing.

    python gen_probe_issue.py > probe_issue.hip
    hipcc --offload-arch=gfx950 -O3 -o probe_issue probe_issue.hip
"""
import sys

ACC0 = 80           # accumulators v[80:127]
NPAIR = 24


def pk(acc, w, sel, x, idx=False):
    return ("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]"
            % (acc, acc + 1, w, w + 1, x, x + 1, acc, acc + 1, sel, sel))


def pkrun(n, start=0, x0=36):
    """n independent pk_fma over the accumulator file, inputs v[x0..x0+15], weight v[68:69]."""
    return [pk(ACC0 + 2 * ((start + i) % NPAIR), 68, i & 1, x0 + 2 * (i % 8)) for i in range(n)]


def interleave(pks, others):
    """Spread `others` evenly between the pk instructions."""
    out = []
    if not others:
        return list(pks)
    step = len(pks) / float(len(others))
    nxt, j = step, 0
    for i, p in enumerate(pks):
        out.append(p)
        while j < len(others) and i + 1 >= nxt - 1e-9:
            out.append(others[j])
            j += 1
            nxt += step
    out += others[j:]
    return out


VARIANTS = []


def variant(name, npk, desc):
    def deco(fn):
        VARIANTS.append((name, npk, desc, fn))
        return fn
    return deco


@variant("pk16", 16, "16 pk_fma only")
def v_pk16():
    return pkrun(16)


for ns in (4, 8, 16):
    def mk(ns=ns):
        return interleave(pkrun(16), ["s_add_u32 s%d, s%d, 1" % (40 + (i & 3), 40 + (i & 3)) for i in range(ns)])
    VARIANTS.append(("pk16_s%d" % ns, 16, "16 pk_fma + %d s_add_u32" % ns, mk))

for nv in (4, 8):
    def mk(nv=nv):
        return interleave(pkrun(16), ["v_add_u32 v%d, 1, v%d" % (33 + (i & 1), 33 + (i & 1)) for i in range(nv)])
    VARIANTS.append(("pk16_v%d" % nv, 16, "16 pk_fma + %d v_add_u32" % nv, mk))

for nl in (4, 8):
    def mk(nl=nl):
        # reads land in v[52:67] (not read by the pk's), waited for at the top of the next iteration
        rd = ["ds_read_b128 v[%d:%d], v32 offset:%d" % (52 + 4 * (i & 3), 55 + 4 * (i & 3), 1024 * (i & 7)) for i in range(nl)]
        return ["s_waitcnt lgkmcnt(0)"] + interleave(pkrun(16), rd)
    VARIANTS.append(("pk16_l%d" % nl, 16, "16 pk_fma + %d ds_read_b128" % nl, mk))


@variant("pk16_s8_l4", 16, "16 pk_fma + 8 s_add + 4 ds_read_b128")
def v_mix():
    rd = ["ds_read_b128 v[%d:%d], v32 offset:%d" % (52 + 4 * (i & 3), 55 + 4 * (i & 3), 1024 * i) for i in range(4)]
    sa = ["s_add_u32 s%d, s%d, 1" % (40 + (i & 3), 40 + (i & 3)) for i in range(8)]
    oth = []
    for i in range(4):
        oth += [sa[2 * i], rd[i], sa[2 * i + 1]]
    return ["s_waitcnt lgkmcnt(0)"] + interleave(pkrun(16), oth)


def b_body(T, depth1=True):
    """'One LDS read per nonzero' loop, static accumulators, T quads per lane.  Two nonzeros per
    iteration; payload quad [w_a, off_a, w_b, off_b] broadcast from the staging area.  X buffers
    v[36:51] / v[52:67] (T <= 4)."""
    L = []
    X = [36, 52]
    P = [68, 72]
    for half in (0, 1):        # two payload quads per iteration = 4 nonzeros
        pcur, pnxt = P[half], P[1 - half]
        for k in (0, 1):       # nonzero k of this payload quad; its data is in X[k]
            xcur, xnxt = X[k], X[1 - k]
            L.append("s_waitcnt lgkmcnt(0)")
            # address of the next nonzero: its offset sits in the payload (current quad .w, or next quad .y)
            offreg = pcur + 3 if k == 0 else pnxt + 1
            L.append("v_add_u32 v33, v32, v%d" % offreg)
            for t in range(T):
                L.append("ds_read_b128 v[%d:%d], v33 offset:%d" % (xnxt + 4 * t, xnxt + 4 * t + 3, 1024 * t))
            if k == 0:
                L.append("ds_read_b128 v[%d:%d], v35 offset:%d" % (pnxt, pnxt + 3, 16 * (half + 1)))
            w, sel = (pcur, 0) if k == 0 else (pcur + 2, 0)
            for t in range(T):
                for h in (0, 2):
                    acc = ACC0 + 2 * ((2 * t + h // 2 + 8 * k) % NPAIR)
                    L.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]"
                             % (acc, acc + 1, w, w + 1, xcur + 4 * t + h, xcur + 4 * t + h + 1, acc, acc + 1, sel, sel))
    L.append("s_add_u32 s40, s40, -1")
    return L


for T in (2, 3, 4):
    VARIANTS.append(("bbody_T%d" % T, 4 * 2 * T, "static acc, 1 LDS read/nonzero, T=%d (4 nonzeros/iter)" % T,
                     (lambda T=T: b_body(T))))


def a_body(n, sdwa=False):
    """Row-group loop, T=2, accumulator index through M0 written by v_readfirstlane.  Two groups
    per iteration (X double buffer A: v[36:39]/v[44:47], B: v[40:43]/v[48:51]); payload quads hold
    two records [m0_a, w_a, m0_b, w_b]; bits 16..31 of m0_a carry the row offset of the next group."""
    L = []
    XA, XB = [36, 44], [40, 48]
    nq = (n + 1) // 2
    PQ = [[68, 72], [52, 56]]            # payload quads of phase 0 / 1 (up to 2 quads per group)
    for p in (0, 1):
        if sdwa:
            L.append("v_add_u32_sdwa v33, s42, v32 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
        else:
            L.append("s_mov_b32 m0, s44")
            L.append("v_add_u32 v33, s42, v32")
        L.append("ds_read_b128 v[%d:%d], v33" % (XA[1 - p], XA[1 - p] + 3))
        L.append("ds_read_b128 v[%d:%d], v33 offset:1024" % (XB[1 - p], XB[1 - p] + 3))
        for q in range(nq):
            L.append("ds_read_b128 v[%d:%d], v34 offset:%d" % (PQ[1 - p][q], PQ[1 - p][q] + 3, 16 * (p * nq + q)))
        L.append("s_waitcnt lgkmcnt(%d)" % (2 + nq))
        for r in range(n):
            q, h = r // 2, r % 2
            preg = PQ[p][q] + 2 * h
            L.append(".long 0x%08x" % (0x7EF80500 + preg))   # v_readfirstlane_b32 m0, v<preg> (the assembler refuses M0 as its destination)
            if r == 0:
                pass
            for (acc, x) in ((ACC0, XA[p]), (ACC0 + 2, XA[p] + 2), (ACC0 + 4, XB[p]), (ACC0 + 6, XB[p] + 2)):
                # value = high half of the pair (m0 word, value)
                L.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                         % (acc, acc + 1, preg, preg + 1, x, x + 1, acc, acc + 1))
            if r == 0:
                L.append("s_lshr_b32 s42, m0, 16")
    L.append("s_add_u32 s40, s40, -1")
    return L


for n in (2, 3, 4):
    VARIANTS.append(("abody_n%d" % n, 2 * 4 * n, "row groups of %d records, M0 via v_readfirstlane, idx0 via s_mov m0" % n,
                     (lambda n=n: a_body(n))))
VARIANTS.append(("abody_n2_sdwa", 16, "same, n=2, address add as SDWA (no idx0)", (lambda: a_body(2, True))))


def idx_body(kind, per):
    """16 pk_fma, the accumulator index (M0) changed every `per` of them: what one index switch
    costs.  kind: "set" = s_set_gpr_idx_idx, "mov" = s_mov_b32 m0, "movnop" = s_mov_b32 m0 + s_nop 0,
    "shift" = s_lshr_b32 m0 (shift and set in one SALU instruction), "none" = no switch."""
    L = []
    pks = pkrun(16)
    for i, pkl in enumerate(pks):
        if i % per == 0 and kind != "none":
            sreg = 45 + (i // per) % 4          # s45..s48 hold 0xC000 | {0, 4, 8, 12}
            if kind == "set":
                L.append("s_set_gpr_idx_idx s%d" % sreg)
            elif kind == "mov":
                L.append("s_mov_b32 m0, s%d" % sreg)
            elif kind == "movnop":
                L.append("s_mov_b32 m0, s%d" % sreg)
                L.append("s_nop 0")
            elif kind == "shift":
                L.append("s_lshr_b32 m0, s49, %d" % (16 * ((i // per) % 2)))   # s49 = 0xC004C000
        L.append(pkl)
    return L


for kind in ("none", "set", "mov", "movnop", "shift"):
    for per in (2, 4, 8):
        if kind == "none" and per != 4:
            continue
        VARIANTS.append(("idx_%s_%d" % (kind, per), 16, "16 pk_fma, M0 switched every %d by %s" % (per, kind),
                         (lambda kind=kind, per=per: idx_body(kind, per))))


def kernel(name, lines, unroll):
    body = "\n".join('      "%s\\n"' % ln for ln in lines * unroll)
    return """
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) k_%s(float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  // payload area A (bytes 0..16383): [m0 word, 0.5] pairs; area B (16384..32767): [0.5, offset 0] pairs
  for (int i = threadIdx.x; i < 8192; i += blockDim.x)
    lds[i] = i < 4096 ? ((i & 1) ? 0x3f000000u : 0x0000C000u) : ((i & 1) ? 0u : 0x3f000000u);
  __syncthreads();
  const unsigned lane16 = (threadIdx.x & 63) * 16;
  SETUP(lane16);
  for (int it = 0; it < iters; ++it) {
    asm volatile(
%s
      ::: CLOB);
  }
  SINK(out);
}
""" % (name, body)


HEADER = r"""// GENERATED by gen_probe_issue.py -- hardware probe, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define CLOB "memory", "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67", \
  "v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85", \
  "v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102", \
  "v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117", \
  "v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
// X buffers 0.0 (LDS is read as floats of tiny magnitude), payload regs: (m0 word 0xC000, 0.5)
#define SETUP(l16) asm volatile( \
  "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x4000\n" \
  "v_mov_b32 v68, 0xC000\n v_mov_b32 v69, 0.5\n v_mov_b32 v70, 0xC000\n v_mov_b32 v71, 0.5\n" \
  "v_mov_b32 v72, 0xC000\n v_mov_b32 v73, 0.5\n v_mov_b32 v74, 0xC000\n v_mov_b32 v75, 0.5\n" \
  "v_mov_b32 v52, 0xC000\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 0xC000\n v_mov_b32 v55, 0.5\n" \
  "v_mov_b32 v56, 0xC000\n v_mov_b32 v57, 0.5\n v_mov_b32 v58, 0xC000\n v_mov_b32 v59, 0.5\n" \
  "s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_mov_b32 s44, 0xC000\n" \
  "s_mov_b32 s45, 0xC000\n s_mov_b32 s46, 0xC004\n s_mov_b32 s47, 0xC008\n s_mov_b32 s48, 0xC00C\n s_mov_b32 s49, 0xC004C000\n" \
  "s_set_gpr_idx_on s44, gpr_idx(SRC2,DST)\n" :: "v"(l16) : CLOB)
#define SINK(out) do { float r0; asm volatile("s_set_gpr_idx_off\n v_add_f32 %0, v80, v82" : "=v"(r0) :: "v80", "v82"); \
  if (r0 == 12345.678f) out[threadIdx.x] = r0; } while (0)
"""

MAIN_HEAD = r"""
struct Var { const char *name; const char *desc; void (*fn)(float *, int); int npk; int unroll; };

// ---- correctness: M0 written by VALU, consumed by the next indexed v_pk_fma_f32; SDWA vs indexing ----
__global__ void __attribute__((amdgpu_num_vgpr(32))) k_sem(float *out, int nops) {
  const int lane = threadIdx.x;
  float r[8], a33 = 0.f, a35 = 0.f;
  // v70 = M0 word selecting accumulator offset 4 (mode SRC2|DST); acc v[80:87] zero; x = 1,2
  asm volatile(
      "v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n"
      "v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n"
      "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 2.0\n v_mov_b32 v70, 0xC004\n v_mov_b32 v71, 3.0\n"
      "v_mov_b32 v72, 0xC002\n"
      "v_mov_b32 v32, 100\n v_mov_b32 v33, 0\n v_mov_b32 v35, 0\n"
      "s_mov_b32 s44, 0xC000\n s_mov_b32 s42, 7\n"
      "s_set_gpr_idx_on s44, gpr_idx(SRC2,DST)\n"
      "s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n"
      ".long 0x7EF80546\n s_nop 4\n"
      "v_pk_fma_f32 v[80:81], v[70:71], v[40:41], v[80:81] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
      "s_branch 2f\n"
      "1:\n"
      ".long 0x7EF80546\n"
      "v_pk_fma_f32 v[80:81], v[70:71], v[40:41], v[80:81] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
      "2:\n"
      // plain SALU write of M0 (index 2), consumed by the very next VALU instruction: v82/v83 += (3, 6)
      "s_mov_b32 s45, 0xC002\n s_mov_b32 m0, s45\n"
      "v_pk_fma_f32 v[80:81], v[70:71], v[40:41], v[80:81] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
      "s_mov_b32 s45, 0xC006C000\n s_lshr_b32 m0, s45, 16\n"
      "v_pk_fma_f32 v[80:81], v[70:71], v[40:41], v[80:81] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
      // SDWA add while idx = 4 (variant 0) .. then idx 2 via v72: does the dst move?
      ".long 0x7EF80548\n s_nop 4\n"
      "v_add_u32_sdwa v33, s42, v32 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
      "s_set_gpr_idx_off\n"
      "v_mov_b32 %0, v33\n v_mov_b32 %1, v35\n"
      : "=v"(a33), "=v"(a35) : "s"(nops) : CLOB);
  asm volatile(
      "v_mov_b32 %0, v80\n v_mov_b32 %1, v81\n v_mov_b32 %2, v82\n v_mov_b32 %3, v83\n"
      "v_mov_b32 %4, v84\n v_mov_b32 %5, v85\n v_mov_b32 %6, v86\n v_mov_b32 %7, v87\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) :: CLOB);
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) out[i] = r[i];
    out[8] = a33; out[9] = a35;
  }
}

static double time_kernel(void (*fn)(float *, int), int wg_per_cu, int iters, float *dout) {
  // 256-thread workgroups (one wave per SIMD each); LDS sized so that exactly wg_per_cu fit a CU
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu) / 1024 * 1024 - (wg_per_cu > 1 ? 1024 : 0);
  hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(256 * wg_per_cu), dim3(256), lds, 0, dout, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(256 * wg_per_cu), dim3(256), lds, 0, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char **argv) {
  float *dout; CK(hipMalloc(&dout, 1 << 20));
  for (int nops = 0; nops < 2; ++nops) {
    CK(hipMemset(dout, 0, 64));
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dout, nops ? 0 : 1);
    float h[10]; CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    printf("sem (%s): acc v80..87 =", nops ? "no nop after v_readfirstlane m0" : "s_nop 4 after v_readfirstlane m0");
    for (int i = 0; i < 8; ++i) printf(" %g", h[i]);
    printf("   [v_readfirstlane->m0: expect v84=3 v85=6; s_mov m0: v82=3 v83=6; s_lshr m0: v86=3 v87=6]   sdwa add: v33=%g v35=%g [v33=107: SDWA ignores indexing]\n", h[8], h[9]);
  }
  const int iters = 4000;
"""


def main():
    out = sys.stdout
    out.write(HEADER)
    table = []
    for (name, npk, desc, fn) in VARIANTS:
        lines = fn()
        total = len(lines)
        unroll = max(1, 64 // max(1, total))
        out.write(kernel(name, lines, unroll))
        table.append((name, desc, npk, unroll, total))
    out.write(MAIN_HEAD)
    out.write("  Var vars[] = {\n")
    for (name, desc, npk, unroll, total) in table:
        out.write('    {"%s", "%s [%d instr/iter]", k_%s, %d, %d},\n' % (name, desc, total, name, npk, unroll))
    out.write("  };\n")
    out.write(r"""
  printf("%-16s %6s %10s %10s %12s  %s\n", "variant", "w/SIMD", "TFLOP/s", "ns/iter", "cyc/iter@2.4", "what");
  for (const Var &v : vars) {
    if (argc > 1 && !strstr(v.name, argv[1])) continue;
    for (int w : {1, 2, 4}) {
      const double ms = time_kernel(v.fn, w, iters, dout);
      const double n_it = (double)iters * v.unroll;
      const double flops = 2.0 * 128.0 * v.npk * n_it * 4.0 * 256.0 * w;
      const double ns = ms * 1e6 / n_it;
      printf("%-16s %6d %10.1f %10.1f %12.0f  %s\n", v.name, w, flops / (ms * 1e-3) / 1e12, ns, ns * 2.4, v.desc);
    }
  }
  return 0;
}
""")


if __name__ == "__main__":
    main()
