#!/usr/bin/env python3
"""Generates stream_loop_asm.inc: the gfx950 inline-asm pieces of the tiled kernel
(sconv_tiled.hip; stream format in stream_builder.h).

Design notes (all measured on MI355X, tools/probes/):
  * fp32 FMA throughput needs v_pk_fma_f32 (plain v_fma_f32 tops out near 57 TFLOP/s, packed
    near 125) and >= 2 waves per SIMD;
  * the accumulator a nonzero updates is data dependent -> GPR-index mode (M0) with VDST and
    VSRC2 relative; works with VOP3P;
  * at 2 waves/SIMD every instruction a wave issues costs it an issue slot, scalar ones included
    (s_nop is the exception that misled an earlier version): the loop is built to need as few
    instructions per group as the data flow allows;
  * scalar loads cannot stream the weights (about two 64-byte lines in flight per wave) and
    v_readlane broadcasts out of a lane-distributed VGPR window cost 7.5 ns each: the unit's body
    is staged in LDS with the planes (LDS-DMA, one block ahead); a group's quad(s)
    [meta, v0, v1, v2] ([meta2, v3, v4, v5]) arrive as broadcast ds_read_b128 one group ahead and
    the values feed v_pk_fma_f32 as VGPR pairs (op_sel picks the half): no vector-memory wait
    inside the loop, one v_readfirstlane (meta) per group;
  * a lane owns two pixel quads (tiles A and B) so one record = 4 v_pk_fma_f32.

Register contract with sconv_tiled.hip (C++ compiled with amdgpu_num_vgpr(NV): the compiler
never touches v[NV..255]):
    v32,v33   LDS addresses of the input quads of the next group / the one after it (tile A;
              tile B = tile A + 1 KiB)               v34  payload pointer (trails by LAG bytes)
    v[36:51]  input quads: phase p -> A: v[36+8p..], B: v[40+8p..]
    v[52:55], v[56:59]  first payload quad, two phases            v[60:63] second payload quad
    v[64:159] tile-A accumulators, v[160:255] tile-B accumulators (same index + 96)
    s[32:45]  scratch owned by the asm
Operands: %[h0] lead word of the unit, %[h1]..%[h6] = END_6..END_1 (stream_builder.h),
%[lbA] the lane's LDS byte address of its tile-A quad (plane row 0, channel 0; %[lbB] is unused),
%[sbase] LDS byte address of the first quad in the wave's staging area.  The *_DMA copies of the
loop (escoin_sconv_tiled_dma_kernel) also take the plane-DMA operands of dma_site().
Groups run four / two to a counter update (body2), priorities per half of the workgroup
(PRIO_*), tile epilogues as single asm blocks (ESC_EPI3S / EPI5S / EPI1S, main()).

    python gen_stream_loop.py > stream_loop_asm.inc
"""
import sys

NV = 32
VA, VA2, VP = 32, 33, 34
XA = [36, 44]
XB = [40, 48]
P0 = [52, 56]
P1 = 60
ACC_A, ACC_B = 64, 160
NACC_TILE = 96
CNT = 32                     # groups left in the current bucket after the current one
META_P = [33, 34]            # meta of the current group, by phase: the NEXT group reads its
                             # record-0 accumulator straight out of bits 0..7
META2 = 35
HDR2 = 36                    # row offset / 32 of the group after the current one
IXT = [37, 38]               # alternating accumulator-index temporaries
END0 = 38                    # END_n in s[END0 + n], n = 1..6; s[END0 + 7] = 0
SGPR_LAST = 45
MAX_SLOTS2 = 6
ABL = set()                  # timing-only ablations (see main())
PRIO_HI = 1                  # priority of a wave's even groups (odd groups run at 0)
PRIO_QUADS = True            # four-group runs: one priority switch per two groups
STORE_MOD = ""               # ESC_GEN_STORE_MOD: modifier on the epilogues' stores (experiments: " nt", " sc1")
PRIO_BASE = 0                # added to both (the second-dispatched half of the workgroup: constant level 1, see main())


def bfe(dst, src, off, width):
    return "s_bfe_u32 s%d, s%d, 0x%x" % (dst, src, (width << 16) | off)


def pk4v(L, r, p):
    """Record r of the current group: value = half of a payload VGPR pair."""
    if "nopk" in ABL:
        return
    base = P0[p] if r < 3 else P1
    q = r % 3
    pair, sel = (base, 1) if q == 0 else ((base + 2, 0) if q == 1 else (base + 2, 1))
    for (acc, x) in ((ACC_A, XA[p]), (ACC_A + 2, XA[p] + 2), (ACC_B, XB[p]), (ACC_B + 2, XB[p] + 2)):
        L.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]"
                 % (acc, acc + 1, pair, pair + 1, x, x + 1, acc, acc + 1, sel, sel))


def xreads(L, p_next, areg):
    """Input quads of a group (tile A, tile B = tile A + 1 KiB) through the address in v[areg]."""
    if "noxp" in ABL:
        return
    L.append("ds_read_b128 v[%d:%d], v%d" % (XA[p_next], XA[p_next] + 3, areg))
    L.append("ds_read_b128 v[%d:%d], v%d offset:1024" % (XB[p_next], XB[p_next] + 3, areg))


def xaddr(L, areg):
    """LDS address of a group's tile-A quad from its row offset / 32 in s[HDR2].  Needs GPR index 0
    (a VALU instruction: its VGPR operands are relative like everybody's)."""
    if "noxp" not in ABL:
        L.append("v_lshl_add_u32 v%d, s%d, 5, %%[lbA]" % (areg, HDR2))


def pread(L, reg, off):
    if off:
        L.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (reg, reg + 3, VP, off))
    else:
        L.append("ds_read_b128 v[%d:%d], v%d" % (reg, reg + 3, VP))


LAG = 32     # v[VP] trails the payload address it stands for by two quads (see body2)
QUAD_MAX_N = 3   # buckets of up to this many records also run four groups per counter update


def body2(L, n, p, label, role):
    """Group k (phase p).  On entry: X(k), P0(k) were requested at the top of group k-1;
    s[HDR2] = row offset / 32 of group k+1, s[META_P[1-p]] bits 0..7 = accumulator of this group's
    record 0 (left there by group k-1, or by the prologue).

    Groups run four (A B C D) or two (P Q) to a counter update, phases 0 1 0 1.  Everything that
    needs GPR index 0 -- the address adds and the payload pointer's move -- sits in the phase-0
    bodies: A / C / P compute their successor's input address before their own records and the
    address of the group after it once their meta word is in, so B / D / Q switch the index only
    for their records.  Payload pointer:
      v[VP]  = payload address of group k - LAG            at the top of A, P, S (phase 0),
             = payload address of group k + stride - LAG   at the top of Q, T (phase 1)
    (stride: 32 bytes for groups of more than 3 records, 16 otherwise).  A moves it by four
    strides, P and S by two, once their own reads are out; B, C, D, Q read at offsets that
    compensate -- the lag keeps them non-negative.  Where the 32-byte groups end in phase 0, the
    bucket entry ESC2_E3_1 takes the difference off.  S (a bucket's odd last group) and T (a
    bucket entered in phase 1) run alone and compute only their own successor's address."""
    A = L.append
    stride = 32 if n > 3 else 16
    meta = META_P[p]
    assert p == (0 if role in "APCS" else 1)
    assert n <= QUAD_MAX_N or role in "PQST"
    A("%s_%%=:" % label)
    # The two waves of a SIMD are arbitrated oldest-first: left alone, the younger one runs ~25 %
    # slower all kernel long and every block waits for it.  Alternating priority by group parity
    # lets whichever wave is behind win its even groups.
    if "noprio" not in ABL and PRIO_HI != 0:
        if PRIO_QUADS and role in "ABCD":
            # four-group runs: two groups up, two groups down, one switch per two groups
            if role in "AC":
                A("s_setprio %d" % ((1 if role == "A" else 0) * PRIO_HI + PRIO_BASE))
        else:
            A("s_setprio %d" % ((1 - p) * PRIO_HI + PRIO_BASE))
    if p == 0:
        A("s_set_gpr_idx_idx 0")
        if n > 3:
            pread(L, P1, LAG + 16)
        # group k+1: a whole group of FMAs to land in
        xaddr(L, VA)
        xreads(L, 1, VA)
        pread(L, P0[1], LAG - stride if role == "C" else LAG + stride)
        if role != "C":
            A("v_add_u32 v%d, %d, v%d" % (VP, (4 if role == "A" else 2) * stride, VP))
    elif role == "T":
        A("s_set_gpr_idx_idx 0")
        if n > 3:
            pread(L, P1, LAG - stride + 16)
        xaddr(L, VA)
        xreads(L, 0, VA)
        pread(L, P0[0], LAG)
    else:
        if n > 3:
            pread(L, P1, LAG - stride + 16)
        xreads(L, 0, VA2)                   # address left by the phase-0 body
        pread(L, P0[0], LAG - 2 * stride if role == "B" else LAG)
    pair = role in "APC"
    if "noxp" in ABL:
        A("s_waitcnt lgkmcnt(%d)" % (2 if n > 3 else 1))
    else:
        A("s_waitcnt lgkmcnt(%d)" % (4 if n > 3 else 3))   # X(k), P0(k) landed
    # meta of THIS group and the accumulator indices of its records 1, 2 are extracted before
    # record 0's FMAs: the VALU -> SGPR -> SALU -> M0 chain then runs under those 4 packed FMAs
    # instead of between record 0 and record 1
    A("v_readfirstlane_b32 s%d, v%d" % (meta, P0[p]))
    A("s_lshr_b32 s%d, s%d, 21" % (HDR2, meta))          # row offset / 32 of group k+2
    if p == 0 and pair:
        xaddr(L, VA2)                                    # ... whose reads the phase-1 body issues
    for r in range(1, min(n, 3)):
        t = IXT[r % 2]
        if r == 2:
            A(bfe(t, meta, 14, 7))
        else:
            A("s_lshr_b32 s%d, s%d, %d" % (t, meta, 7 * r))
    A("s_set_gpr_idx_idx s%d" % META_P[1 - p])
    pk4v(L, 0, p)
    for r in range(1, n):
        t = IXT[r % 2]
        if r == 3:
            A("s_waitcnt lgkmcnt(%d)" % (1 if "noxp" in ABL else 3))
            A("v_readfirstlane_b32 s%d, v%d" % (META2, P1))
            for r2 in range(4, n):
                A("s_lshr_b32 s%d, s%d, %d" % (IXT[r2 % 2], META2, 7 * (r2 - 3)))
            A("s_set_gpr_idx_idx s%d" % META2)
        else:
            A("s_set_gpr_idx_idx s%d" % t)
        pk4v(L, r, p)


def generate2():
    L = []
    A = L.append
    A("s_waitcnt lgkmcnt(0)")
    A("s_cmp_eq_u32 %%[h%d], 0" % 6)                      # END_1 = number of groups
    A("s_cbranch_scc1 ESC2_X_%=")
    A("s_mov_b32 s%d, %%[h0]" % META_P[1])                # plays the meta of "group -1"
    A(bfe(HDR2, META_P[1], 8, 11))                       # group 0
    A("v_add_u32 v%d, %d, %%[sbase]" % (VP, -LAG))
    xaddr(L, VA)
    xreads(L, 0, VA)                                     # ... its reads fly under the bookkeeping
    pread(L, P0[0], LAG)
    for n in range(1, MAX_SLOTS2 + 1):
        A("s_mov_b32 s%d, %%[h%d]" % (END0 + n, 7 - n))
    A("s_mov_b32 s%d, 0" % (END0 + MAX_SLOTS2 + 1))
    A("s_lshr_b32 s%d, s%d, 21" % (HDR2, META_P[1]))     # group 1
    A("s_set_gpr_idx_on s%d, gpr_idx(SRC2,DST)" % META_P[1])
    if "noprio" not in ABL and PRIO_HI == 0:
        A("s_setprio %d" % PRIO_BASE)                    # a constant level: set once
    A("s_branch ESC2_E%d_0_%%=" % MAX_SLOTS2)
    for n in range(MAX_SLOTS2, 0, -1):
        # bucket n: groups [END_(n+1), END_n).  s[CNT] = groups of the bucket still to run (R).
        # Groups run in pairs (phase 0, phase 1) with ONE counter update and branch per pair; a
        # bucket entered in phase 1 runs one group alone first, an odd one out runs alone last
        # (and hands the next bucket phase 1).
        A("ESC2_E%d_0_%%=:" % n)
        dma_site(L, n, 0)
        A("s_sub_u32 s%d, s%d, s%d" % (CNT, END0 + n, END0 + n + 1))
        A("s_cmp_eq_u32 s%d, 0" % CNT)
        A("s_cbranch_scc1 ESC2_E%d_0_%%=" % (n - 1))
        A("ESC2_C%d_%%=:" % n)
        if n <= QUAD_MAX_N:
            A("s_sub_u32 s%d, s%d, 4" % (CNT, CNT))      # SCC = borrow: fewer than four groups left
            A("s_cbranch_scc1 ESC2_D%d_%%=" % n)
            body2(L, n, 0, "ESC2_QA%d" % n, "A")
            body2(L, n, 1, "ESC2_QB%d" % n, "B")
            body2(L, n, 0, "ESC2_QC%d" % n, "C")
            body2(L, n, 1, "ESC2_QD%d" % n, "D")
            A("s_sub_u32 s%d, s%d, 4" % (CNT, CNT))
            A("s_cbranch_scc0 ESC2_QA%d_%%=" % n)        # another four
            A("ESC2_D%d_%%=:" % n)
            A("s_add_u32 s%d, s%d, 4" % (CNT, CNT))      # 0..3 groups left
            A("s_cmp_eq_u32 s%d, 0" % CNT)
            A("s_cbranch_scc1 ESC2_E%d_0_%%=" % (n - 1))
        A("s_sub_u32 s%d, s%d, 2" % (CNT, CNT))          # SCC = borrow: exactly one group left
        A("s_cbranch_scc1 ESC2_S%d_%%=" % n)
        body2(L, n, 0, "ESC2_P%d" % n, "P")
        body2(L, n, 1, "ESC2_Q%d" % n, "Q")
        A("s_sub_u32 s%d, s%d, 2" % (CNT, CNT))
        A("s_cbranch_scc0 ESC2_P%d_%%=" % n)             # another whole pair
        A("s_cmp_eq_i32 s%d, -2" % CNT)                  # -2: the bucket is done; -1: one group left
        A("s_cbranch_scc1 ESC2_E%d_0_%%=" % (n - 1))
        body2(L, n, 0, "ESC2_S%d" % n, "S")
        A("s_branch ESC2_E%d_1_%%=" % (n - 1))
        A("ESC2_E%d_1_%%=:" % n)
        dma_site(L, n, 1)
        if n == 3:
            # the last 32-byte group ran in phase 0 and moved v[VP] by two of ITS strides: the
            # 16-byte groups' phase 1 expects one of theirs
            A("s_set_gpr_idx_idx 0")
            A("v_add_u32 v%d, -16, v%d" % (VP, VP))
        A("s_sub_u32 s%d, s%d, s%d" % (CNT, END0 + n, END0 + n + 1))
        A("s_cmp_eq_u32 s%d, 0" % CNT)
        A("s_cbranch_scc1 ESC2_E%d_1_%%=" % (n - 1))
        body2(L, n, 1, "ESC2_T%d" % n, "T")
        A("s_sub_u32 s%d, s%d, 1" % (CNT, CNT))
        A("s_cmp_eq_u32 s%d, 0" % CNT)
        A("s_cbranch_scc1 ESC2_E%d_0_%%=" % (n - 1))
        A("s_branch ESC2_C%d_%%=" % n)
    A("ESC2_E0_0_%=:")
    A("ESC2_E0_1_%=:")
    A("s_setprio 0")
    A("s_set_gpr_idx_off")
    A("ESC2_X_%=:")
    A("s_waitcnt lgkmcnt(0)")
    if ALIGN is not None:
        L, _ = align8(L, ALIGN)
    return L


DMA_SITES = (6, 5, 4, 3, 2)  # bucket entries that put plane DMA of the next block in flight
DMA_PER_SITE = 2
DMA_ON = False               # generate2() emits the sites (the *_DMA copies of the loop)


def dma_site(L, n, p):
    """At the entry of bucket n: up to DMA_PER_SITE of the LDS-DMA instructions this wave still owes
    the fill in flight (%[nch] of them: LDS address %[dst], channel as the scalar offset %[soff],
    the lane's table entry %[tv]; both move on by this wave's stride).  Issued in a burst after
    the barrier an LDS-DMA instruction holds its wave for ~100 cycles (issue is throttled by
    completion); one or two at a time among the FMAs cost 5-20 cycles each
    (tools/probes/probe_ldsdma.hip).  The sites fall at about 0, 8, 21, 42 and 69 % of the walk at
    the benchmarked densities.  M0 is the LDS address of a DMA instruction: it is put back to
    "index 0, SRC2 | DST relative" afterwards (every body sets the index before its first indexed
    instruction; no VALU instruction executes in between).  Kept to the bare instructions, and to
    a kernel of its own (escoin_sconv_tiled_dma_kernel): versions issuing whole channel planes of
    1-4 pieces under lane masks, or switching index mode off and on around the DMA, lost more in
    the sites than they won, and the sites' operands and checks cost a layer that cannot use them
    1.5-3 %."""
    if n not in DMA_SITES or not DMA_ON:
        return
    A = L.append
    A("s_cmp_eq_u32 %[nch], 0")
    A("s_cbranch_scc1 ESC2_NS%d_%d_%%=" % (n, p))
    for k in range(DMA_PER_SITE):
        if k:
            A("s_cmp_eq_u32 %[nch], 0")
            A("s_cbranch_scc1 ESC2_NR%d_%d_%%=" % (n, p))
        A("s_mov_b32 m0, %[dst]")
        A("s_add_u32 %[dst], %[dst], %[dstep]")
        A("s_sub_u32 %[nch], %[nch], 1")
        A("buffer_load_dwordx4 %[tv], %[rsrc], %[soff] offen lds")
        A("s_add_u32 %[soff], %[soff], %[sstep]")
    A("ESC2_NR%d_%d_%%=:" % (n, p))
    A("s_mov_b32 m0, 0xc000")
    A("ESC2_NS%d_%d_%%=:" % (n, p))


LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
# dummy registers for the inline-asm operands, only to let the assembler tell instruction sizes
_SIZE_OPERANDS = {"h0": "s8", "h1": "s9", "h2": "s10", "h3": "s11", "h4": "s12", "h5": "s13", "h6": "s14",
                  "lbA": "v1", "lbB": "v2", "sbase": "v3", "dst": "s16", "soff": "s17", "nch": "s18",
                  "rsrc": "s[20:23]", "tv": "v4", "dstep": "s24", "sstep": "s25"}
ALIGN = None                 # None: no alignment pass; 0 / 4: 8-byte instructions at 0 / 4 mod 8 (align8)


def insn_sizes(lines):
    """Encoded size in bytes of every line (0 for labels), from the assembler itself."""
    import re
    import subprocess
    txt = []
    for ln in lines:
        ln = ln.replace("%=", "0").replace("%%", "%")
        ln = re.sub(r"%\[(\w+)\]", lambda m: _SIZE_OPERANDS[m.group(1)], ln)
        txt.append(ln)
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input="\n".join(txt) + "\n",
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0 or "error" in r.stderr:
        raise RuntimeError("llvm-mc: " + r.stderr[:2000])
    enc = [len(m.split(",")) for m in re.findall(r"encoding: \[([^\]]*)\]", r.stdout)]
    sizes, k = [], 0
    for ln in txt:
        if ln.rstrip().endswith(":") or ln.lstrip().startswith("."):
            sizes.append(0)
        else:
            sizes.append(enc[k])
            k += 1
    assert k == len(enc), (k, len(enc))
    return sizes


def align8(lines, phase):
    """Code placement (MI355X_MICROARCH.md, two waves per SIMD, item 8): a hand-written stream can
    lose up to 13 % when its 8-byte instructions sit at 4 mod 8.  The block starts 8-byte aligned
    (.p2align 3) and an s_nop goes in front of every 8-byte instruction that would not start at
    `phase` mod 8.  Addresses are static, so one linear pass serves every path through the labels.
    phase 4 exists to MEASURE the sensitivity (same instruction count, opposite placement)."""
    sizes = insn_sizes(lines)
    out, off, pads = [".p2align 3"], 0, 0
    for ln, sz in zip(lines, sizes):
        if sz == 8 and off % 8 != phase:
            out.append("s_nop 0")
            off += 4
            pads += 1
        out.append(ln)
        off += sz
    return out, pads


def clobbers():
    c = ["memory", "scc", "m0"]
    c += ["s%d" % i for i in range(32, SGPR_LAST + 1)]
    c += ["v%d" % i for i in range(VA, 256)]
    return c


def emit_macro(out, name, lines):
    out.write("#define %s \\\n" % name)
    for ln in lines:
        out.write('  "%s\\n" \\\n' % ln)
    out.write('  ""\n')


def main():
    import os
    if os.environ.get("ESC_GEN_NOPRIO"):
        ABL.add("noprio")
    global PRIO_HI, PRIO_QUADS, ALIGN, STORE_MOD
    STORE_MOD = os.environ.get("ESC_GEN_STORE_MOD", "")
    if os.environ.get("ESC_GEN_ALIGN", "") != "":
        ALIGN = int(os.environ["ESC_GEN_ALIGN"])
    PRIO_QUADS = os.environ.get("ESC_GEN_PRIO_QUADS", "1") == "1"
    PRIO_HI = int(os.environ.get("ESC_GEN_PRIO_HI", PRIO_HI))
    out = sys.stdout
    out.write("// GENERATED by gen_stream_loop.py -- do not edit.\n")
    out.write("#define ESC_NV %d\n#define ESC_NACC_TILE %d\n" % (NV, NACC_TILE))
    global PRIO_BASE
    global DMA_ON
    for DMA_ON in (False, True):
        sfx = "_DMA" if DMA_ON else ""
        emit_macro(out, "ESC2_LOOP_ASM_BAND" + sfx, generate2())
        PRIO_BASE = int(os.environ.get("ESC_GEN_PRIO_YOUNG", "1"))
        old_hi = PRIO_HI
        PRIO_HI = int(os.environ.get("ESC_GEN_PRIO_YOUNG_HI", 0))
        emit_macro(out, "ESC2_LOOP_ASM_BAND_YOUNG" + sfx, generate2())
        PRIO_BASE = 0
        PRIO_HI = old_hi
    DMA_ON = False
    # timing-only ablations (wrong results), compiled in with -DESCOIN_ABLATIONS
    out.write("#ifdef ESCOIN_ABLATIONS\n")
    for name in ("nopk", "noxp"):
        ABL.add(name)
        emit_macro(out, "ESC2_LOOP_ASM_" + name.upper(), generate2())
        ABL.discard(name)
    out.write("#endif\n")
    out.write("#define ESC_STREAM_LOOP_CLOBBERS \\\n  ")
    out.write(", ".join('"%s"' % c for c in clobbers()))
    out.write("\n")
    zero = ["v_mov_b64 v[%d:%d], 0" % (i, i + 1) for i in range(ACC_A, 256, 2)]
    emit_macro(out, "ESC_ZERO_ACC_ASM", zero)
    out.write("#define ESC_ACC_CLOBBERS ")
    out.write(", ".join('"v%d"' % i for i in range(ACC_A, 256)))
    out.write("\n")
    # ESC_EPI3_<tile>_<g>: 3x3 / pad 1 epilogue of output channel g of a tile, in place on the
    # accumulators: out[e] = C[e] + L[e-1] + R[e+1] (L, C, R = the kc = 0, 1, 2 classes); the two
    # elements that live in the neighbouring quad come through DPP row_shr/row_shl with
    # bound_ctrl (0 at the row edge).  %4..%7 are per-lane 0/1 masks applied first when the
    # MASKED variant is used (columns >= W hold garbage when W is not a multiple of 4).
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for g in range(NACC_TILE // 12):
            L0 = base + 12 * g
            C0 = L0 + 4
            R0 = L0 + 8
            for masked in (False, True):
                lines = []
                if masked:
                    for cls in (L0, C0, R0):
                        for e in range(4):
                            lines.append("v_mul_f32 v%d, v%d, %%%d" % (cls + e, cls + e, 4 + e))
                lines += [
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 0, C0 + 0, R0 + 1),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 1, C0 + 1, L0 + 0),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 1, C0 + 1, R0 + 2),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 2, C0 + 2, L0 + 1),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 2, C0 + 2, R0 + 3),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 3, C0 + 3, L0 + 2),
                    "v_add_f32_dpp v%d, v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    % (C0 + 0, L0 + 3, C0 + 0),
                    "v_add_f32_dpp v%d, v%d, v%d row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    % (C0 + 3, R0 + 0, C0 + 3),
                    "v_mov_b32 %%0, v%d" % (C0 + 0), "v_mov_b32 %%1, v%d" % (C0 + 1),
                    "v_mov_b32 %%2, v%d" % (C0 + 2), "v_mov_b32 %%3, v%d" % (C0 + 3),
                ]
                emit_macro(out, "ESC_EPI3%s_%d_%d" % ("M" if masked else "", tile, g), lines)
    # ESC_EPI3S_<tile>_<r>: the whole 3x3 / pad 1 epilogue of a tile in one block, r = OW % 4: per
    # channel the shift-and-sum of the three classes in place, bias and ReLU when bit 0 / bit 1
    # of %[flags] say so, and the stores straight from the accumulator registers through an SGPR
    # base that moves on by one channel plane (%[ostr] bytes) -- no copies into compiler
    # registers, no per-channel address arithmetic on the vector side.  Lanes in %[ok] own a whole
    # output quad (one global_store_dwordx4; dword alignment is all a global store needs); for
    # r != 0 the lanes in %[okp] own the row's last, partial quad: they store r elements, and the
    # two class values of theirs that a STORED output takes from beyond the row (kc = 2 at column
    # OW for their own last output, kc = 0 at the quad's last column for the first output of the
    # row that follows in the lane order: neighbouring rows' data, not padding) are multiplied by
    # %[pm] (0 in those lanes, 1 elsewhere) first.  %[voff] = the lane's byte offset from %[base] (channel m0 of
    # the wave), %[bias] lane g = bias of channel m0 + g, %[gcount] = channels of this wave
    # (1..8).  s[32:37] scratch.
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for r in range(4):
            lines = ["s_mov_b64 s[32:33], exec", "s_mov_b64 s[34:35], %[base]"]
            ng = NACC_TILE // 12
            for g in range(ng):
                L0 = base + 12 * g
                C0 = L0 + 4
                R0 = L0 + 8
                if r:
                    lines.append("v_mul_f32 v%d, v%d, %%[pm]" % (R0 + r, R0 + r))
                    lines.append("v_mul_f32 v%d, v%d, %%[pm]" % (L0 + 3, L0 + 3))
                lines += [
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 0, C0 + 0, R0 + 1),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 1, C0 + 1, L0 + 0),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 1, C0 + 1, R0 + 2),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 2, C0 + 2, L0 + 1),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 2, C0 + 2, R0 + 3),
                    "v_add_f32 v%d, v%d, v%d" % (C0 + 3, C0 + 3, L0 + 2),
                    "v_add_f32_dpp v%d, v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    % (C0 + 0, L0 + 3, C0 + 0),
                    "v_add_f32_dpp v%d, v%d, v%d row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    % (C0 + 3, R0 + 0, C0 + 3),
                    "s_bitcmp1_b32 %[flags], 0",
                    "s_cbranch_scc0 ESC_EB%d_%d_%%=" % (tile, g),
                    "v_readlane_b32 s36, %%[bias], %d" % g,
                    "s_nop 1",
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0, C0 + 1, C0, C0 + 1),
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0 + 2, C0 + 3, C0 + 2, C0 + 3),
                    "ESC_EB%d_%d_%%=:" % (tile, g),
                    "s_bitcmp1_b32 %[flags], 1",
                    "s_cbranch_scc0 ESC_ER%d_%d_%%=" % (tile, g),
                ]
                lines += ["v_max_f32 v%d, 0, v%d" % (C0 + e, C0 + e) for e in range(4)]
                lines += [
                    "ESC_ER%d_%d_%%=:" % (tile, g),
                    # the channel in slot g (channels may have been re-dealt over the waves): its plane
                    "v_readlane_b32 s36, %%[chanv], %d" % g,
                    "s_nop 0",
                    "s_mul_i32 s36, s36, %[ostr]",
                    "s_add_u32 s38, s34, s36",
                    "s_addc_u32 s39, s35, 0",
                    "s_mov_b64 exec, %[ok]",
                    "global_store_dwordx4 %%[voff], v[%d:%d], s[38:39]%s" % (C0, C0 + 3, STORE_MOD),
                ]
                if r:
                    lines.append("s_mov_b64 exec, %[okp]")
                    if r == 1:
                        lines.append("global_store_dword %%[voff], v%d, s[38:39]" % C0)
                    else:
                        lines.append("global_store_dwordx%d %%[voff], v[%d:%d], s[38:39]" % (r, C0, C0 + r - 1))
                lines.append("s_mov_b64 exec, s[32:33]")
                if g + 1 < ng:
                    lines += [
                        "s_cmp_eq_u32 %%[gcount], %d" % (g + 1),
                        "s_cbranch_scc1 ESC_EX%d_%%=" % tile,
                    ]
            lines.append("ESC_EX%d_%%=:" % tile)
            emit_macro(out, "ESC_EPI3S_%d_%d" % (tile, r), lines)
    # ESC_EPI5S_<tile>_<r>: the same for 5x5 / pad 2 (five classes per channel, 4 channels per
    # wave): out[e] = sum over kc of class_kc[e + kc - 2], accumulated in place on class 2; positions
    # -2, -1 come from the left neighbour's elements 2, 3 and positions 4, 5 from the right
    # neighbour's 0, 1 through DPP.  In the lanes of a row's partial last quad (r != 0) every class
    # value at an element >= r is multiplied by %[pm] = 0 first (they hold a neighbouring row's
    # data, and stored outputs two columns away reach them).
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for r in range(4):
            lines = ["s_mov_b64 s[32:33], exec", "s_mov_b64 s[34:35], %[base]"]
            ng = NACC_TILE // 20
            for g in range(ng):
                cls = [base + 20 * g + 4 * kc for kc in range(5)]
                C0 = cls[2]
                if r:
                    for kc in (0, 1, 3, 4):
                        for e in range(r, 4):
                            lines.append("v_mul_f32 v%d, v%d, %%[pm]" % (cls[kc] + e, cls[kc] + e))
                for e in range(4):
                    for kc in (0, 1, 3, 4):
                        pos = e + kc - 2
                        if 0 <= pos <= 3:
                            lines.append("v_add_f32 v%d, v%d, v%d" % (C0 + e, C0 + e, cls[kc] + pos))
                for e in range(4):
                    for kc in (0, 1, 3, 4):
                        pos = e + kc - 2
                        if pos < 0:
                            lines.append("v_add_f32_dpp v%d, v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                         % (C0 + e, cls[kc] + pos + 4, C0 + e))
                        elif pos > 3:
                            lines.append("v_add_f32_dpp v%d, v%d, v%d row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                         % (C0 + e, cls[kc] + pos - 4, C0 + e))
                lines += [
                    "s_bitcmp1_b32 %[flags], 0",
                    "s_cbranch_scc0 ESC_FB%d_%d_%%=" % (tile, g),
                    "v_readlane_b32 s36, %%[bias], %d" % g,
                    "s_nop 1",
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0, C0 + 1, C0, C0 + 1),
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0 + 2, C0 + 3, C0 + 2, C0 + 3),
                    "ESC_FB%d_%d_%%=:" % (tile, g),
                    "s_bitcmp1_b32 %[flags], 1",
                    "s_cbranch_scc0 ESC_FR%d_%d_%%=" % (tile, g),
                ]
                lines += ["v_max_f32 v%d, 0, v%d" % (C0 + e, C0 + e) for e in range(4)]
                lines += [
                    "ESC_FR%d_%d_%%=:" % (tile, g),
                    "v_readlane_b32 s36, %%[chanv], %d" % g,
                    "s_nop 0",
                    "s_mul_i32 s36, s36, %[ostr]",
                    "s_add_u32 s38, s34, s36",
                    "s_addc_u32 s39, s35, 0",
                    "s_mov_b64 exec, %[ok]",
                    "global_store_dwordx4 %%[voff], v[%d:%d], s[38:39]%s" % (C0, C0 + 3, STORE_MOD),
                ]
                if r:
                    lines.append("s_mov_b64 exec, %[okp]")
                    if r == 1:
                        lines.append("global_store_dword %%[voff], v%d, s[38:39]" % C0)
                    else:
                        lines.append("global_store_dwordx%d %%[voff], v[%d:%d], s[38:39]" % (r, C0, C0 + r - 1))
                lines.append("s_mov_b64 exec, s[32:33]")
                if g + 1 < ng:
                    lines += [
                        "s_cmp_eq_u32 %%[gcount], %d" % (g + 1),
                        "s_cbranch_scc1 ESC_FX%d_%%=" % tile,
                    ]
            lines.append("ESC_FX%d_%%=:" % tile)
            emit_macro(out, "ESC_EPI5S_%d_%d" % (tile, r), lines)
    # ESC_EPI1S_<tile>: the same for pointwise layers (one class per channel, up to 24 channels per
    # wave, nothing to shift) whose output rows are whole quads
    # ESC_EPI1SN_<tile>: the same with non-temporal stores (plan option "stream_stores")
    for tile, base, mod, mname in ((0, ACC_A, STORE_MOD, "ESC_EPI1S"), (1, ACC_B, STORE_MOD, "ESC_EPI1S"),
                                   (0, ACC_A, " nt", "ESC_EPI1SN"), (1, ACC_B, " nt", "ESC_EPI1SN")):
        lines = ["s_mov_b64 s[32:33], exec", "s_mov_b64 s[34:35], %[base]", "s_mov_b64 exec, %[ok]"]
        ng = NACC_TILE // 4
        for g in range(ng):
            C0 = base + 4 * g
            lines += [
                "s_bitcmp1_b32 %[flags], 0",
                "s_cbranch_scc0 ESC_PB%d_%d_%%=" % (tile, g),
                "v_readlane_b32 s36, %%[bias], %d" % g,
                "s_nop 1",
                "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0, C0 + 1, C0, C0 + 1),
                "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0 + 2, C0 + 3, C0 + 2, C0 + 3),
                "ESC_PB%d_%d_%%=:" % (tile, g),
                "s_bitcmp1_b32 %[flags], 1",
                "s_cbranch_scc0 ESC_PR%d_%d_%%=" % (tile, g),
            ]
            lines += ["v_max_f32 v%d, 0, v%d" % (C0 + e, C0 + e) for e in range(4)]
            lines += [
                "ESC_PR%d_%d_%%=:" % (tile, g),
                "global_store_dwordx4 %%[voff], v[%d:%d], s[34:35]%s" % (C0, C0 + 3, mod),
            ]
            if g + 1 < ng:
                lines += [
                    "s_add_u32 s34, s34, %[ostr]",
                    "s_addc_u32 s35, s35, 0",
                    "s_cmp_eq_u32 %%[gcount], %d" % (g + 1),
                    "s_cbranch_scc1 ESC_PX%d_%%=" % tile,
                ]
        lines.append("ESC_PX%d_%%=:" % tile)
        lines.append("s_mov_b64 exec, s[32:33]")
        emit_macro(out, "%s_%d" % (mname, tile), lines)
    # ESC_EPI1SP_<tile>_<r>: pointwise layers whose output rows end in a PARTIAL quad of r = OW % 4 elements
    # (7 x 7 walked as 1 x 49: twelve quads and one element): whole quads go out as dwordx4 under %[ok], the
    # row's last quad as dword / dwordx2 / dwordx3 under %[okp].  (The C++ epilogue these layers took before
    # spends ~60 instructions per channel on address arithmetic and branches over the quad's width.)
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for r in (1, 2, 3):
            lines = ["s_mov_b64 s[32:33], exec", "s_mov_b64 s[34:35], %[base]"]
            ng = NACC_TILE // 4
            for g in range(ng):
                C0 = base + 4 * g
                lines += [
                    "s_bitcmp1_b32 %[flags], 0",
                    "s_cbranch_scc0 ESC_QB%d_%d_%d_%%=" % (tile, r, g),
                    "v_readlane_b32 s36, %%[bias], %d" % g,
                    "s_nop 1",
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0, C0 + 1, C0, C0 + 1),
                    "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0 + 2, C0 + 3, C0 + 2, C0 + 3),
                    "ESC_QB%d_%d_%d_%%=:" % (tile, r, g),
                    "s_bitcmp1_b32 %[flags], 1",
                    "s_cbranch_scc0 ESC_QR%d_%d_%d_%%=" % (tile, r, g),
                ]
                lines += ["v_max_f32 v%d, 0, v%d" % (C0 + e, C0 + e) for e in range(4)]
                part = {1: "global_store_dword %%[voff], v%d, s[34:35]" % C0,
                        2: "global_store_dwordx2 %%[voff], v[%d:%d], s[34:35]" % (C0, C0 + 1),
                        3: "global_store_dwordx3 %%[voff], v[%d:%d], s[34:35]" % (C0, C0 + 2)}[r]
                lines += [
                    "ESC_QR%d_%d_%d_%%=:" % (tile, r, g),
                    "s_mov_b64 exec, %[ok]",
                    "global_store_dwordx4 %%[voff], v[%d:%d], s[34:35]%s" % (C0, C0 + 3, STORE_MOD),
                    "s_mov_b64 exec, %[okp]",
                    part + STORE_MOD,
                    "s_mov_b64 exec, s[32:33]",
                ]
                if g + 1 < ng:
                    lines += [
                        "s_add_u32 s34, s34, %[ostr]",
                        "s_addc_u32 s35, s35, 0",
                        "s_cmp_eq_u32 %%[gcount], %d" % (g + 1),
                        "s_cbranch_scc1 ESC_QX%d_%d_%%=" % (tile, r),
                    ]
            lines.append("ESC_QX%d_%d_%%=:" % (tile, r))
            emit_macro(out, "ESC_EPI1SP_%d_%d" % (tile, r), lines)
    # ESC_EPI1SS_<tile>: strided pointwise layers (1x1, stride 2): a lane's quad holds four INPUT columns, elements
    # 0 and 2 of it are two adjacent outputs -- packed into one register pair and stored as a dwordx2 under %[ok];
    # a row's last quad may hold one output only (input width 14: columns 12, 13 and two of the next row): a dword
    # under %[okp]
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        lines = ["s_mov_b64 s[32:33], exec", "s_mov_b64 s[34:35], %[base]"]
        ng = NACC_TILE // 4
        for g in range(ng):
            C0 = base + 4 * g
            lines += [
                "v_mov_b32 v%d, v%d" % (C0 + 1, C0 + 2),
                "s_bitcmp1_b32 %[flags], 0",
                "s_cbranch_scc0 ESC_SB%d_%d_%%=" % (tile, g),
                "v_readlane_b32 s36, %%[bias], %d" % g,
                "s_nop 1",
                "v_pk_add_f32 v[%d:%d], v[%d:%d], s[36:37] op_sel_hi:[1,0]" % (C0, C0 + 1, C0, C0 + 1),
                "ESC_SB%d_%d_%%=:" % (tile, g),
                "s_bitcmp1_b32 %[flags], 1",
                "s_cbranch_scc0 ESC_SR%d_%d_%%=" % (tile, g),
                "v_max_f32 v%d, 0, v%d" % (C0, C0),
                "v_max_f32 v%d, 0, v%d" % (C0 + 1, C0 + 1),
                "ESC_SR%d_%d_%%=:" % (tile, g),
                "s_mov_b64 exec, %[ok]",
                "global_store_dwordx2 %%[voff], v[%d:%d], s[34:35]%s" % (C0, C0 + 1, STORE_MOD),
                "s_mov_b64 exec, %[okp]",
                "global_store_dword %%[voff], v%d, s[34:35]%s" % (C0, STORE_MOD),
                "s_mov_b64 exec, s[32:33]",
            ]
            if g + 1 < ng:
                lines += [
                    "s_add_u32 s34, s34, %[ostr]",
                    "s_addc_u32 s35, s35, 0",
                    "s_cmp_eq_u32 %%[gcount], %d" % (g + 1),
                    "s_cbranch_scc1 ESC_SX%d_%%=" % tile,
                ]
        lines.append("ESC_SX%d_%%=:" % tile)
        emit_macro(out, "ESC_EPI1SS_%d" % tile, lines)
    out.write("#define ESC_EPI3S_CLOBBERS \"memory\", \"scc\", \"s32\", \"s33\", \"s34\", \"s35\", \"s36\", \"s37\", \"s38\", \"s39\", ")
    out.write(", ".join('\"v%d\"' % i for i in range(ACC_A, 256)))
    out.write("\n")
    # ESC_READ_QUAD_<tile>_<q>: asm text moving accumulator quad q of a tile into %0..%3
    out.write("#define ESC_NQUADS_TILE %d\n" % (NACC_TILE // 4))
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for q in range(NACC_TILE // 4):
            r = base + 4 * q
            out.write('#define ESC_READ_QUAD_%d_%d "v_mov_b32 %%0, v%d\\n v_mov_b32 %%1, v%d\\n '
                      'v_mov_b32 %%2, v%d\\n v_mov_b32 %%3, v%d\\n"\n' % (tile, q, r, r + 1, r + 2, r + 3))


if __name__ == "__main__":
    main()
