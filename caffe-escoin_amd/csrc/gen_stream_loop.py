#!/usr/bin/env python3
"""Generates stream_loop_asm.inc: the gfx950 inline-asm pieces of the tiled kernel
(sconv_tiled.hip; stream format in stream_builder.h).

Design notes (all measured on MI355X, tools/probes/):
  * fp32 FMA throughput needs v_pk_fma_f32 (plain v_fma_f32 tops out near 57 TFLOP/s, packed
    near 125) and >= 2 waves per SIMD;
  * the accumulator a nonzero updates is data dependent -> GPR-index mode (M0) with VDST and
    VSRC2 relative; works with VOP3P;
  * scalar loads cannot stream the weights (about two 64-byte lines in flight per wave, ~750
    cycles per line): the stream comes in through the VECTOR memory path instead, 64 chunks of
    48 bytes held lane-distributed in 12 VGPRs (lane l = chunk l mod 64), refilled half a window
    (32 lanes, EXEC-masked) at a time 32 chunks ahead of use, and single dwords are broadcast
    with v_readlane_b32;
  * a lane owns two pixel quads (tiles A and B) so one record = 4 v_pk_fma_f32.

Register contract with sconv_tiled.hip (C++ compiled with amdgpu_num_vgpr(NV): the compiler
never touches v[NV..255]):
    v32,v33          LDS addresses of the next group's quads (tile A, tile B)
    v[36:51]         input quads: phase p -> A: v[36+8p..], B: v[40+8p..]
    v[52:63]         stream window (dword d of the chunks in v[52+d])
    v[64:159]        tile-A accumulators, v[160:255] tile-B accumulators (same index + 96)
    s[32:59]         scratch owned by the asm (cursor, bucket ends, values, ...)
Operands: %[k] chunk cursor (in/out), %[plo]/%[phi] address of the next half window to request
(in/out), %[pend] 1 while a half-window request may be in flight (in/out; the kernel clears it
after a full vmcnt drain), %[lbA]/%[lbB] the lane's LDS byte addresses of its two quads (plane row 0, channel
0), %[voff] (lane & 31) * 48.

    python gen_stream_loop.py > stream_loop_asm.inc
"""
import sys

NV = 32
VA, VB = 32, 33
XA = [36, 44]
XB = [40, 48]
SW = 52                      # stream window registers
ACC_A, ACC_B = 64, 160
NACC_TILE = 96
CUR = [32, 33]               # chunk cursor, alternating by phase
STOP = 34
HDR = 35
IX, IX2 = 36, 37
VAL = [38, 40]               # SGPR pairs (value, junk)
END0 = 41                    # END_n in s[END0 + n], n = 1..8
UEND = 50
TMP = 51
PTR = 54                     # s[54:55]
EXS = 56                     # s[56:57] saved exec
IXT = [58, 59]               # alternating shifted-index temporaries
PEND = 52                    # 1: a half-window request may still be in flight
_label = [0]
MAX_SLOTS = 8
ABL = set()                  # generator switches: 'band' + timing-only ablations (see main())
HALF_BYTES = 32 * 48


def refill(L, cur):
    """Cursor `cur` (SGPR number) sits on a multiple of 32: wait for its half window and
    request the next one into the lanes of the half that was just consumed."""
    A = L.append
    # The half holding chunk `cur` was requested 32 chunks ago.  vmcnt completes in order, so
    # waiting for it also waits for everything issued since -- in particular the next block's
    # LDS-DMA, which the workgroup issues at every block start.  The kernel drains vmcnt at each
    # block start anyway and clears PEND there: only a request younger than that drain needs a wait.
    _label[0] += 1
    A("s_cmp_eq_u32 s%d, 0" % PEND)
    A("s_cbranch_scc1 ESC_RW%d_%%=" % _label[0])
    A("s_waitcnt vmcnt(0)")
    A("ESC_RW%d_%%=:" % _label[0])
    A("s_bitcmp1_b32 s%d, 5" % cur)
    A("s_cselect_b32 exec_lo, -1, 0")
    A("s_cselect_b32 exec_hi, 0, -1")
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d]" % (SW, SW + 3, PTR, PTR + 1))
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:16" % (SW + 4, SW + 7, PTR, PTR + 1))
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:32" % (SW + 8, SW + 11, PTR, PTR + 1))
    A("s_mov_b64 exec, s[%d:%d]" % (EXS, EXS + 1))
    A("s_add_u32 s%d, s%d, %d" % (PTR, PTR, HALF_BYTES))
    A("s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1))
    A("s_mov_b32 s%d, 1" % PEND)


def x_prefetch(L, cur, xset):
    """Issue the LDS reads of group `cur`'s two quads into X set `xset` (index 0: no relocation)."""
    A = L.append
    A("v_readlane_b32 s%d, v%d, s%d" % (HDR, SW, cur))
    if "noxp" in ABL:
        return
    A("s_set_gpr_idx_idx 0")
    A("v_add_u32 v%d, s%d, %%[lbA]" % (VA, HDR))
    if "band" in ABL:
        # band mode: tile B is the next 64-quad slab of the same plane, 1 KiB further
        A("ds_read_b128 v[%d:%d], v%d" % (XA[xset], XA[xset] + 3, VA))
        A("ds_read_b128 v[%d:%d], v%d offset:1024" % (XB[xset], XB[xset] + 3, VA))
        return
    A("v_add_u32 v%d, s%d, %%[lbB]" % (VB, HDR))
    A("ds_read_b128 v[%d:%d], v%d" % (XA[xset], XA[xset] + 3, VA))
    A("ds_read_b128 v[%d:%d], v%d" % (XB[xset], XB[xset] + 3, VB))


def pk4(L, val, p):
    A = L.append
    if "nopk" in ABL:
        return
    for (acc, x) in ((ACC_A, XA[p]), (ACC_A + 2, XA[p] + 2), (ACC_B, XB[p]), (ACC_B + 2, XB[p] + 2)):
        A("v_pk_fma_f32 v[%d:%d], s[%d:%d], v[%d:%d], v[%d:%d] op_sel_hi:[0,1,1]"
          % (acc, acc + 1, val, val + 1, x, x + 1, acc, acc + 1))


def entry(L, n, p):
    """Bucket entry / stop handler: bucket exhausted -> next bucket; window crossing -> refill and
    redo the X prefetch; then recompute the stop and fall into the loop."""
    A = L.append
    c = CUR[p]
    A("ESC_E%d_%d_%%=:" % (n, p))
    A("s_cmp_eq_u32 s%d, s%d" % (c, END0 + n))
    A("s_cbranch_scc1 ESC_E%d_%d_%%=" % (n - 1, p))
    A("s_and_b32 s%d, s%d, 31" % (TMP, c))
    A("s_cmp_lg_u32 s%d, 0" % TMP)
    A("s_cbranch_scc1 ESC_S%d_%d_%%=" % (n, p))
    refill(L, c)
    x_prefetch(L, c, p)
    A("ESC_S%d_%d_%%=:" % (n, p))
    A("s_or_b32 s%d, s%d, 31" % (TMP, c))
    A("s_add_u32 s%d, s%d, 1" % (TMP, TMP))
    A("s_min_u32 s%d, s%d, s%d" % (STOP, TMP, END0 + n))


def loop_body(L, n, p):
    A = L.append
    c, nx = CUR[p], CUR[1 - p]
    A("ESC_L%d_%d_%%=:" % (n, p))
    A("v_readlane_b32 s%d, v%d, s%d" % (IX, SW + 1, c))
    A("v_readlane_b32 s%d, v%d, s%d" % (VAL[0], SW + 3, c))
    A("s_add_u32 s%d, s%d, 1" % (nx, c))
    if n > 4:
        A("v_readlane_b32 s%d, v%d, s%d" % (IX2, SW + 2, c))
    if n > 1:
        A("v_readlane_b32 s%d, v%d, s%d" % (VAL[1], SW + 4, c))
    A("s_waitcnt lgkmcnt(0)")

    def ixreg(r):
        if r == 0:
            return IX
        if r == 4:
            return IX2
        return IXT[r % 2]

    for r in range(n):
        A("s_set_gpr_idx_idx s%d" % ixreg(r))
        if r + 1 < n and (r + 1) % 4 != 0:
            # next record's index byte, computed while this record's FMAs issue
            A("s_lshr_b32 s%d, s%d, %d" % (ixreg(r + 1), IX if r + 1 < 4 else IX2, 8 * ((r + 1) % 4)))
        pk4(L, VAL[r % 2], p)
        if r + 2 < n:
            A("v_readlane_b32 s%d, v%d, s%d" % (VAL[r % 2], SW + 3 + r + 2, c))
        if r == 0:
            x_prefetch(L, nx, 1 - p)
    if "nop4" in ABL:
        A("s_nop 0"); A("s_nop 0"); A("s_nop 0"); A("s_nop 0")
    if "vnop4" in ABL:
        A("v_nop"); A("v_nop"); A("v_nop"); A("v_nop")
    A("s_cmp_eq_u32 s%d, s%d" % (nx, STOP))
    A("s_cbranch_scc1 ESC_E%d_%d_%%=" % (n, 1 - p))
    if p == 1:
        A("s_branch ESC_L%d_0_%%=" % n)


def generate():
    L = []
    _label[0] = 0
    A = L.append
    c0 = CUR[0]
    A("s_mov_b64 s[%d:%d], exec" % (EXS, EXS + 1))
    A("s_mov_b32 s%d, %%[k]" % c0)
    A("s_mov_b32 s%d, %%[plo]" % PTR)
    A("s_mov_b32 s%d, %%[phi]" % (PTR + 1))
    A("s_mov_b32 s%d, %%[pend]" % PEND)
    # unit header chunk (may sit on a window crossing)
    A("s_and_b32 s%d, s%d, 31" % (TMP, c0))
    A("s_cmp_lg_u32 s%d, 0" % TMP)
    A("s_cbranch_scc1 ESC_H_%=")
    refill(L, c0)
    A("ESC_H_%=:")
    A("v_readlane_b32 s%d, v%d, s%d" % (UEND, SW, c0))
    for n in range(1, MAX_SLOTS + 1):
        A("v_readlane_b32 s%d, v%d, s%d" % (END0 + n, SW + n, c0))
    A("s_add_u32 s%d, s%d, 1" % (c0, c0))
    A("s_nop 3")                                   # SALU write -> v_readlane lane select
    A("s_set_gpr_idx_on s%d, gpr_idx(SRC2,DST)" % c0)   # index set before every use
    x_prefetch(L, c0, 0)
    for n in range(MAX_SLOTS, 0, -1):
        # the entries for both phases, then the two loop bodies
        entry(L, n, 0)
        A("s_branch ESC_L%d_0_%%=" % n)
        entry(L, n, 1)
        A("s_branch ESC_L%d_1_%%=" % n)
        loop_body(L, n, 0)
        loop_body(L, n, 1)
    A("ESC_E0_1_%=:")
    A("s_mov_b32 s%d, s%d" % (CUR[0], CUR[1]))
    A("ESC_E0_0_%=:")
    A("s_set_gpr_idx_off")
    A("s_waitcnt lgkmcnt(0)")
    A("s_mov_b32 %%[k], s%d" % CUR[0])
    A("s_mov_b32 %%[plo], s%d" % PTR)
    A("s_mov_b32 %%[phi], s%d" % (PTR + 1))
    A("s_mov_b32 %%[pend], s%d" % PEND)
    return L


def generate_init():
    """Requests half window 0 (chunks 0..31 of the wave's stream) into lanes 0..31."""
    L = []
    A = L.append
    A("s_mov_b64 s[%d:%d], exec" % (EXS, EXS + 1))
    A("s_mov_b32 s%d, %%[plo]" % PTR)
    A("s_mov_b32 s%d, %%[phi]" % (PTR + 1))
    A("s_mov_b32 exec_lo, -1")
    A("s_mov_b32 exec_hi, 0")
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d]" % (SW, SW + 3, PTR, PTR + 1))
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:16" % (SW + 4, SW + 7, PTR, PTR + 1))
    A("global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:32" % (SW + 8, SW + 11, PTR, PTR + 1))
    A("s_mov_b64 exec, s[%d:%d]" % (EXS, EXS + 1))
    A("s_add_u32 s%d, s%d, %d" % (PTR, PTR, HALF_BYTES))
    A("s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1))
    A("s_mov_b32 %%[plo], s%d" % PTR)
    A("s_mov_b32 %%[phi], s%d" % (PTR + 1))
    return L


# ------------------------------------------------------------------------------------------
# Stream format 2 (stream_builder.h): the unit's body is staged in LDS with the planes; a group's
# quad(s) [meta, v0, v1, v2] ([meta2, v3, v4, v5]) arrive as broadcast ds_read_b128 one group
# ahead and the values feed v_pk_fma_f32 as VGPR pairs (op_sel picks the half): no v_readlane,
# no vector-memory wait inside the loop.  Per group the vector ALU sees one v_readfirstlane
# (meta), the address adds and the FMAs.
#
#   v32,v33   LDS addresses of the next group's input quads      v34  LDS address of the
#   v[36:51]  input quads, two phases (as above)                       current group's payload
#   v[52:55], v[56:59]  first payload quad, two phases            v[60:63] second payload quad
# Operands: %[h0] lead meta of group 0, %[h1]..%[h6] = END_6..END_1, %[lbA]/%[lbB], %[sbase] LDS
# byte address of the wave's staging area.
# ------------------------------------------------------------------------------------------
P0 = [52, 56]
P1 = 60
VP = 34
META, META2, HDR2 = 34, 35, 36
IX0 = [38, 39]
MAX_SLOTS2 = 6


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


def prefetch2(L, p_next, stride, band):
    """LDS reads of the next group: its two input quads (row offset / 32 in s[HDR2]) and its first
    payload quad.  Runs with GPR index 0."""
    A = L.append
    if "noxp" in ABL:
        if stride:
            A("ds_read_b128 v[%d:%d], v%d offset:%d" % (P0[p_next], P0[p_next] + 3, VP, stride))
            A("v_add_u32 v%d, %d, v%d" % (VP, stride, VP))
        else:
            A("ds_read_b128 v[%d:%d], v%d" % (P0[p_next], P0[p_next] + 3, VP))
        return
    A("v_lshl_add_u32 v%d, s%d, 5, %%[lbA]" % (VA, HDR2))
    if band:
        A("ds_read_b128 v[%d:%d], v%d" % (XA[p_next], XA[p_next] + 3, VA))
        A("ds_read_b128 v[%d:%d], v%d offset:1024" % (XB[p_next], XB[p_next] + 3, VA))
    else:
        A("v_lshl_add_u32 v%d, s%d, 5, %%[lbB]" % (VB, HDR2))
        A("ds_read_b128 v[%d:%d], v%d" % (XA[p_next], XA[p_next] + 3, VA))
        A("ds_read_b128 v[%d:%d], v%d" % (XB[p_next], XB[p_next] + 3, VB))
    if stride:
        A("ds_read_b128 v[%d:%d], v%d offset:%d" % (P0[p_next], P0[p_next] + 3, VP, stride))
        A("v_add_u32 v%d, %d, v%d" % (VP, stride, VP))
    else:
        A("ds_read_b128 v[%d:%d], v%d" % (P0[p_next], P0[p_next] + 3, VP))


CNT = 32                     # groups left in the current bucket after the current one


def ix_field(dst, src, r):
    """Accumulator VGPR offset of record r (r = 1, 2 from meta; 3..5 from meta2; 'n' = the next
    group's record 0): one scalar op."""
    off = {1: 7, 2: 0, 3: 0, 4: 7, 5: 14, "n": 14}[r]
    if off == 0:
        return "s_and_b32 s%d, s%d, 0x7f" % (dst, src)
    return bfe(dst, src, off, 7)


def body2(L, n, p, band):
    """Group k (phase p).  On entry: X(k), P0(k) were requested at the top of group k-1;
    s[HDR2] = row offset / 32 of group k+1, s[IX0[p]] = accumulator of this group's record 0."""
    A = L.append
    stride = 32 if n > 3 else 16
    A("ESC2_L%d_%d_%%=:" % (n, p))
    A("s_set_gpr_idx_idx 0")
    if n > 3:
        A("ds_read_b128 v[%d:%d], v%d offset:16" % (P1, P1 + 3, VP))
    prefetch2(L, 1 - p, stride, band)          # group k+1: a whole group of FMAs to land in
    if "noxp" in ABL:
        A("s_waitcnt lgkmcnt(%d)" % (2 if n > 3 else 1))
    else:
        A("s_waitcnt lgkmcnt(%d)" % (4 if n > 3 else 3))   # X(k), P0(k) landed
    A("s_set_gpr_idx_idx s%d" % IX0[p])
    pk4v(L, 0, p)
    A("v_readfirstlane_b32 s%d, v%d" % (META, P0[p]))
    A("s_lshr_b32 s%d, s%d, 21" % (HDR2, META))          # row offset / 32 of group k+2
    A(ix_field(IX0[1 - p], META, "n"))
    for r in range(1, n):
        t = IXT[r % 2]
        if r == 3:
            A("s_waitcnt lgkmcnt(%d)" % (1 if "noxp" in ABL else 3))   # the second quad (older than the prefetches) landed
            A("v_readfirstlane_b32 s%d, v%d" % (META2, P1))
        A(ix_field(t, META if r < 3 else META2, r))
        A("s_set_gpr_idx_idx s%d" % t)
        pk4v(L, r, p)
    # s_add_u32 x, x, -1: SCC = carry = (x was not 0) = another group follows in this bucket
    A("s_add_u32 s%d, s%d, -1" % (CNT, CNT))
    if p == 0:
        A("s_cbranch_scc0 ESC2_E%d_1_%%=" % (n - 1))     # falls through into phase 1
    else:
        A("s_cbranch_scc1 ESC2_L%d_0_%%=" % n)
        A("s_branch ESC2_E%d_0_%%=" % (n - 1))


def generate2(band):
    L = []
    A = L.append
    A("s_waitcnt lgkmcnt(0)")
    for n in range(1, MAX_SLOTS2 + 1):
        A("s_mov_b32 s%d, %%[h%d]" % (END0 + n, 7 - n))
    A("s_mov_b32 s%d, 0" % (END0 + MAX_SLOTS2 + 1))
    A("s_cmp_eq_u32 s%d, 0" % (END0 + 1))
    A("s_cbranch_scc1 ESC2_X_%=")
    A("s_mov_b32 s%d, %%[h0]" % META)
    A("s_and_b32 s%d, s%d, 0x7ff" % (HDR2, META))        # group 0
    A(ix_field(IX0[0], META, "n"))
    A("v_mov_b32 v%d, %%[sbase]" % VP)
    prefetch2(L, 0, 0, band)
    A("s_lshr_b32 s%d, s%d, 21" % (HDR2, META))          # group 1
    A("s_set_gpr_idx_on s%d, gpr_idx(SRC2,DST)" % IX0[0])
    A("s_branch ESC2_E%d_0_%%=" % MAX_SLOTS2)
    for n in range(MAX_SLOTS2, 0, -1):
        for p in (0, 1):
            # bucket n: groups [END_(n+1), END_n)
            A("ESC2_E%d_%d_%%=:" % (n, p))
            A("s_sub_u32 s%d, s%d, s%d" % (CNT, END0 + n, END0 + n + 1))
            A("s_cmp_eq_u32 s%d, 0" % CNT)
            A("s_cbranch_scc1 ESC2_E%d_%d_%%=" % (n - 1, p))
            A("s_sub_u32 s%d, s%d, 1" % (CNT, CNT))
            A("s_branch ESC2_L%d_%d_%%=" % (n, p))
        body2(L, n, 0, band)
        body2(L, n, 1, band)
    A("ESC2_E0_0_%=:")
    A("ESC2_E0_1_%=:")
    A("s_set_gpr_idx_off")
    A("ESC2_X_%=:")
    A("s_waitcnt lgkmcnt(0)")
    return L


def clobbers():
    c = ["memory", "scc", "m0", "exec"]
    c += ["s%d" % i for i in range(32, 64)]
    c += ["v%d" % i for i in range(VA, 256)]
    return c


def emit_macro(out, name, lines):
    out.write("#define %s \\\n" % name)
    for ln in lines:
        out.write('  "%s\\n" \\\n' % ln)
    out.write('  ""\n')


def main():
    out = sys.stdout
    out.write("// GENERATED by gen_stream_loop.py -- do not edit.\n")
    out.write("#define ESC_NV %d\n#define ESC_NACC_TILE %d\n" % (NV, NACC_TILE))
    emit_macro(out, "ESC_STREAM_LOOP_ASM", generate())
    ABL.add("band")
    emit_macro(out, "ESC_STREAM_LOOP_ASM_BAND", generate())
    ABL.discard("band")
    # timing-only ablations (wrong results), compiled in with -DESCOIN_ABLATIONS
    out.write("#ifdef ESCOIN_ABLATIONS\n")
    for name in ("nopk", "noxp", "nop4", "vnop4"):
        ABL.add(name)
        emit_macro(out, "ESC_STREAM_LOOP_ASM_" + name.upper(), generate())
        if name in ("nopk", "noxp"):
            emit_macro(out, "ESC2_LOOP_ASM_" + name.upper(), generate2(False))
        ABL.discard(name)
    out.write("#endif\n")
    emit_macro(out, "ESC_STREAM_INIT_ASM", generate_init())
    emit_macro(out, "ESC2_LOOP_ASM", generate2(False))
    emit_macro(out, "ESC2_LOOP_ASM_BAND", generate2(True))
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
    # ESC_READ_QUAD_<tile>_<q>: asm text moving accumulator quad q of a tile into %0..%3
    out.write("#define ESC_NQUADS_TILE %d\n" % (NACC_TILE // 4))
    for tile, base in ((0, ACC_A), (1, ACC_B)):
        for q in range(NACC_TILE // 4):
            r = base + 4 * q
            out.write('#define ESC_READ_QUAD_%d_%d "v_mov_b32 %%0, v%d\\n v_mov_b32 %%1, v%d\\n '
                      'v_mov_b32 %%2, v%d\\n v_mov_b32 %%3, v%d\\n"\n' % (tile, q, r, r + 1, r + 2, r + 3))


if __name__ == "__main__":
    main()
