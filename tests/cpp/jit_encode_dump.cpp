// tests/cpp/jit_encode_dump.cpp -- prints "<assembly text> | <hex dwords>" for random operands of
// every instruction form jit_codegen.h encodes; tests/test_jit_codegen.py assembles the text with
// llvm-mc and compares the bytes.  Test code.
#include <cstdio>
#include <vector>

#include "jit_codegen.h"

using namespace escoin::jit;

static unsigned st = 99;
static unsigned rnd(unsigned n) { st = st * 1664525u + 1013904223u; return (st >> 8) % n; }

static void dump(const char *text, const std::vector<uint32_t> &c) {
  printf("%s |", text);
  for (uint32_t d : c) printf(" %08x", d);
  printf("\n");
}

int main() {
  char buf[256];
  for (int i = 0; i < 200; ++i) {
    std::vector<uint32_t> c;
    const int vdst = 36 + 4 * rnd(6), vaddr = 32 + rnd(2);
    const unsigned off = 16 * rnd(4096);
    enc_ds_read_b128(c, vdst, vaddr, off);
    if (off) snprintf(buf, sizeof buf, "ds_read_b128 v[%d:%d], v%d offset:%u", vdst, vdst + 3, vaddr, off);
    else snprintf(buf, sizeof buf, "ds_read_b128 v[%d:%d], v%d", vdst, vdst + 3, vaddr);
    dump(buf, c);
    c.clear();
    const int n = rnd(16);
    enc_waitcnt_lgkm(c, n);
    snprintf(buf, sizeof buf, "s_waitcnt lgkmcnt(%d)", n);
    dump(buf, c);
    c.clear();
    // (literals that happen to be inline constants would be assembled in the short form: the
    // generator always uses the long one, which is just as valid -- keep them out of the comparison)
    uint32_t lit = (rnd(65536) << 16) | rnd(65536);
    if (lit <= 64 || lit >= 0xFFFFFFF0u || lit == 0x3f000000u || lit == 0xbf000000u || lit == 0x3f800000u ||
        lit == 0xbf800000u || lit == 0x40000000u || lit == 0xc0000000u || lit == 0x40800000u || lit == 0xc0800000u ||
        lit == 0x3e22f983u)
      lit = 0x12345678u;
    const int sd = rnd(2) ? kSWeight0 : kSWeight1;
    enc_s_mov_lit(c, sd, lit);
    snprintf(buf, sizeof buf, "s_mov_b32 s%d, 0x%x", sd, lit);
    dump(buf, c);
    c.clear();
    const int acc = 64 + 2 * rnd(96), x = 36 + 2 * rnd(12);
    enc_pk_fma(c, acc, sd, x);
    snprintf(buf, sizeof buf, "v_pk_fma_f32 v[%d:%d], s[%d:%d], v[%d:%d], v[%d:%d] op_sel_hi:[0,1,1]", acc, acc + 1, sd,
             sd + 1, x, x + 1, acc, acc + 1);
    dump(buf, c);
  }
  // accumulators initialised by the code (Options::self_zero)
  for (int i = 0; i < 60; ++i) {
    std::vector<uint32_t> c;
    const int acc = 64 + 2 * rnd(96), x = 36 + 2 * rnd(12);
    const int lit_pair = rnd(2) ? kSWeight0 : kSWeight1, pair = ((i & 1) ? kSWBuf1 : kSWBuf0) + 2 * rnd(8);
    enc_pk_mul(c, acc, lit_pair, x);
    snprintf(buf, sizeof buf, "v_pk_mul_f32 v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[0,1]", acc, acc + 1, lit_pair, lit_pair + 1, x, x + 1);
    dump(buf, c);
    c.clear();
    enc_pk_mul(c, acc, pair, x);
    snprintf(buf, sizeof buf, "v_pk_mul_f32 v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[0,1]", acc, acc + 1, pair, pair + 1, x, x + 1);
    dump(buf, c);
    c.clear();
    enc_pk_mul_hi(c, acc, pair, x);
    snprintf(buf, sizeof buf, "v_pk_mul_f32 v[%d:%d], s[%d:%d], v[%d:%d] op_sel:[1,0] op_sel_hi:[1,1]", acc, acc + 1, pair, pair + 1, x, x + 1);
    dump(buf, c);
    c.clear();
    enc_pk_zero(c, acc);
    snprintf(buf, sizeof buf, "v_pk_mov_b32 v[%d:%d], 0, 0", acc, acc + 1);
    dump(buf, c);
  }
  // weights through the scalar cache (Options::sweights)
  for (int i = 0; i < 60; ++i) {
    std::vector<uint32_t> c;
    const int acc = 64 + 2 * rnd(96), x = 36 + 2 * rnd(12), pair = ((i & 1) ? kSWBuf1 : kSWBuf0) + 2 * rnd(8);
    enc_pk_fma_hi(c, acc, pair, x);
    snprintf(buf, sizeof buf, "v_pk_fma_f32 v[%d:%d], s[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0,0] op_sel_hi:[1,1,1]", acc, acc + 1, pair, pair + 1, x, x + 1,
             acc, acc + 1);
    dump(buf, c);
    c.clear();
    enc_pk_fma(c, acc, pair, x);
    snprintf(buf, sizeof buf, "v_pk_fma_f32 v[%d:%d], s[%d:%d], v[%d:%d], v[%d:%d] op_sel_hi:[0,1,1]", acc, acc + 1, pair, pair + 1, x, x + 1, acc, acc + 1);
    dump(buf, c);
    c.clear();
    const int sd = (i & 1) ? kSWBuf1 : kSWBuf0;
    const unsigned off = 64 * rnd(16000);
    enc_s_load_x16(c, sd, kSWBase, off);
    snprintf(buf, sizeof buf, "s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x", sd, sd + 15, kSWBase, kSWBase + 1, off);
    dump(buf, c);
    c.clear();
    const int dw = 1 + (int)rnd(30000);
    enc_s_branch(c, dw);
    snprintf(buf, sizeof buf, "s_branch %d", dw);
    dump(buf, c);
    c.clear();
    enc_getpc(c, kSWBase);
    snprintf(buf, sizeof buf, "s_getpc_b64 s[%d:%d]", kSWBase, kSWBase + 1);
    dump(buf, c);
    c.clear();
    const uint32_t lit = 0x100u + 64u * rnd(100000);
    enc_s_add_lit(c, kSWBase, lit);
    snprintf(buf, sizeof buf, "s_add_u32 s%d, s%d, 0x%x", kSWBase, kSWBase, lit);
    dump(buf, c);
    c.clear();
    enc_s_addc(c, kSWBase + 1, false);
    snprintf(buf, sizeof buf, "s_addc_u32 s%d, s%d, 0", kSWBase + 1, kSWBase + 1);
    dump(buf, c);
  }
  for (int i = 0; i < 40; ++i) {
    std::vector<uint32_t> c;
    const int vd = (i & 1) ? kVTab1 + (int)rnd(2) : kVTab0;
    const unsigned off = 4 * rnd(16384);
    enc_ds_read_b32(c, vd, kVTabAddr, off);
    if (off) snprintf(buf, sizeof buf, "ds_read_b32 v%d, v%d offset:%u", vd, kVTabAddr, off);
    else snprintf(buf, sizeof buf, "ds_read_b32 v%d, v%d", vd, kVTabAddr);
    dump(buf, c);
    c.clear();
    const uint32_t lit = 1024u * (1 + rnd(60)) + 65u;
    enc_s_add_m0_lit(c, kSFillBase, lit);
    snprintf(buf, sizeof buf, "s_add_u32 m0, s%d, 0x%x", kSFillBase, lit);
    dump(buf, c);
    c.clear();
    enc_s_mov_lit(c, kSSoff, 0x00abc123u + i);
    snprintf(buf, sizeof buf, "s_mov_b32 s%d, 0x%x", kSSoff, 0x00abc123u + i);
    dump(buf, c);
    c.clear();
    enc_s_mov_lit(c, kSExecLo, 0x0000ffffu + i * 4096u);
    snprintf(buf, sizeof buf, "s_mov_b32 exec_lo, 0x%x", 0x0000ffffu + i * 4096u);
    dump(buf, c);
    c.clear();
    enc_s_mov_lit(c, kSExecHi, 0x00ffff00u + i * 4096u);
    snprintf(buf, sizeof buf, "s_mov_b32 exec_hi, 0x%x", 0x00ffff00u + i * 4096u);
    dump(buf, c);
    c.clear();
    enc_lds_dma16(c, vd, kSRsrc, kSSoff, (i & 2) != 0);
    snprintf(buf, sizeof buf, "buffer_load_dwordx4 v%d, s[%d:%d], s%d offen%s lds", vd, kSRsrc, kSRsrc + 3, kSSoff, (i & 2) ? " nt" : "");
    dump(buf, c);
    c.clear();
    enc_s_add_lit(c, kSPref, 0x1000u + 77u * i);
    snprintf(buf, sizeof buf, "s_add_u32 s%d, s%d, 0x%x", kSPref, kSPref, 0x1000u + 77u * i);
    dump(buf, c);
  }
  // block-to-block chaining (ChainPlan): counted vmcnt waits, the buffer rotation, the address moves
  for (int i = 0; i < 64; ++i) {
    std::vector<uint32_t> c;
    enc_waitcnt_vm(c, i);
    snprintf(buf, sizeof buf, "s_waitcnt vmcnt(%d)", i);
    dump(buf, c);
    c.clear();
    const int sa = 44 + (int)rnd(12), sb = 44 + (int)rnd(12), sc = 44 + (int)rnd(12);
    enc_s_mov(c, sa, sb);
    snprintf(buf, sizeof buf, "s_mov_b32 s%d, s%d", sa, sb);
    dump(buf, c);
    c.clear();
    const uint32_t lit = 1024u * (1 + rnd(160)) + 1024u * 65u;
    enc_s_add_u32_lit(c, sa, sb, lit);
    snprintf(buf, sizeof buf, "s_add_u32 s%d, s%d, 0x%x", sa, sb, lit);
    dump(buf, c);
    c.clear();
    enc_s_sub_u32(c, sa, sb, sc);
    snprintf(buf, sizeof buf, "s_sub_u32 s%d, s%d, s%d", sa, sb, sc);
    dump(buf, c);
    c.clear();
    enc_s_cmp_lt_u32_lit(c, sa, lit);
    snprintf(buf, sizeof buf, "s_cmp_lt_u32 s%d, 0x%x", sa, lit);
    dump(buf, c);
    c.clear();
    enc_s_cselect_or_zero(c, sa, sb);
    snprintf(buf, sizeof buf, "s_cselect_b32 s%d, s%d, 0", sa, sb);
    dump(buf, c);
    c.clear();
    const int vd = 32 + (int)rnd(3), vs = 32 + (int)rnd(3);
    enc_v_add_u32_s(c, vd, sa, vs);
    snprintf(buf, sizeof buf, "v_add_u32 v%d, s%d, v%d", vd, sa, vs);
    dump(buf, c);
  }
  std::vector<uint32_t> c;
  enc_barrier(c);
  dump("s_barrier", c);
  c.clear();
  enc_exec_all(c);
  dump("s_mov_b64 exec, -1", c);
  c.clear();
  enc_getpc(c, kSPref);
  dump("s_getpc_b64 s[50:51]", c);
  c.clear();
  enc_s_addc(c, kSPref + 1, false);
  dump("s_addc_u32 s51, s51, 0", c);
  c.clear();
  enc_s_addc(c, kSPref + 1, true);
  dump("s_addc_u32 s51, s51, -1", c);
  c.clear();
  enc_global_load_dword(c, kVPrefDead, kVPrefLane, kSPref);
  dump("global_load_dword v62, v63, s[50:51]", c);
  c.clear();
  enc_setpc_return(c);
  dump("s_setpc_b64 s[30:31]", c);
  c.clear();
  enc_setprio(c, 1);
  dump("s_setprio 1", c);
  c.clear();
  enc_nop(c);
  dump("s_nop 0", c);
  return 0;
}
