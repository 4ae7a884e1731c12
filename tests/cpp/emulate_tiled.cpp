// tests/cpp/emulate_tiled.cpp -- CPU emulation of the tiled kernel's dataflow (test code).
//
// Builds the tiling + weight stream with the PRODUCT's stream builder (csrc/stream_builder.cpp)
// and then walks it exactly the way sconv_tiled.hip does -- LDS planes, lane -> quad mapping,
// bucket walk driven by the END_n counts, leads handed on by the previous groups' meta words,
// dst-relative accumulator classes, shift-and-sum
// epilogue -- and compares against a plain dense convolution.  Lets the stream format and all
// index arithmetic be validated without a GPU.  Not part of the product; not the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jit_codegen.h"
#include "stream_builder.h"

using namespace escoin;
static const int kAccAll = kTilesPerLane * kAccRegsPerTile;

static unsigned rng_state = 12345;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

struct Case { int N, C, H, W, M, KH, KW, ph, pw, group; float sparsity; int waves; int lds; int ncu = 1; };

// ---- interpreter of the code jit_codegen.cpp generates (the five instruction forms it emits) ----
// Registers of one wave: v[lane][256].  LDS reads land only when a counted s_waitcnt retires them
// (in order), and an FMA whose input register still has a read pending is an error: a wrong wait
// count in the generator shows up here, not only on the GPU.
struct JitWave {
  std::vector<float> v;   // [64][256]
  JitWave() : v(64 * 256, 0.f) {}
};
struct PendingRead { int vdst; int n; std::vector<float> data; };   // data: [64][n]
struct DmaPiece { uint32_t m0, soff; unsigned long long exec; std::vector<uint32_t> voff; bool nt; uint32_t records = 0; bool next_table = false; int vm_id = -1; };
struct JitPref { long long base = -1; int touches = 0; };
static size_t last_return_pc = 0;   // code touches of a unit: first address, count
struct JitDmaCtx {
  const std::vector<uint32_t> *tabx = nullptr;   // the quad table of the tile being staged (period entries)
  uint32_t fill_base = 0;
  std::vector<DmaPiece> issued;
  // chained units (jit_codegen.h ChainPlan): the NEXT tile's table (reached once v34 has moved by tab_delta)
  const std::vector<uint32_t> *tabx_next = nullptr;
};
// What a chain of units (one tile of one wave) runs against: one LDS image per block, the buffer rotation,
// and a hook at every in-code barrier that checks the pieces the unit just finished has issued.
struct JitChainCtx {
  const std::vector<std::vector<float>> *lds_of_block = nullptr;
  int nbuf = 2, n_icb = 1, ahead = 1;
  uint32_t buf_bytes = 0, walk_base0 = 0, fill_base0 = 0, records_cur = 0x1000, records_next = 0x2000, tab_delta = 0x40;
  int blk = 0;                 // block being walked
  int barriers = 0;
  std::vector<size_t> unit_first_piece;    // index into JitDmaCtx::issued of each unit's first piece
  std::vector<JitPref> prefs;              // code touches per unit
  std::vector<size_t> unit_end;            // pc just past each unit's last instruction
};

static long g_weight_lines = 0;      // s_load_dwordx16 executed (Options::sweights)
static long g_first_products = 0, g_cleared = 0;   // v_pk_mul_f32 / v_pk_mov_b32 executed (Options::self_zero)
static int jit_run_unit(const std::vector<uint32_t> &code, size_t pc, JitWave &w, const std::vector<float> &lds,
                        const std::vector<uint32_t> &laneA /* LDS byte address of the lane's tile-A quad */,
                        JitDmaCtx *dma = nullptr, JitPref *pref = nullptr, JitChainCtx *chain = nullptr) {
  long long s50 = -1;
  uint32_t addr_shift = 0, addr_shift_b = 0, tab_shift = 0;   // what the code has added to v32 / v33 / v34
  bool scc = false;
  std::vector<int> vm_out;     // outstanding vector-memory operations of this wave, oldest first: block a piece stages, -1 = code touch
  int vm_next_id = 0;
  (void)vm_next_id;
  std::vector<PendingRead> pending;
  uint32_t sreg[128] = {0};
  uint32_t m0 = 0;
  unsigned long long exec = ~0ull;
  bool sw_base_set = false;             // s[88:89] hold this unit's weight-line address
  bool smem_pending[2] = {false, false};   // a line is on its way into s[56:71] / s[72:87]: lands at the next lgkmcnt(0)
  uint32_t smem_data[2][16];
  bool tab_unissued[256] = {false};     // a table entry was read into this register and its piece has not gone out
  if (dma) sreg[48] = dma->fill_base;
  if (chain) {
    sreg[52] = chain->walk_base0; sreg[48] = chain->fill_base0; sreg[53] = chain->records_next; sreg[54] = chain->tab_delta;
    sreg[46] = chain->n_icb <= chain->ahead ? chain->records_next : chain->records_cur;
    addr_shift = addr_shift_b = chain->walk_base0;
    tab_shift = chain->n_icb <= chain->ahead ? chain->tab_delta : 0u;
    chain->unit_first_piece.assign(1, 0);
    chain->prefs.assign(1, JitPref());
    pref = &chain->prefs.back();
  }
  auto retire_to = [&](size_t keep) {
    while (pending.size() > keep) {
      const PendingRead &r = pending.front();
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < r.n; ++e) w.v[(size_t)lane * 256 + r.vdst + e] = r.data[lane * r.n + e];
      pending.erase(pending.begin());
    }
  };
  auto is_pending = [&](int reg) {
    for (const PendingRead &r : pending)
      if (reg >= r.vdst && reg < r.vdst + r.n) return true;
    return false;
  };
  for (long steps = 0; steps < 100000000; ++steps) {
    if (pc / 4 >= code.size()) { printf("jit: ran off the end of the code\n"); return 3; }
    const uint32_t d0 = code[pc / 4];
    if (d0 == 0xBE801D1Eu) {   // s_setpc_b64 s[30:31]
      if (!pending.empty()) { printf("jit: unit returns with LDS reads pending\n"); return 3; }
      if (exec != ~0ull) { printf("jit: unit returns with a partial EXEC\n"); return 3; }
      last_return_pc = pc;
      if (chain) {
        if (chain->blk != chain->n_icb - 1) { printf("jit chain: returned after block %d of %d\n", chain->blk, chain->n_icb); return 3; }
        chain->unit_end.push_back(pc + 4);
      }
      return 0;
    }
    if (d0 == 0xBEFE01C1u) { exec = ~0ull; pc += 4; continue; }   // s_mov_b64 exec, -1
    // ---- weights through the scalar cache (jit_codegen.h Options::sweights) ----
    if (d0 == (0xBE801C00u | (88u << 16))) { sreg[88] = (uint32_t)(pc + 4); sreg[89] = 0; sw_base_set = true; pc += 4; continue; }   // s_getpc_b64 s[88:89]
    if (d0 == 0x82598059u) { pc += 4; continue; }                  // s_addc_u32 s89, s89, 0
    if ((d0 & 0xFFFFE03Fu) == (0xC0120000u | 44u)) {               // s_load_dwordx16 s[sd:sd+15], s[88:89], imm
      const int sd = (int)((d0 >> 6) & 0x7F);
      const uint32_t off = code[pc / 4 + 1];
      if (!sw_base_set || (sd != 56 && sd != 72) || (off & 63)) { printf("jit: bad s_load_dwordx16 (sd %d, offset %u)\n", sd, off); return 3; }
      const size_t at = ((size_t)sreg[88] + off) / 4;
      if (((size_t)sreg[88] + off) % 64 || at + 16 > code.size()) { printf("jit: weight line at %zu is not a 64-byte line of the blob\n", at * 4); return 3; }
      if (smem_pending[sd == 56 ? 0 : 1]) { printf("jit: weight buffer s%d loaded twice without a wait\n", sd); return 3; }
      smem_pending[sd == 56 ? 0 : 1] = true;
      ++g_weight_lines;
      for (int e = 0; e < 16; ++e) smem_data[sd == 56 ? 0 : 1][e] = code[at + e];
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFF0000u) == 0xBF820000u) {                       // s_branch: over a unit's weight lines into the next unit
      const int simm = (int16_t)(d0 & 0xFFFFu);
      if (!chain || simm <= 0) { printf("jit: unexpected s_branch %d\n", simm); return 3; }
      pc += 4 + 4 * (size_t)simm;
      if (pc % 64) { printf("jit: s_branch lands off a unit's 64-byte boundary\n"); return 3; }
      continue;
    }
    if (d0 == 0xBEB21C00u) { s50 = (long long)pc + 4; pc += 4; continue; }   // s_getpc_b64 s[50:51]
    if (d0 == 0x8032FF32u) { s50 += (long long)(int32_t)code[pc / 4 + 1]; pc += 8; continue; }   // s_add_u32 s50, s50, lit (+ the carry below)
    if (d0 == 0x82338033u || d0 == 0x8233C133u) {   // s_addc_u32 s51, s51, 0 | -1: the sign of the literal just added
      const bool neg = (int32_t)code[pc / 4 - 1] < 0;
      if (neg != (d0 == 0x8233C133u)) { printf("jit: carry word does not match the sign of the distance\n"); return 3; }
      pc += 4;
      continue;
    }
    if (d0 == 0xDC508000u) {   // global_load_dword v62, v63, s[50:51]: a touch of the next unit's code
      if (code[pc / 4 + 1] != (0x3Fu | (50u << 16) | (62u << 24)) || !pref || s50 < 0) { printf("jit: bad code touch\n"); return 3; }
      if (s50 + 4096 > (long long)code.size() * 4) { printf("jit: code touch past the blob\n"); return 3; }
      if (pref->touches == 0) pref->base = s50;
      else if (s50 != pref->base + 4096ll * pref->touches) { printf("jit: code touches not contiguous\n"); return 3; }
      pref->touches++;
      vm_out.push_back(-1);
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFF0000u) == 0xBF8C0000u && (d0 & 0x0F70u) == 0x0F70u && (d0 & 0xF0FFu) != 0xC07Fu) {   // s_waitcnt vmcnt(n)
      if (!chain) { printf("jit: s_waitcnt vmcnt outside a chain\n"); return 3; }
      const size_t n = (d0 & 15u) | (((d0 >> 14) & 3u) << 4);
      while (vm_out.size() > n) vm_out.erase(vm_out.begin());     // (in order: LDS-DMA and plain loads retire oldest first)
      pc += 4;
      continue;
    }
    if (d0 == 0xBF8A0000u) {   // s_barrier: the chain moves on to the next block
      if (!chain) { printf("jit: s_barrier outside a chain\n"); return 3; }
      if (!pending.empty()) { printf("jit chain: barrier with LDS reads pending\n"); return 3; }
      if (exec != ~0ull) { printf("jit chain: barrier with a partial EXEC\n"); return 3; }
      for (int b : vm_out)      // (entries: the step a piece stages, counted on from this tile's block 0)
        if (b == chain->blk + 1) { printf("jit chain: block %d entered with its own pieces in flight\n", chain->blk + 1); return 3; }
      chain->barriers++;
      chain->unit_end.push_back(pc + 4);
      chain->blk++;
      if (chain->blk >= chain->n_icb) { printf("jit chain: more barriers than blocks\n"); return 3; }
      chain->unit_first_piece.push_back(dma ? dma->issued.size() : 0);
      chain->prefs.push_back(JitPref());
      pref = &chain->prefs.back();
      s50 = -1;
      pc += 4;
      continue;
    }
    if ((d0 & 0xFF80FF00u) == 0xBE800000u && (d0 & 0xFFu) < 102u) {   // s_mov_b32 s, s
      sreg[(d0 >> 16) & 0x7F] = sreg[d0 & 0xFF];
      pc += 4;
      continue;
    }
    if ((d0 & 0xFF80FF00u) == 0x8000FF00u && ((d0 >> 16) & 0x7F) != 50 && ((d0 >> 16) & 0x7F) < 102) {   // s_add_u32 s, s, literal (not the code-touch address)
      sreg[(d0 >> 16) & 0x7F] = sreg[d0 & 0xFF] + code[pc / 4 + 1];
      pc += 8;
      continue;
    }
    if ((d0 & 0xFF800000u) == 0x80800000u) {   // s_sub_u32 s, s, s
      sreg[(d0 >> 16) & 0x7F] = sreg[d0 & 0xFF] - sreg[(d0 >> 8) & 0xFF];
      pc += 4;
      continue;
    }
    if ((d0 & 0xFFFFFF00u) == 0xBF0AFF00u) {   // s_cmp_lt_u32 s, literal
      scc = sreg[d0 & 0xFF] < code[pc / 4 + 1];
      pc += 8;
      continue;
    }
    if ((d0 & 0xFF80FF00u) == 0x85008000u) {   // s_cselect_b32 s, s, 0
      sreg[(d0 >> 16) & 0x7F] = scc ? sreg[d0 & 0xFF] : 0u;
      pc += 4;
      continue;
    }
    if ((d0 & 0xFE000000u) == 0x68000000u && (d0 & 0x1FFu) < 102u) {   // v_add_u32 v, s, v: the chain's address registers only
      const int vd = (int)((d0 >> 17) & 0xFF), vs = (int)((d0 >> 9) & 0xFF);
      if (!chain || vd != vs || (vd != 32 && vd != 33 && vd != 34)) { printf("jit: unexpected v_add_u32 v%d, s, v%d\n", vd, vs); return 3; }
      if (vd == 32) addr_shift += sreg[d0 & 0x1FF];
      else if (vd == 33) addr_shift_b += sreg[d0 & 0x1FF];
      else tab_shift += sreg[d0 & 0x1FF];
      pc += 4;
      continue;
    }
    if ((d0 & 0xFFFF0000u) == 0xBF8C0000u) {   // s_waitcnt lgkmcnt(n)
      if ((d0 & 0xF0FFu) != 0xC07Fu) { printf("jit: unexpected s_waitcnt fields\n"); return 3; }
      retire_to((d0 >> 8) & 15);
      if (((d0 >> 8) & 15) == 0)          // (scalar loads return out of order with LDS reads: only a wait for everything settles them)
        for (int b = 0; b < 2; ++b)
          if (smem_pending[b]) {
            smem_pending[b] = false;
            for (int e = 0; e < 16; ++e) sreg[(b ? 72 : 56) + e] = smem_data[b][e];
          }
      pc += 4;
      continue;
    }
    if ((d0 & 0xFFFF0000u) == 0xBF8F0000u || d0 == 0xBF800000u) { pc += 4; continue; }   // s_setprio, s_nop
    const uint32_t d1 = code[pc / 4 + 1];
    if ((d0 & 0xFFFF0000u) == 0xD9FE0000u) {   // ds_read_b128
      const unsigned off = d0 & 0xFFFFu;
      const int vdst = (int)(d1 >> 24), vaddr = (int)(d1 & 0xFF);
      // (destinations: the input sets v[36:59]; code without a tile B may also use tile B's accumulators v[160:255])
      if ((d1 & 0x00FFFF00u) != 0 || (vaddr != 32 && vaddr != 33) || vdst < 36 || (vdst + 3 > 59 && (vdst < 160 || vdst + 3 > 255 || vaddr != 32))) { printf("jit: bad ds_read operands\n"); return 3; }
      if (is_pending(vdst)) { printf("jit: read into a register with a read pending\n"); return 3; }
      PendingRead r;
      r.vdst = vdst;
      r.n = 4;
      r.data.resize(64 * 4);
      const std::vector<float> *img = &lds;
      uint32_t shift = 0;
      if (chain) {
        // the address registers must point into the buffer the rotation has reached, and what is there is this block's image
        const uint32_t sh = vaddr == 33 ? addr_shift_b : addr_shift;
        const uint32_t want_buf = (chain->walk_base0 / chain->buf_bytes + (uint32_t)chain->blk) % (uint32_t)chain->nbuf;
        if (sh != want_buf * chain->buf_bytes) { printf("jit chain: block %d reads buffer offset %u, rotation says %u\n", chain->blk, sh, want_buf * chain->buf_bytes); return 3; }
        if (sh != sreg[52]) { printf("jit chain: v32 and s52 disagree\n"); return 3; }
        img = &(*chain->lds_of_block)[chain->blk];
        (void)shift;
      }
      for (int lane = 0; lane < 64; ++lane) {
        const size_t a = ((size_t)laneA[lane] + (vaddr == 33 ? 1024u : 0u) + off) / 4;
        for (int e = 0; e < 4; ++e) r.data[lane * 4 + e] = (a + e < img->size()) ? (*img)[a + e] : 0.f;
      }
      pending.push_back(r);
      if (pending.size() > 15) { printf("jit: more than 15 LDS reads in flight\n"); return 3; }
      pc += 8;
      continue;
    }
    if ((d0 & 0xFF80FFFFu) == 0xBE8000FFu) {   // s_mov_b32 s, literal (also exec_lo / exec_hi)
      const int sd = (d0 >> 16) & 0x7F;
      if (sd == 0x7E) exec = (exec & 0xFFFFFFFF00000000ull) | d1;
      else if (sd == 0x7F) exec = (exec & 0xFFFFFFFFull) | ((unsigned long long)d1 << 32);
      else sreg[sd] = d1;
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFF0000u) == 0xD86C0000u) {   // ds_read_b32: an entry of the quad table
      const unsigned off = d0 & 0xFFFFu;
      const int vdst = (int)(d1 >> 24), vaddr = (int)(d1 & 0xFF);
      if (!dma || vaddr != 34 || (vdst != 35 && (vdst < 60 || vdst > 61)) || (off & 3)) { printf("jit: bad ds_read_b32\n"); return 3; }
      if (tab_unissued[vdst]) { printf("jit: table register v%d overwritten before its piece went out\n", vdst); return 3; }
      tab_unissued[vdst] = true;
      if (is_pending(vdst)) { printf("jit: table read into a register with a read pending\n"); return 3; }
      PendingRead r;
      r.vdst = vdst;
      r.n = 1;
      r.data.resize(64);
      const std::vector<uint32_t> *tb = dma->tabx;
      if (chain) {
        if (tab_shift != 0 && tab_shift != chain->tab_delta) { printf("jit chain: table address moved by %u\n", tab_shift); return 3; }
        if (tab_shift) tb = dma->tabx_next;
      }
      for (int lane = 0; lane < 64; ++lane) {
        const size_t e = off / 4 + lane;
        if (e >= tb->size()) { printf("jit: table read past the table\n"); return 3; }
        uint32_t u = (*tb)[e];
        std::memcpy(&r.data[lane], &u, 4);
      }
      pending.push_back(r);
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFFFF00u) == 0x807CFF00u) {   // s_add_u32 m0, s, literal
      m0 = sreg[d0 & 0xFF] + d1;
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFDFFFFu) == 0xE05D1000u) {   // buffer_load_dwordx4 v, s[44:47], s49 offen [nt] lds
      const int vaddr = (int)(d1 & 0xFF), srsrc = (int)((d1 >> 16) & 0x1F) * 4, soff = (int)(d1 >> 24);
      if (!dma || srsrc != 44 || soff != 49 || (vaddr != 35 && (vaddr < 60 || vaddr > 61))) { printf("jit: bad LDS-DMA operands\n"); return 3; }
      if (!tab_unissued[vaddr]) { printf("jit: LDS-DMA through v%d without a fresh table entry\n", vaddr); return 3; }
      tab_unissued[vaddr] = false;
      if (is_pending(vaddr)) { printf("jit: LDS-DMA issued before its table entry landed\n"); return 3; }
      DmaPiece pcs;
      pcs.m0 = m0; pcs.soff = sreg[49]; pcs.exec = exec; pcs.nt = (d0 >> 17) & 1;
      pcs.voff.resize(64);
      for (int lane = 0; lane < 64; ++lane) std::memcpy(&pcs.voff[lane], &w.v[(size_t)lane * 256 + vaddr], 4);
      pcs.records = sreg[46];
      pcs.next_table = tab_shift != 0;
      if (chain) {
        if (sreg[48] != ((chain->fill_base0 / chain->buf_bytes + (uint32_t)chain->blk) % (uint32_t)chain->nbuf) * chain->buf_bytes) {
          printf("jit chain: block %d fills buffer offset %u\n", chain->blk, sreg[48]);
          return 3;
        }
        vm_out.push_back(chain->blk + chain->ahead);
      }
      dma->issued.push_back(pcs);
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFFF700u) == 0xD3B04000u) {   // v_pk_fma_f32 acc, s[w:w+1], v[x:x+1], acc op_sel_hi:[0,1,1] | op_sel:[1,0,0] op_sel_hi:[1,1,1]
      const int acc = (int)(d0 & 0xFF), sw = (int)(d1 & 0x1FF);
      const int hi = (d0 >> 11) & 1;        // the weight is the pair's high dword
      const int x = (int)((d1 >> 9) & 0x1FF) - 256, acc2 = (int)((d1 >> 18) & 0x1FF) - 256;
      if (acc != acc2 || acc < 64 || acc > 254 || (acc & 1) || x < 36 || (x > 58 && (x < 160 || x > 254 || acc >= 160)) || sw > 101 || (sw & 1) ||
          (d1 >> 27) != (hi ? 3u : 2u) || (hi && sw < 56)) { printf("jit: bad v_pk_fma_f32 operands\n"); return 3; }
      if (is_pending(x) || is_pending(x + 1)) { printf("jit: FMA reads v%d before its LDS read was waited for\n", x); return 3; }
      if (sw >= 56 && sw < 88 && smem_pending[sw >= 72 ? 1 : 0]) { printf("jit: FMA reads weight buffer s%d before its line was waited for\n", sw); return 3; }
      float wv;
      std::memcpy(&wv, &sreg[sw + hi], 4);
      for (int lane = 0; lane < 64; ++lane) {
        float *V = &w.v[(size_t)lane * 256];
        if (V[acc] != V[acc] || V[acc + 1] != V[acc + 1]) { printf("jit: FMA onto an accumulator nothing initialised (v%d)\n", acc); return 3; }
        V[acc] = std::fmaf(wv, V[x], V[acc]);
        V[acc + 1] = std::fmaf(wv, V[x + 1], V[acc + 1]);
      }
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFFF700u) == 0xD3B14000u) {   // v_pk_mul_f32 acc, s[w:w+1], v[x:x+1] op_sel_hi:[0,1] | op_sel:[1,0]: a quad's FIRST product (Options::self_zero)
      const int acc = (int)(d0 & 0xFF), sw = (int)(d1 & 0x1FF);
      const int hi = (d0 >> 11) & 1;
      const int x = (int)((d1 >> 9) & 0x1FF) - 256;
      if (((d1 >> 18) & 0x1FF) != 0 || acc < 64 || acc > 254 || (acc & 1) || x < 36 || (x > 58 && (x < 160 || x > 254 || acc >= 160)) || sw > 101 || (sw & 1) ||
          (d1 >> 27) != (hi ? 3u : 2u) || (hi && sw < 56)) { printf("jit: bad v_pk_mul_f32 operands\n"); return 3; }
      if (is_pending(x) || is_pending(x + 1)) { printf("jit: product reads v%d before its LDS read was waited for\n", x); return 3; }
      if (sw >= 56 && sw < 88 && smem_pending[sw >= 72 ? 1 : 0]) { printf("jit: product reads weight buffer s%d before its line was waited for\n", sw); return 3; }
      float wv;
      std::memcpy(&wv, &sreg[sw + hi], 4);
      for (int lane = 0; lane < 64; ++lane) {
        float *V = &w.v[(size_t)lane * 256];
        if (V[acc] == V[acc] || V[acc + 1] == V[acc + 1]) { printf("jit: v_pk_mul_f32 overwrites an initialised accumulator v%d\n", acc); return 3; }
        V[acc] = wv * V[x];
        V[acc + 1] = wv * V[x + 1];
      }
      ++g_first_products;
      pc += 8;
      continue;
    }
    if ((d0 & 0xFFFFFF00u) == 0xD3B34000u && d1 == 0x18010080u) {   // v_pk_mov_b32 acc, 0, 0: a quad block 0 never touches
      const int acc = (int)(d0 & 0xFF);
      if (acc < 64 || acc > 254 || (acc & 1)) { printf("jit: bad v_pk_mov_b32 destination\n"); return 3; }
      for (int lane = 0; lane < 64; ++lane) {
        float *V = &w.v[(size_t)lane * 256];
        if (V[acc] == V[acc] || V[acc + 1] == V[acc + 1]) { printf("jit: v_pk_mov_b32 clears an initialised accumulator v%d\n", acc); return 3; }
        V[acc] = V[acc + 1] = 0.f;
      }
      ++g_cleared;
      pc += 8;
      continue;
    }
    printf("jit: unknown instruction %08x at %zu\n", d0, pc);
    return 3;
  }
  return 3;
}

static int run(const Case &cs, bool use_jit) {
  ConvGeom g{cs.N, cs.C, cs.H, cs.W, cs.M, cs.KH, cs.KW, cs.ph, cs.pw, cs.group, 0, 0, 0, 0};
  g.OH = cs.H + 2 * cs.ph - cs.KH + 1;
  g.OW = cs.W + 2 * cs.pw - cs.KW + 1;
  g.Cg = cs.C / cs.group;
  g.Mg = cs.M / cs.group;
  g.density = 1.0f - cs.sparsity;
  // ncu = 1 (default): as on a full chip with a large batch, the fewest passes win; the cases
  // with ncu = 256 see a nearly empty chip and spread the channels over many workgroups
  Tiling t = choose_tiling(g, cs.waves, cs.lds, cs.ncu, use_jit);   // (one quad per lane: generated code only)
  if (!t.ok) { printf("tiling rejected\n"); return 2; }
  const int kdim = g.Cg * g.KH * g.KW;
  std::vector<float> w((size_t)g.M * kdim), x((size_t)g.N * g.C * g.H * g.W), bias(g.M);
  for (auto &v : w) { float r = frand(); v = (std::fabs(frand()) < cs.sparsity) ? 0.f : (r == 0 ? 0.5f : r); }
  for (auto &v : x) v = frand();
  for (auto &v : bias) v = 0.1f * frand();
  std::vector<std::vector<int>> rp(g.group), ci(g.group);
  std::vector<std::vector<float>> va(g.group);
  for (int cg = 0; cg < g.group; ++cg) {
    rp[cg].assign(g.Mg + 1, 0);
    for (int m = 0; m < g.Mg; ++m) {
      for (int j = 0; j < kdim; ++j) {
        float v = w[((size_t)cg * g.Mg + m) * kdim + j];
        if (v != 0) { va[cg].push_back(v); ci[cg].push_back(j); }
      }
      rp[cg][m + 1] = (int)ci[cg].size();
    }
  }
  WeightStream ws2 = build_stream(g, t, rp, ci, va);
  jit::Program jp;
  jit::DmaPlan jdma;
  bool jit_self_zero = false;
  if (use_jit) {
    jit::Options jo;
    jo.depth = 1 + (cs.N & 1);            // both read-ahead depths and both weight placements get exercised
    jo.hoist_weight = (cs.C >> 1) & 1;
    jo.prio_rows = (cs.M & 1) ? 2 : 0;
    jo.hi_sets = (cs.N & 1) ? 24 : 0;     // (used only by code without a tile B: deeper read-ahead through tile B's registers)
    jo.depth_one_tile = (cs.N & 1) ? 5 + cs.N % 9 : 5;
    jo.sweights = cs.KW != 1 && (cs.C & 3) != 1;
    jo.self_zero = (cs.M % 3) != 0;       // the code initialises its accumulators (most geometries; the kernel's own clearing stays covered)
    jit_self_zero = jo.self_zero != 0;     // weights through the scalar cache: most 3x3 / 5x5 geometries (not all: both forms stay covered)
    // plane DMA from inside the code wherever one wave owns an oc-group (whatever the table's size: the
    // product bounds it, the emulation does not need to)
    if (t.pix_waves == 1 && (t.waves == 8 || t.waves == 4)) {
      int padded = 0;
      jo.dma.period = jit::dma_period(t.plane_ch_floats / 4, 1 << 24, 0.0, &padded);
      jo.dma.on = jo.dma.period > 0 && padded == t.plane_ch_floats / 4;
      jo.dma.qpc = t.plane_ch_floats / 4;
      jo.dma.waves = t.waves;
      jo.dma.chan_bytes = (uint32_t)(cs.H * cs.W * 4);
      jo.dma.nt = (cs.N & 2) != 0;
      jo.dma.spread_pct = 40 + 10 * (cs.M % 5);
      jo.dma.ahead = (t.n_icb >= 2 && (cs.C & 1)) ? 2 : 1;      // two fills in flight: the unit of block k stages block k + 2
      // a tile's units as one chain (where every wave has an oc-group: build_program decides); off for some image sizes
      jo.chain.on = jo.dma.on && (cs.H % 3 != 0);
      jo.chain.nbuf = jo.dma.ahead + 1;
      jo.chain.buf_bytes = (uint32_t)((t.planes_bytes + 1023) / 1024 * 1024 + 1024);
    }
    jdma = jo.dma;
    jp = jit::build_program(g, t, rp, ci, va, jo);
    if (jp.overflow) { printf("jit: LDS offset overflow\n"); return 3; }
    if (jp.n_records != ws2.n_records) { printf("jit: %ld records, stream has %ld\n", jp.n_records, ws2.n_records); return 3; }
    ws2.chan = jp.chan;                   // (the generated code deals the channels on its own cost model)
  }
  // the slot -> channel table deals every channel of a conv group exactly once
  if (ws2.chan.size() != (size_t)g.group * t.n_ocg * t.G) { printf("chan table has the wrong size\n"); return 3; }
  for (int cgi = 0; cgi < g.group; ++cgi) {
    std::vector<int> seen(g.Mg, 0);
    for (int s = 0; s < t.n_ocg * t.G; ++s) {
      const uint32_t c = ws2.chan[(size_t)cgi * t.n_ocg * t.G + s];
      if ((int)c >= g.Mg) { printf("chan table entry out of range\n"); return 3; }
      if (s < g.Mg) seen[c]++;
    }
    for (int m = 0; m < g.Mg; ++m)
      if (seen[m] != 1) { printf("channel %d dealt %d times\n", m, seen[m]); return 3; }
  }
  if (!use_jit && stage_bytes_for(ws2.max_body_bytes) > 16384) { printf("staging area too large\n"); return 5; }

  // reference dense conv (double)
  std::vector<double> want((size_t)g.N * g.M * g.OH * g.OW, 0.0);
  for (int n = 0; n < g.N; ++n)
    for (int oc = 0; oc < g.M; ++oc) {
      const int cg = oc / g.Mg;
      for (int oh = 0; oh < g.OH; ++oh)
        for (int ow = 0; ow < g.OW; ++ow) {
          double s = bias[oc];
          for (int ic = 0; ic < g.Cg; ++ic)
            for (int kr = 0; kr < g.KH; ++kr)
              for (int kc = 0; kc < g.KW; ++kc) {
                const int ih = oh + kr - g.pad_h, iw = ow + kc - g.pad_w;
                if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) continue;
                s += (double)w[(size_t)oc * kdim + (ic * g.KH + kr) * g.KW + kc] *
                     x[(((size_t)n * g.C + cg * g.Cg + ic) * g.H + ih) * g.W + iw];
              }
          want[(((size_t)n * g.M + oc) * g.OH + oh) * g.OW + ow] = s;
        }
    }

  std::vector<JitPref> pref_of(use_jit ? jp.unit_off.size() : 0);
  std::vector<size_t> unit_end(use_jit ? jp.unit_off.size() : 0, 0);
  long dma_checked = 0;
  std::vector<float> got((size_t)g.N * g.M * g.OH * g.OW, -777.f);
  std::vector<int> written(got.size(), 0);
  const int n_tiles = t.band_mode ? g.N * t.bands : (g.N + t.nseg - 1) / t.nseg;
  std::vector<float> lds((size_t)t.icb * t.plane_ch_floats);
  for (int tile = 0; tile < n_tiles; ++tile)
    for (int cg = 0; cg < g.group; ++cg)
      for (int ocblk = 0; ocblk < t.n_ocblk; ++ocblk) {
        // per-wave accumulators: [wave][lane][192]
        // (Options::self_zero: nobody clears the accumulators for the code -- they start as NaN, and an FMA onto one is an error)
        const float acc0 = (use_jit && jit_self_zero) ? std::nanf("") : 0.f;
        std::vector<float> acc((size_t)t.waves * 64 * kAccAll, acc0);
        auto fill_block = [&](int blk, std::vector<float> &lds) {
          std::fill(lds.begin(), lds.end(), 0.f);
          for (int icl = 0; icl < t.icb; ++icl) {
            const int ic = blk * t.icb + icl;
            if (ic >= g.Cg) continue;
            for (int seg = 0; seg < t.nseg; ++seg) {
              int n, y0;
              if (t.band_mode) { n = tile / t.bands; y0 = (tile % t.bands) * t.tr; }
              else { n = tile * t.nseg + seg; y0 = 0; }
              if (n >= g.N) continue;
              for (int pr = 0; pr < t.plane_rows; ++pr) {
                const int yin = y0 + pr - g.pad_h;
                if (yin < 0 || yin >= t.H) continue;
                for (int xx = 0; xx < t.W; ++xx)
                  lds[(size_t)icl * t.plane_ch_floats + ((size_t)pr * t.nseg + seg) * t.RS + xx] =
                      x[(((size_t)n * g.C + cg * g.Cg + ic) * t.H + yin) * t.W + xx];
              }
            }
          }
        };
        // the synthetic quad tables of "this tile" and "the next tile" (chained units use both), period entries each
        auto make_tables = [&](uint32_t base, std::vector<uint32_t> &tab, std::vector<uint32_t> &tabx) {
          tab.assign(jdma.on ? jdma.qpc : 0, 0u);
          tabx.assign(jdma.on ? jdma.period : 0, 0u);
          for (size_t f = 0; f < tab.size(); ++f) tab[f] = (f % 7 == 3) ? 0xFFFFFFF0u : (uint32_t)(base + 16 * f);
          for (size_t e = 0; e < tabx.size(); ++e) {
            const size_t c = e / jdma.qpc, f = e % jdma.qpc;
            tabx[e] = tab[f] == 0xFFFFFFF0u ? 0xFFFFFFF0u : tab[f] + (uint32_t)(c * jdma.chan_bytes);
          }
        };
        // checks the pieces one unit issued: this wave's share of block `nb`'s image, every lane from the right
        // table entry and channel
        auto check_pieces = [&](const std::vector<DmaPiece> &issued, size_t first, size_t last, int ocg, int nb, uint32_t fill_base,
                                const std::vector<uint32_t> &tab) -> int {
          const int nch = std::min(t.icb, g.Cg - nb * t.icb);
          const long total = (long)nch * jdma.qpc;
          const int n_instr = (int)((total + 63) / 64);
          std::vector<int> seen(n_instr, 0);
          for (size_t pi = first; pi < last; ++pi) {
            const DmaPiece &pcs = issued[pi];
            const uint32_t rel = pcs.m0 - fill_base;
            if (rel % 1024 || (int)(rel / 1024) >= n_instr) { printf("jit dma: bad LDS address m0=%u fill_base=%u n_instr=%d nb=%d\n", pcs.m0, fill_base, n_instr, nb); return 3; }
            const int i = (int)(rel / 1024);
            if (i % jdma.waves != ocg % jdma.waves) { printf("jit dma: piece %d issued by wave %d\n", i, ocg % jdma.waves); return 3; }
            seen[i]++;
            if (pcs.nt != jdma.nt) { printf("jit dma: nt flag\n"); return 3; }
            for (int lane = 0; lane < 64; ++lane) {
              const long e = (long)i * 64 + lane;
              const bool on = (pcs.exec >> lane) & 1ull;
              if (on != (e < total)) { printf("jit dma: lane %d of piece %d: exec %d, in image %d\n", lane, i, (int)on, (int)(e < total)); return 3; }
              if (!on) continue;
              const long icl = e / jdma.qpc, f = e % jdma.qpc;
              if (tab[f] == 0xFFFFFFF0u) {
                if (pcs.voff[lane] != 0xFFFFFFF0u) { printf("jit dma: halo quad not marked\n"); return 3; }
              } else {
                const uint32_t want_g = tab[f] + (uint32_t)(((long)nb * t.icb + icl) * jdma.chan_bytes);
                if (pcs.voff[lane] == 0xFFFFFFF0u || pcs.voff[lane] + pcs.soff != want_g) { printf("jit dma: piece %d lane %d reads %u, want %u\n", i, lane, pcs.voff[lane] + pcs.soff, want_g); return 3; }
              }
            }
          }
          for (int i = ocg % 8; i < n_instr; i += 8)
            if (seen[i] != 1) { printf("jit dma: piece %d issued %d times\n", i, seen[i]); return 3; }
          return 0;
        };
        if (use_jit && jp.chained) {
          // ---- chained units: a wave runs the tile's blocks as ONE call ----
          std::vector<std::vector<float>> lds_of_block(t.n_icb, std::vector<float>((size_t)t.icb * t.plane_ch_floats));
          for (int blk = 0; blk < t.n_icb; ++blk) fill_block(blk, lds_of_block[blk]);
          const uint32_t buf_bytes = (uint32_t)((t.planes_bytes + 1023) / 1024 * 1024 + 1024);
          std::vector<uint32_t> tab_cur, tabx_cur, tab_nxt, tabx_nxt;
          make_tables(4096u, tab_cur, tabx_cur);
          make_tables(1u << 20, tab_nxt, tabx_nxt);
          for (int wave = 0; wave < t.waves; ++wave) {
            const int pw = wave % t.pix_waves, ow_ = wave / t.pix_waves;
            const int ocg = ocblk * t.oc_waves + ow_;
            if (ocg >= t.n_ocg) { printf("jit chain: a wave without an oc-group\n"); return 3; }
            JitWave jw;
            for (int lane = 0; lane < 64; ++lane)
              for (int r = 0; r < kAccAll; ++r) jw.v[(size_t)lane * 256 + 64 + r] = acc0;
            std::vector<uint32_t> laneA(64);
            for (int lane = 0; lane < 64; ++lane) {
              const int fr = (pw * t.tpl) * t.rows_per_slab + lane / t.S4;
              laneA[lane] = (uint32_t)(((size_t)fr * t.RS + 4 * (lane % t.S4)) * 4);
            }
            JitChainCtx cc;
            cc.lds_of_block = &lds_of_block;
            cc.nbuf = jdma.ahead + 1; cc.n_icb = t.n_icb; cc.ahead = jdma.ahead; cc.buf_bytes = buf_bytes;
            const int c_buf0 = (tile * 7 + cg) % cc.nbuf;            // wherever the rotation happens to stand at the tile's start
            cc.walk_base0 = (uint32_t)c_buf0 * buf_bytes;
            cc.fill_base0 = (uint32_t)((c_buf0 + cc.ahead) % cc.nbuf) * buf_bytes;
            JitDmaCtx dctx;
            dctx.tabx = &tabx_cur;
            dctx.tabx_next = &tabx_nxt;
            const size_t ui0 = ((size_t)cg * t.n_ocg + ocg) * t.n_icb;
            const int rc = jit_run_unit(jp.code, jp.unit_off[ui0], jw, lds_of_block[0], laneA, &dctx, nullptr, &cc);
            if (rc) return rc;
            if (cc.barriers != t.n_icb - 1 || (int)cc.unit_end.size() != t.n_icb) { printf("jit chain: %d barriers for %d blocks\n", cc.barriers, t.n_icb); return 3; }
            cc.unit_first_piece.push_back(dctx.issued.size());
            for (int blk = 0; blk < t.n_icb; ++blk) {
              const size_t ui = ui0 + blk;
              // where each unit starts: the chain must have passed through every unit's own entry
              if (blk > 0 && jp.unit_off[ui] < cc.unit_end[blk - 1]) { printf("jit chain: unit %d overlaps its predecessor\n", blk); return 3; }
              pref_of[ui] = cc.prefs[blk];
              unit_end[ui] = cc.unit_end[blk];
              const bool next_tile = blk + cc.ahead >= t.n_icb;
              const int nb = (blk + cc.ahead) % t.n_icb;
              const uint32_t fill_base = (uint32_t)((c_buf0 + blk + cc.ahead) % cc.nbuf) * buf_bytes;
              for (size_t pi = cc.unit_first_piece[blk]; pi < cc.unit_first_piece[blk + 1]; ++pi) {
                if (dctx.issued[pi].next_table != next_tile) { printf("jit chain: unit %d staged through the wrong tile's table\n", blk); return 3; }
                if (dctx.issued[pi].records != (next_tile ? cc.records_next : cc.records_cur)) { printf("jit chain: unit %d staged with the wrong record count\n", blk); return 3; }
              }
              const int rcp = check_pieces(dctx.issued, cc.unit_first_piece[blk], cc.unit_first_piece[blk + 1], ocg, nb, fill_base,
                                           next_tile ? tab_nxt : tab_cur);
              if (rcp) return rcp;
            }
            dma_checked += (long)dctx.issued.size();
            for (int lane = 0; lane < 64; ++lane)
              for (int r = 0; r < kAccAll; ++r) acc[((size_t)wave * 64 + lane) * kAccAll + r] = jw.v[(size_t)lane * 256 + 64 + r];
          }
        } else
        for (int blk = 0; blk < t.n_icb; ++blk) {
          // ---- fill ----
          fill_block(blk, lds);
          // ---- stream walk per wave ----
          for (int wave = 0; wave < t.waves; ++wave) {
            const int pw = wave % t.pix_waves, ow_ = wave / t.pix_waves;
            const int ocg = ocblk * t.oc_waves + ow_;
            if (ocg >= t.n_ocg) continue;
            if (use_jit) {
              // registers: the accumulators live in acc[] between blocks (192 per lane, tile A then B)
              JitWave jw;
              std::vector<uint32_t> laneA(64);
              for (int lane = 0; lane < 64; ++lane) {
                const int fr = (pw * t.tpl) * t.rows_per_slab + lane / t.S4;
                laneA[lane] = (uint32_t)(((size_t)fr * t.RS + 4 * (lane % t.S4)) * 4);
                if (t.tpl == 2 && t.rows_per_slab * t.RS * 4 != 1024) { printf("tile B is not tile A + 1 KiB\n"); return 3; }
                for (int r = 0; r < kAccAll; ++r) jw.v[(size_t)lane * 256 + 64 + r] = acc[((size_t)wave * 64 + lane) * kAccAll + r];
              }
              const size_t ui = ((size_t)cg * t.n_ocg + ocg) * t.n_icb + blk;
              if (jp.unit_off[ui] % jit::kUnitAlign) { printf("jit: unit not aligned\n"); return 3; }
              // the quad table of "the tile being staged": synthetic offsets, some of them the
              // out-of-range marker; the pieces the unit issues must cover this wave's share of block
              // blk + 1's image exactly, every lane from the right table entry and channel
              std::vector<uint32_t> tab(jdma.on ? jdma.qpc : 0), tabx(jdma.on ? jdma.period : 0);
              for (size_t f = 0; f < tab.size(); ++f) tab[f] = (f % 7 == 3) ? 0xFFFFFFF0u : (uint32_t)(4096 + 16 * f);
              for (size_t e = 0; e < tabx.size(); ++e) {
                const size_t c = e / jdma.qpc, f = e % jdma.qpc;
                tabx[e] = tab[f] == 0xFFFFFFF0u ? 0xFFFFFFF0u : tab[f] + (uint32_t)(c * jdma.chan_bytes);
              }
              JitDmaCtx dctx;
              dctx.tabx = &tabx;
              dctx.fill_base = 0x10000u * (unsigned)((blk + 1) & 1);
              JitPref pref;
              const int rc = jit_run_unit(jp.code, jp.unit_off[ui], jw, lds, laneA, jdma.on ? &dctx : nullptr, &pref);
              if (rc) return rc;
              pref_of[ui] = pref;
              unit_end[ui] = last_return_pc + 4;
              if (jdma.on) {
                const int nb = (blk + jdma.ahead) % t.n_icb;
                const int nch = std::min(t.icb, g.Cg - nb * t.icb);
                const long total = (long)nch * jdma.qpc;
                const int n_instr = (int)((total + 63) / 64);
                std::vector<int> seen(n_instr, 0);
                for (const DmaPiece &pcs : dctx.issued) {
                  const uint32_t rel = pcs.m0 - dctx.fill_base;
                  if (rel % 1024 || (int)(rel / 1024) >= n_instr) { printf("jit dma: bad LDS address m0=%u fill_base=%u n_instr=%d nb=%d\n", pcs.m0, dctx.fill_base, n_instr, nb); return 3; }
                  const int i = (int)(rel / 1024);
                  if (i % jdma.waves != ocg % jdma.waves) { printf("jit dma: piece %d issued by wave %d\n", i, ocg % jdma.waves); return 3; }
                  seen[i]++;
                  if (pcs.nt != jdma.nt) { printf("jit dma: nt flag\n"); return 3; }
                  for (int lane = 0; lane < 64; ++lane) {
                    const long e = (long)i * 64 + lane;
                    const bool on = (pcs.exec >> lane) & 1ull;
                    if (on != (e < total)) { printf("jit dma: lane %d of piece %d: exec %d, in image %d\n", lane, i, (int)on, (int)(e < total)); return 3; }
                    if (!on) continue;
                    const long icl = e / jdma.qpc, f = e % jdma.qpc;
                    if (tab[f] == 0xFFFFFFF0u) {
                      if (pcs.voff[lane] != 0xFFFFFFF0u) { printf("jit dma: halo quad not marked\n"); return 3; }
                    } else {
                      const uint32_t want_g = tab[f] + (uint32_t)(((long)nb * t.icb + icl) * jdma.chan_bytes);
                      if (pcs.voff[lane] == 0xFFFFFFF0u || pcs.voff[lane] + pcs.soff != want_g) { printf("jit dma: piece %d lane %d reads %u, want %u\n", i, lane, pcs.voff[lane] + pcs.soff, want_g); return 3; }
                    }
                  }
                }
                for (int i = ocg % 8; i < n_instr; i += 8)
                  if (seen[i] != 1) { printf("jit dma: piece %d issued %d times\n", i, seen[i]); return 3; }
                dma_checked += (long)dctx.issued.size();
              }
              for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < kAccAll; ++r) acc[((size_t)wave * 64 + lane) * kAccAll + r] = jw.v[(size_t)lane * 256 + 64 + r];
            } else {
              // the unit's body sits in the wave's staging area; row offset and first
              // accumulator of a group come from the previous group's meta word
              const size_t ui = ((size_t)cg * t.n_ocg + ocg) * t.n_icb + blk;
              const uint32_t *hdr = &ws2.unit_hdr[ui * kUnitHdrDwords];
              const uint32_t *body = &ws2.words[hdr[7] / 4];
              for (int d = 0; d < kUnitHdrDwords; ++d)
                if (body[d] != hdr[d]) { printf("body does not start with the header\n"); return 3; }
              body += kUnitHdrDwords;
              if (hdr[0] & 0x80) { printf("bit 7 of the lead word is not clear\n"); return 3; }
              uint32_t row_cur = (hdr[0] >> 8) & 0x7FF, ix0 = hdr[0] & 127, row_next = hdr[0] >> 21;
              size_t pos = 0;   // dword position in the body
              uint32_t k2 = 0;
              for (int n = kMaxSlots; n >= 1; --n) {
                const uint32_t end_n = hdr[1 + (kMaxSlots - n)];
                while (k2 != end_n) {
                  const uint32_t *q = body + pos;
                  const uint32_t row_off = row_cur * 32;
                  int idx[6];
                  idx[0] = (int)ix0;
                  idx[1] = (q[0] >> 7) & 127;
                  idx[2] = (q[0] >> 14) & 127;
                  if (n > 3) { idx[3] = q[4] & 127; idx[4] = (q[4] >> 7) & 127; idx[5] = (q[4] >> 14) & 127; }
                  for (int lane = 0; lane < 64; ++lane)
                    for (int tl = 0; tl < 2; ++tl) {
                      const int fr = (pw * 2 + tl) * t.rows_per_slab + lane / t.S4;
                      const int j = lane % t.S4;
                      const size_t base = (size_t)fr * t.RS + 4 * j;   // lanes = (row, segment, quad)
                      const size_t a = base + row_off / 4;
                      float *A = &acc[((size_t)wave * 64 + lane) * kAccAll + tl * kAccRegsPerTile];
                      for (int s = 0; s < n; ++s) {
                        float v;
                        std::memcpy(&v, &q[s < 3 ? 1 + s : 5 + (s - 3)], 4);
                        if (idx[s] % 4 != 0 || idx[s] + 3 >= kAccRegsPerTile) { printf("acc idx out of range\n"); return 3; }
                        for (int e = 0; e < 4; ++e) {
                          const float xv = (a + e < lds.size()) ? lds[a + e] : 0.f;
                          A[idx[s] + e] = std::fmaf(v, xv, A[idx[s] + e]);
                        }
                      }
                    }
                  row_cur = row_next;
                  row_next = q[0] >> 21;
                  ix0 = q[0] & 127;
                  pos += n > 3 ? 8 : 4;
                  ++k2;
                }
              }
              if (row_cur != 0 || row_next != 0 || ix0 != 0) { printf("leads past the last group are not empty\n"); return 3; }
              if ((int)(pos * 4) + 4 * kUnitHdrDwords > ws2.max_body_bytes) { printf("body longer than max_body_bytes\n"); return 3; }
            }
          }
        }
        // ---- epilogue ----
        for (int wave = 0; wave < t.waves; ++wave) {
          const int pw = wave % t.pix_waves, ow_ = wave / t.pix_waves;
          const int ocg = ocblk * t.oc_waves + ow_;
          if (ocg >= t.n_ocg) continue;
          for (int gl = 0; gl < t.G; ++gl) {
            if (ocg * t.G + gl >= g.Mg) break;
            const int m = (int)ws2.chan[((size_t)cg * t.n_ocg + ocg) * t.G + gl];   // channel in this slot
            const int oc = cg * g.Mg + m;
            // (one quad per lane: slot gl >= 24 lives in what would be tile B's registers -- the
            // accumulator file is one run of 192 registers, indexed 4 * (gl * KW + kc) either way)
            for (int tl = 0; tl < t.tpl; ++tl)
              for (int lane = 0; lane < 64; ++lane) {
                const int fr = (pw * t.tpl + tl) * t.rows_per_slab + lane / t.S4;
                const int yl = fr / t.nseg, seg = fr % t.nseg, j = lane % t.S4;
                if (yl >= t.tr) continue;
                int n, y;
                if (t.band_mode) { n = tile / t.bands; y = (tile % t.bands) * t.tr + yl; }
                else { n = tile * t.nseg + seg; y = yl; }
                if (n >= g.N || y >= t.OH) continue;
                for (int e = 0; e < 4; ++e) {
                  const int xo = 4 * j + e;
                  if (xo >= t.OW) continue;
                  float sum = 0.f;
                  for (int kc = 0; kc < g.KW; ++kc) {
                    const int pos = e + kc - g.pad_w;   // position relative to own quad
                    int src_lane = lane, el = pos;
                    if (pos < 0) { if (j == 0) continue; src_lane = lane - 1; el = pos + 4; }
                    else if (pos > 3) { if (j == t.S4 - 1) continue; src_lane = lane + 1; el = pos - 4; }
                    sum += acc[((size_t)wave * 64 + src_lane) * kAccAll + tl * kAccRegsPerTile +
                               4 * (gl * g.KW + kc) + el];
                  }
                  sum += bias[oc];
                  const size_t o = (((size_t)n * g.M + oc) * t.OH + y) * t.OW + xo;
                  got[o] = sum;
                  written[o]++;
                }
              }
          }
        }
      }
  // every unit touched the code of the unit its wave runs next, from its first byte to its return
  if (use_jit)
    for (size_t ui = 0; ui < jp.unit_off.size(); ++ui) {
      if (!unit_end[ui]) continue;     // (never run: an oc-group slot past the last channel)
      const size_t blk = ui % t.n_icb, nxt = ui - blk + (blk + 1) % t.n_icb;
      const JitPref &pf = pref_of[ui];
      if (pf.touches < 1 || pf.base != (long long)jp.unit_off[nxt] || (size_t)pf.base + 4096ull * pf.touches < unit_end[nxt]) {
        printf("jit: code touches %lld+%d do not cover the next unit [%u, %zu)\n", pf.base, pf.touches, jp.unit_off[nxt], unit_end[nxt]);
        return 3;
      }
    }
  double maxerr = 0, maxref = 0;
  for (size_t i = 0; i < got.size(); ++i) {
    if (written[i] != 1) { printf("output %zu written %d times\n", i, written[i]); return 4; }
    maxerr = std::fmax(maxerr, std::fabs(got[i] - want[i]));
    maxref = std::fmax(maxref, std::fabs(want[i]));
  }
  const double rel = maxerr / std::fmax(1e-6, maxref);
  printf("%s%sN%d C%d %dx%d M%d K%dx%d p%d,%d g%d sp%.2f waves%d: tpl=%d S4=%d G=%d ocw=%d pw=%d tr=%d nseg=%d band=%d icb=%d/%d lds=%d "
         "groups=%ld recs=%ld recs/group=%.2f dma=%ld lines=%ld init=%ld+%ld rel_err=%.2e\n",
         use_jit ? "jit " : "", use_jit && jp.chained ? "chained " : "", cs.N, cs.C, cs.H, cs.W, cs.M, cs.KH, cs.KW, cs.ph, cs.pw, cs.group, cs.sparsity, cs.waves, t.tpl, t.S4, t.G,
         t.oc_waves, t.pix_waves, t.tr, t.nseg, (int)t.band_mode, t.icb, t.n_icb, t.planes_bytes,
         ws2.n_groups, ws2.n_records, ws2.n_groups ? (double)ws2.n_records / (double)ws2.n_groups : 0.0, dma_checked, g_weight_lines, g_first_products, g_cleared, rel);
  g_weight_lines = g_first_products = g_cleared = 0;
  if (use_jit && jdma.on && dma_checked == 0) { printf("jit dma: nothing was checked\n"); return 3; }
  return rel <= 1e-5 ? 0 : 1;
}

int main() {
  const Case cases[] = {
      {3, 8, 7, 7, 40, 3, 3, 1, 1, 1, 0.9f, 8, 65536},      // res5-like: whole images per WG
      {2, 16, 14, 14, 24, 3, 3, 1, 1, 1, 0.9f, 8, 65536},   // res4-like
      {2, 6, 28, 28, 20, 3, 3, 1, 1, 1, 0.8f, 8, 8192},     // res3-like, several ic blocks
      {2, 5, 56, 56, 16, 3, 3, 1, 1, 1, 0.9f, 8, 65536},    // res2-like: band mode
      {2, 5, 56, 56, 70, 3, 3, 1, 1, 1, 0.9f, 8, 65536},    // 8 waves, oc tail
      {2, 8, 27, 27, 16, 5, 5, 2, 2, 2, 0.8f, 8, 65536},    // alex conv2-like: 5x5, groups
      {3, 12, 13, 13, 20, 3, 3, 1, 1, 2, 0.8f, 8, 65536},   // alex conv4-like
      {2, 20, 12, 12, 50, 5, 5, 0, 0, 1, 0.5f, 8, 65536},   // lenet conv2: valid 5x5
      {2, 24, 28, 28, 33, 1, 1, 0, 0, 1, 0.95f, 8, 65536},  // googlenet 1x1
      {1, 3, 20, 20, 8, 3, 3, 2, 2, 1, 0.5f, 8, 65536},     // pad 2 with 3x3 (OH > H)
      {2, 4, 9, 70, 8, 3, 3, 1, 1, 1, 0.6f, 8, 65536},      // wide: S4 = 32
      {1, 4, 5, 200, 4, 3, 1, 1, 0, 1, 0.5f, 8, 65536},     // KW = 1 with KH = 3, S4 = 64
      {2, 4, 6, 6, 4, 2, 2, 1, 1, 1, 0.3f, 8, 65536},       // even kernel
      {1, 2, 4, 4, 3, 3, 3, 1, 1, 1, 0.0f, 8, 65536},       // dense tiny
      {1, 2, 4, 4, 3, 3, 3, 1, 1, 1, 1.0f, 8, 65536},       // all pruned
      {5, 64, 7, 7, 48, 3, 3, 1, 1, 1, 0.5f, 8, 65536},     // dense-ish rows: groups split at 7 slots
      {7, 40, 14, 14, 16, 1, 1, 0, 0, 1, 0.9f, 8, 65536},   // pointwise, W % 4 != 0: re-cut to 196 x 1
      {5, 30, 7, 7, 48, 1, 1, 0, 0, 1, 0.9f, 8, 65536},     // pointwise 7x7 -> 49 x 1, few channels/wave
      {300, 6, 7, 7, 12, 1, 1, 0, 0, 1, 0.8f, 8, 65536},    // batch > CUs: several images per workgroup
      {3, 6, 13, 13, 10, 1, 1, 0, 0, 2, 0.7f, 8, 65536},    // pointwise 13x13 (169 = 13^2), groups
      {3, 8, 21, 6, 7, 5, 5, 4, 4, 1, 0.9f, 8, 65536},      // pad 4 with 5x5: OW = 10 > W = 6
      {2, 4, 5, 7, 6, 3, 3, 2, 2, 1, 0.5f, 8, 65536},       // pad 2 with 3x3: OW = 9 > RS(W) = 8
      {2, 5, 56, 56, 70, 3, 3, 1, 1, 1, 0.9f, 8, 65536, 256},   // empty chip: one channel per wave, 9 passes
      {5, 30, 7, 7, 48, 1, 1, 0, 0, 1, 0.9f, 8, 65536, 256},    // empty chip, pointwise
      {40, 16, 7, 7, 64, 1, 1, 0, 0, 1, 0.9f, 8, 65536, 8},     // 8 CUs: images per workgroup vs passes
      {1, 48, 56, 56, 64, 1, 1, 0, 0, 1, 0.97f, 8, 65536, 256}, // one channel per wave, units of 0-2 rows: the code's own plane DMA goes out in the tail
      {32, 300, 14, 14, 256, 1, 1, 0, 0, 1, 0.95f, 8, 65536, 32},    // 256 output channels, one image per tile: generated code takes one quad per lane, 32 channels per wave
      {16, 400, 7, 7, 384, 1, 1, 0, 0, 1, 0.97f, 8, 65536, 16},    // 384: 48 channels per wave, output rows that are not whole quads
      {16, 10, 4, 4, 300, 1, 1, 0, 0, 1, 0.8f, 8, 65536, 16},    // 300: a ragged last slot range (38 channels per wave, 4 in the last)
      // chained units over many blocks (small plane budgets): one and two fills in flight (odd C: two), 3x3 and pointwise,
      // whole images and bands, a tile count that leaves the buffer rotation at every phase
      {2, 21, 14, 14, 16, 3, 3, 1, 1, 1, 0.8f, 8, 16384},
      {3, 24, 28, 28, 64, 3, 3, 1, 1, 1, 0.9f, 8, 16384},
      {5, 33, 14, 14, 32, 1, 1, 0, 0, 1, 0.9f, 8, 8192},
      {4, 45, 7, 7, 24, 3, 3, 1, 1, 1, 0.85f, 8, 8192},
      {3, 20, 56, 56, 8, 3, 3, 1, 1, 1, 0.9f, 8, 32768},
      {9, 37, 7, 7, 64, 1, 1, 0, 0, 1, 0.9f, 8, 4096, 256},
      // two 4-wave workgroups per CU (sconv_tiled.hip, half-workgroup rule): the pieces of a block go round four waves
      {4, 48, 28, 28, 64, 1, 1, 0, 0, 1, 0.95f, 4, 32768, 4},
      {3, 40, 28, 28, 96, 1, 1, 0, 0, 1, 0.9f, 4, 8192, 2},
      {2, 21, 14, 14, 16, 3, 3, 1, 1, 1, 0.8f, 4, 16384},
      {3, 30, 28, 28, 128, 1, 1, 0, 0, 1, 0.9f, 4, 32768, 1},     // ... 128 channels: two columns of four waves
      {2, 26, 28, 28, 176, 1, 1, 0, 0, 1, 0.93f, 4, 8192, 1},     // ... 176 channels, several blocks
  };
  int bad = 0;
  for (const Case &c : cases) {
    bad += run(c, false) != 0;
    bad += run(c, true) != 0;     // the same geometry through the generated code
  }
  printf(bad ? "FAILED %d case(s)\n" : "all cases OK\n", bad);
  return bad ? 1 : 0;
}
