// jit_codegen.cpp -- see jit_codegen.h
#include "jit_codegen.h"

#include "knobs.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "parallel_for.h"
#include <thread>

namespace escoin {
namespace jit {

static long gcd_l(long a, long b) { return b ? gcd_l(b, a % b) : a; }

int dma_period(int qpc, int max_period, double slack, int *padded) {
  int best = 0;
  *padded = 0;
  for (int q = qpc; q <= (int)(qpc * (1.0 + slack)); ++q) {
    const long l = (long)q / gcd_l(q, 64) * 64;
    if (l <= max_period && (best == 0 || l < best)) {
      best = (int)l;
      *padded = q;
    }
    if (best && q == qpc) break;      // the plane as it is will do
  }
  return best;
}

Options options_from_env() {
  Options o;
  // (all compile-time constants in the product build: knobs.h)
  o.depth = std::max(1, std::min(2, (int)ESC_KNOB("JIT_DEPTH", o.depth)));
  o.depth_one_tile = std::max(1, std::min(13, (int)ESC_KNOB("JIT_DEPTH1", o.depth_one_tile)));
  o.hi_sets = std::max(0, std::min(24, (int)ESC_KNOB("JIT_HI_SETS", o.hi_sets)));
  o.hoist_weight = ESC_KNOB("JIT_HOIST", o.hoist_weight) != 0;
  o.prio_rows = std::max(0, (int)ESC_KNOB("JIT_PRIO_ROWS", o.prio_rows));
  o.prio_waves = std::max(0, (int)ESC_KNOB("JIT_PRIO_WAVES", o.prio_waves));
  o.ablate = ESC_ABL_KNOB("JIT_ABL");
  o.prefetch = ESC_KNOB("JIT_PREFETCH", o.prefetch) != 0;
  if (ESC_KNOB_SET("JIT_ONE_TILE")) o.one_tile = ESC_KNOB("JIT_ONE_TILE", 1) != 0 ? 0 : -1;
  return o;
}

namespace {

struct Rec {
  uint32_t bits;   // the weight
  int idx;         // accumulator quad: gl * KW + kc
};
struct Row {
  uint32_t lds_off;
  std::vector<Rec> recs;
};

struct Piece {
  uint32_t tab_off;    // byte offset of the lane-0 table entry from v34
  uint32_t lds_off;    // LDS byte offset of the piece from the fill buffer's base
  uint32_t soff;       // scalar offset: the channel group's bytes from the conv group's channel 0
  int lanes;           // lanes that carry a quad of the block image (64 but for a block's last piece)
};

// The walk of one unit.  Rows are read `depth` ahead into the three input sets in rotation; LDS
// returns data in order, so every wait is a count: "all but the N operations issued after the one
// I need" (Lds below keeps the issue order).  A set is overwritten by the read issued two rows after
// the one that used it: its FMAs were issued (in order) before that read was.
// `pieces`: this wave's LDS-DMA instructions for the block staged next.  A piece's table entry is
// read at the top of one row and the instruction goes out after the wait of the next row.
struct Lds {
  std::vector<uint32_t> &c;
  int issued = 0, done = 0;     // operations issued so far; operations known to have completed
  int ablate;
  int issue() { return issued++; }
  void wait_for(int id) {       // operation `id` (0-based issue order) must have landed
    if (id < done) return;
    if (ablate & 64) { done = id + 1; return; }      // (timing only: nobody waits for LDS)
    enc_waitcnt_lgkm(c, std::min(15, issued - 1 - id));
    done = std::max(done, issued - std::min(15, issued - 1 - id));
  }
};

// n_pref > 0: the unit starts with n_pref loads over the next unit's code; returns the index (in
// `c`) of the distance literal to patch (0: none).
// blk / n_icb: which block of its oc-group's chain this unit is (chaining only).
// n_idx: accumulator quads per tile the epilogue reads (channels per wave x kernel columns): with Options::self_zero
// block 0's unit leaves every one of them initialised.
size_t emit_unit(std::vector<uint32_t> &c, const std::vector<Row> &rows, const std::vector<Piece> &pieces,
                 const Options &opt_in, int n_pref, int wave, int blk, int n_icb, int n_idx) {
  Options opt = opt_in;
  if (opt.prio_waves > 0 && (wave < 0 || wave >= opt.prio_waves)) opt.prio_rows = 0;
  size_t patch = 0;
  if (n_pref > 0) {
    enc_getpc(c, kSPref);                    // s[50:51] = address of the instruction after this one
    patch = c.size() + 1;
    enc_s_add_lit(c, kSPref, 0u);            // + (next unit - that address): patched
    enc_s_addc(c, kSPref + 1, false);        // (patched to -1 for a negative distance)
    for (int k = 0; k < n_pref; ++k) {
      enc_global_load_dword(c, kVPrefDead, kVPrefLane, kSPref);
      if (k + 1 < n_pref) {
        enc_s_add_lit(c, kSPref, 4096u);
        enc_s_addc(c, kSPref + 1, false);
      }
    }
  }
  if (opt.chain.on && blk > 0 && blk == n_icb - opt.dma.ahead) {
    // from this unit on the pieces belong to the NEXT tile: its quad table, its record count
    enc_s_mov(c, kSRecords, kSNextRecords);
    enc_v_add_u32_s(c, kVTabAddr, kSTabDelta, kVTabAddr);
  }
  const int n = (opt.ablate & 8) ? 0 : (int)rows.size();
  // (self_zero) which accumulator quads this unit still has to initialise: all of them in block 0's unit
  const bool init_acc = opt.self_zero && blk == 0 && !(opt.ablate & 1);
  std::vector<char> fresh(init_acc ? (size_t)n_idx : 0, 1);
  if (init_acc) {
    std::vector<char> touched((size_t)n_idx, 0);
    for (int k = 0; k < n; ++k)
      for (const Rec &r : rows[k].recs) touched[(size_t)r.idx] = 1;
    for (int i = 0; i < n_idx; ++i)
      if (!touched[(size_t)i]) {
        enc_pk_zero(c, kAccA + 4 * i);
        enc_pk_zero(c, kAccA + 4 * i + 2);
        if (!opt.one_tile) {
          enc_pk_zero(c, kAccB + 4 * i);
          enc_pk_zero(c, kAccB + 4 * i + 2);
        }
        fresh[(size_t)i] = 0;
      }
  }
  // Weights through the scalar cache (Options::sweights): which line and slot every nonzero of the unit's walk takes.
  // A line ends where the next ROW would not fit (so that the switch to a line sits at a row top, where the wave waits
  // for LDS anyway); a row of more than 16 nonzeros runs over several lines.
  bool sw = opt.sweights && !opt.ablate;
  std::vector<int> line_of, slot_of;
  int n_lines = 0;
  size_t sw_patch = 0;          // index (in `c`) of the literal that carries the distance to the unit's weight lines
  size_t sw_from = 0;           // byte address (in `c`) s_getpc_b64 returns
  if (sw) {
    int fill = kSWLine;         // (forces a new line at the first nonzero)
    for (int k = 0; k < n; ++k) {
      const int rn = (int)rows[k].recs.size();
      if (fill + rn > kSWLine && rn <= kSWLine && fill > 0) fill = kSWLine;     // the row starts a new line
      for (int r = 0; r < rn; ++r) {
        if (fill == kSWLine) { ++n_lines; fill = 0; }
        line_of.push_back(n_lines - 1);
        slot_of.push_back(fill++);
      }
    }
    if ((size_t)n_lines * kSWLine > 30000) {
      // (the branch over the lines has a 16-bit word offset: a unit of more than ~30 000 nonzeros -- a whole layer of tiny
      //  images in one block -- keeps the literal moves; units of both kinds may follow each other)
      sw = false;
      n_lines = 0;
      line_of.clear();
      slot_of.clear();
    }
    if (n_lines > 0) {
      enc_getpc(c, kSWBase);
      sw_from = c.size() * 4;
      sw_patch = c.size() + 1;
      enc_s_add_lit(c, kSWBase, 0u);           // + (first weight line - that address): patched below
      enc_s_addc(c, kSWBase + 1, false);
      enc_s_load_x16(c, kSWBuf0, kSWBase, 0u);
    }
  }
  // Without a tile B the 24 input registers hold SIX quads instead of three pairs: rows are read five
  // ahead.  A row of such a layer (pointwise, 95 % sparse) carries one or two nonzeros -- 10-25 cycles of
  // FMAs -- and an LDS read takes well over a hundred to land: two rows of read-ahead left the walk
  // waiting for LDS latency on every row.
  // ... and where tile B exists but holds no rows (a small image walked one or a few to a workgroup), its 96
  // accumulator registers are free as well: 24 more quads (opt.hi_sets), rows read up to thirteen ahead -- LDS
  // counts at most 15 operations in flight, and the plane DMA's table reads share that count
  const int n_sets = opt.one_tile > 0 ? 2 * kInSets + opt.hi_sets : kInSets;
  const int set_regs = opt.one_tile > 0 ? 4 : 8;
  const int depth = opt.one_tile > 0 ? std::max(1, std::min(opt.depth_one_tile, std::min(13, n_sets - 1))) : opt.depth;
  auto set_base = [&](int k) {
    const int sidx = k % n_sets;
    return opt.one_tile > 0 && sidx >= 2 * kInSets ? kAccB + 4 * (sidx - 2 * kInSets) : kVIn0 + set_regs * sidx;
  };
  Lds lds{c, 0, 0, opt.ablate};
  std::vector<int> row_id(n, -1);       // issue id of a row's second read
  auto issue = [&](int k) {
    if (opt.ablate & 2) return;
    const int base = set_base(k);
    enc_ds_read_b128(c, base, kVAddrA, rows[k].lds_off);
    lds.issue();
    if (opt.one_tile) { row_id[k] = lds.issued - 1; return; }
    enc_ds_read_b128(c, base + 4, kVAddrB, rows[k].lds_off);
    row_id[k] = lds.issue();
  };
  // pieces: table read at row t_row[p], instruction after the wait of row t_row[p] + 1
  const int np = (int)pieces.size();
  std::vector<int> t_row(np, 0), t_id(np, -1);
  const int span = std::max(1, n * opt.dma.spread_pct / 100);
  for (int p = 0; p < np; ++p) t_row[p] = np ? (int)((long)p * span / np) : 0;
  int next_read = 0, next_issue = 0;
  // table entries rotate through three registers; an entry may only be read into a register whose
  // previous piece has gone out, so at most three are in flight
  constexpr int kTabRegs = 3;
  auto tab_reg = [](int p) {
    static const int r[kTabRegs] = {kVTab0, kVTab1, kVTab1 + 1};
    return r[p % kTabRegs];
  };
  auto read_tables = [&](int row) {     // the table entries of the pieces scheduled up to `row` (all: row < 0)
    while (next_read < np && (row < 0 || t_row[next_read] <= row) && next_read - next_issue < kTabRegs) {
      enc_ds_read_b32(c, tab_reg(next_read), kVTabAddr, pieces[next_read].tab_off);
      t_id[next_read] = lds.issue();
      ++next_read;
    }
  };
  auto issue_pieces = [&](int row) {    // ... and the instructions of the pieces read at rows < `row` (all read: row < 0)
    while (next_issue < next_read && (row < 0 || t_row[next_issue] < row)) {
      const Piece &pc = pieces[next_issue];
      lds.wait_for(t_id[next_issue]);
      enc_s_add_m0_lit(c, kSFillBase, pc.lds_off);
      enc_s_mov_lit(c, kSSoff, pc.soff);
      if (pc.lanes < 64) {
        const unsigned long long m = pc.lanes >= 64 ? ~0ull : ((1ull << pc.lanes) - 1);
        enc_s_mov_lit(c, kSExecLo, (uint32_t)m);
        enc_s_mov_lit(c, kSExecHi, (uint32_t)(m >> 32));
      }
      // (M0 written by the scalar unit must be at least one wait state old when an LDS-DMA
      // instruction uses it, as in the hand-written sites: s_mov m0 / s_nop / buffer_load)
      enc_nop(c);
      enc_lds_dma16(c, tab_reg(next_issue), kSRsrc, kSSoff, opt.dma.nt);
      if (pc.lanes < 64) enc_exec_all(c);
      ++next_issue;
    }
  };
  for (int k = 0; k < std::min(depth, n); ++k) issue(k);
  // weights alternate between two SGPR pairs; with hoisting, the s_mov of the NEXT record (of this
  // row or the next one) sits in front of the current record's FMAs: the scalar write is long done
  // when the vector unit reads it
  std::vector<const Rec *> flat;
  std::vector<int> first_of_row(n + 1, 0);
  for (int k = 0; k < n; ++k) {
    first_of_row[k] = (int)flat.size();
    for (const Rec &r : rows[k].recs) flat.push_back(&r);
  }
  first_of_row[n] = (int)flat.size();
  auto sreg = [](int j) { return (j & 1) ? kSWeight1 : kSWeight0; };
  if (!sw && opt.hoist_weight && !flat.empty() && !(opt.ablate & 4)) enc_s_mov_lit(c, sreg(0), flat[0]->bits);
  if ((opt.ablate & 4) && (opt.ablate & 16384)) {
    // no weight moves, but two nonzero weights in the registers: the FMAs do real arithmetic.  (With whatever the
    // registers held -- zeros -- the chip draws less power and clocks higher: bit 2 alone overstates the moves' cost
    // by a factor of three, profiles/r04_walk_limits.md)
    enc_s_mov_lit(c, kSWeight0, 0x3F9E3779u);
    enc_s_mov_lit(c, kSWeight1, 0xBF4A7B2Du);
  }
  int prio = 0;
  // (sweights) the switch to weight line L: everything outstanding lands -- the line, loaded a line ago, and the LDS
  // reads in flight --, then the line after it is requested into the other buffer
  auto switch_line = [&](int L) {
    enc_waitcnt_lgkm(c, 0);
    lds.done = lds.issued;
    if (L + 1 < n_lines) enc_s_load_x16(c, ((L + 1) & 1) ? kSWBuf1 : kSWBuf0, kSWBase, (uint32_t)(L + 1) * 64u);
  };
  for (int k = 0; k < n; ++k) {
    read_tables(k);
    const int j0 = first_of_row[k];
    const bool row_switches = sw && j0 < first_of_row[k + 1] && slot_of[j0] == 0;
    if (row_switches) switch_line(line_of[j0]);       // in FRONT of the read-ahead: the wait then covers reads a row old
    if (k + depth < n) issue(k + depth);
    if (opt.prio_rows > 0 && k % opt.prio_rows == 0) {
      prio ^= 1;
      enc_setprio(c, prio);
    }
    if (opt.ablate & 2) enc_waitcnt_lgkm(c, 0);
    else lds.wait_for(row_id[k]);
    issue_pieces(k);
    const int xa = set_base(k), xb = xa + 4;
    for (int j = first_of_row[k]; j < first_of_row[k + 1]; ++j) {
      if (sw) {
        if (j > j0 && slot_of[j] == 0) switch_line(line_of[j]);     // (a row of more than 16 nonzeros)
      } else if (opt.ablate & 4) {
      } else if (opt.hoist_weight) {
        if (j + 1 < (int)flat.size()) enc_s_mov_lit(c, sreg(j + 1), flat[j + 1]->bits);
      } else {
        enc_s_mov_lit(c, sreg(j), flat[j]->bits);
      }
      if (opt.ablate & 1) continue;
      const int a = 4 * flat[j]->idx;
      // (timing only: 512 = an s_nop behind every FMA -- four more instructions and 16 more bytes per nonzero)
      const bool first = init_acc && fresh[(size_t)flat[j]->idx];     // the quad's first product: multiply, do not accumulate
      auto fma = [&](int acc, int x) {
        if (sw) {
          const int pair = ((line_of[j] & 1) ? kSWBuf1 : kSWBuf0) + (slot_of[j] & ~1);
          if (first) {
            if (slot_of[j] & 1) enc_pk_mul_hi(c, acc, pair, x);
            else enc_pk_mul(c, acc, pair, x);
          } else if (slot_of[j] & 1) enc_pk_fma_hi(c, acc, pair, x);
          else enc_pk_fma(c, acc, pair, x);
          return;
        }
        if (first) enc_pk_mul(c, acc, sreg(j), x);
        else enc_pk_fma(c, acc, sreg(j), x);
        if (opt.ablate & 512) enc_nop(c);
      };
      fma(kAccA + a, xa);
      fma(kAccA + a + 2, xa + 2);
      if (!opt.one_tile) {
        fma(kAccB + a, xb);
        fma(kAccB + a + 2, xb + 2);
      }
      if (first) fresh[(size_t)flat[j]->idx] = 0;
    }
  }
  // whatever the rows did not take (short or empty units): two at a time
  while (next_issue < np) {
    read_tables(-1);
    issue_pieces(-1);
  }
  if (opt.prio_rows > 0 && prio) enc_setprio(c, 0);
  if (opt.chain.on && blk + 1 < n_icb) {
    // on to the next block without leaving the code.  Everything this wave issued for the block about to be
    // walked must have landed: with one fill in flight that is all of it; with two, all but this unit's own
    // vector-memory operations (its code touches and its pieces, which stage the block after the next).
    const int younger = opt.dma.ahead >= 2 ? n_pref + (int)pieces.size() : 0;
    if (!(opt.ablate & 32)) enc_waitcnt_vm(c, std::min(63, younger));
    if (!(opt.ablate & 16)) enc_barrier(c);
    const uint32_t all = (uint32_t)opt.chain.nbuf * opt.chain.buf_bytes;
    enc_s_mov(c, kSChainTmp, kSWalkBase);
    enc_s_add_u32_lit(c, kSWalkBase, kSWalkBase, opt.chain.buf_bytes);
    enc_s_cmp_lt_u32_lit(c, kSWalkBase, all);
    enc_s_cselect_or_zero(c, kSWalkBase, kSWalkBase);
    enc_s_sub_u32(c, kSChainTmp, kSWalkBase, kSChainTmp);          // (mod 2^32: also when the buffer wraps)
    enc_v_add_u32_s(c, kVAddrA, kSChainTmp, kVAddrA);
    if (!opt.one_tile) enc_v_add_u32_s(c, kVAddrB, kSChainTmp, kVAddrB);
    enc_s_add_u32_lit(c, kSFillBase, kSFillBase, opt.chain.buf_bytes);
    enc_s_cmp_lt_u32_lit(c, kSFillBase, all);
    enc_s_cselect_or_zero(c, kSFillBase, kSFillBase);
    if (sw && n_lines > 0) {
      // over the weight lines into the next unit (which starts on the 64-byte boundary behind them)
      const size_t at = c.size();
      enc_s_branch(c, 0);
      while ((c.size() * 4) % 64) enc_nop(c);
      c[at] = 0xBF820000u | (uint32_t)((c.size() - at - 1 + (size_t)n_lines * 16) & 0xFFFFu);
    }
  } else {
    enc_setpc_return(c);
  }
  if (sw && n_lines > 0) {
    // the unit's weight lines: 16 per 64-byte line in walk order, zero padded; never executed
    while ((c.size() * 4) % 64) enc_nop(c);
    const long long dist = (long long)(c.size() * 4) - (long long)sw_from;
    c[sw_patch] = (uint32_t)dist;
    std::vector<uint32_t> lines((size_t)n_lines * kSWLine, 0u);
    for (size_t j = 0; j < flat.size(); ++j) lines[(size_t)line_of[j] * kSWLine + slot_of[j]] = flat[j]->bits;
    c.insert(c.end(), lines.begin(), lines.end());
  }
  return patch;       // (chained: the next unit follows; alignment padding is s_nop)
}

}  // namespace

namespace {
// The units of one (conv group, oc-group) chain, as a blob of its own: unit offsets relative to its first byte, the
// code touches' distances (all inside the chain: a unit touches its successor, the last one the first) patched in.
struct ChainOut {
  std::vector<uint32_t> code;
  std::vector<uint32_t> off;        // [n_icb] byte offset of each unit's entry from the chain's first byte
  long n_rows = 0, n_records = 0, n_dma = 0;
  bool overflow = false;
  size_t max_unit = 0;
  std::vector<double> blk_cost;     // [n_icb] instructions of the walk per block: 3 per nonempty row + 5 per nonzero
};

void emit_chain(const ConvGeom &g, const Tiling &t, const std::vector<int> &rowptr, const std::vector<int> &colidx,
                const std::vector<float> &values, const Options &opt, int n_pref, const uint32_t *chan, int ocg, ChainOut *out) {
  ChainOut &p = *out;
  const int rows_per_blk = t.icb * g.KH;
  std::vector<Row> rows(rows_per_blk), live;
  std::vector<Piece> pieces;
  std::vector<size_t> patches;
  p.off.assign(t.n_icb, 0u);
  for (int blk = 0; blk < t.n_icb; ++blk) {
    for (auto &r : rows) r.recs.clear();
    const int ic_lo = blk * t.icb, ic_hi = std::min(g.Cg, ic_lo + t.icb);
    for (int gl = 0; gl < t.G; ++gl) {
      if (ocg * t.G + gl >= g.Mg) break;
      const int m = (int)chan[gl];
      // (columns ascend within a CSR row: the block's nonzeros are one contiguous run)
      const int *b = colidx.data() + rowptr[m], *e = colidx.data() + rowptr[m + 1];
      const int *lo = std::lower_bound(b, e, ic_lo * g.KH * g.KW), *hi = std::lower_bound(lo, e, ic_hi * g.KH * g.KW);
      for (const int *cp = lo; cp < hi; ++cp) {
        const int col = *cp;
        const int kc = col % g.KW, kr = (col / g.KW) % g.KH, ic = col / (g.KW * g.KH);
        Rec rec;
        std::memcpy(&rec.bits, &values[(size_t)(cp - colidx.data())], 4);
        rec.idx = gl * g.KW + kc;
        rows[(ic - ic_lo) * g.KH + kr].recs.push_back(rec);
      }
    }
    live.clear();
    for (int r = 0; r < rows_per_blk; ++r) {
      if (rows[r].recs.empty()) continue;
      const int icl = r / g.KH, kr = r % g.KH;
      const size_t off = ((size_t)icl * t.plane_ch_floats + (size_t)kr * t.nseg * t.RS) * 4;
      if (off > 0xFFF0u) p.overflow = true;
      Row row;
      row.lds_off = (uint32_t)off;
      row.recs.swap(rows[r].recs);
      p.n_records += (long)row.recs.size();
      live.push_back(std::move(row));
    }
    p.n_rows += (long)live.size();
    {
      long recs = 0;
      for (const Row &r : live) recs += (long)r.recs.size();
      p.blk_cost.push_back(3.0 * (double)live.size() + 5.0 * (double)recs);
    }
    // this wave's pieces of the block staged while this unit runs: block blk + ahead of this tile
    // or, past its last block, of the workgroup's next tile
    pieces.clear();
    if (opt.dma.on) {
      const DmaPlan &d = opt.dma;
      const int wave = ocg % d.waves;
      const int nb = (blk + d.ahead) % t.n_icb;
      const int nch = std::min(t.icb, g.Cg - nb * t.icb);
      const long total = (long)nch * d.qpc;
      const int n_instr = (int)((total + 63) / 64);
      const int ch_per_period = d.period / d.qpc;
      for (int i = wave; i < n_instr; i += d.waves) {
        const long e0 = (long)i * 64;
        Piece pc;
        pc.tab_off = (uint32_t)((e0 % d.period) * 4);
        pc.lds_off = (uint32_t)i * 1024u;
        pc.soff = (uint32_t)(((long)nb * t.icb + (e0 / d.period) * ch_per_period) * d.chan_bytes);
        pc.lanes = (int)std::min<long>(64, total - e0);
        pieces.push_back(pc);
      }
      p.n_dma += (long)pieces.size();
    }
    while ((p.code.size() * 4) % kUnitAlign) enc_nop(p.code);
    p.off[blk] = (uint32_t)(p.code.size() * 4);
    const size_t at = p.code.size();
    patches.push_back(emit_unit(p.code, live, pieces, opt, n_pref, t.pix_waves == 1 ? ocg % t.oc_waves : -1, blk, t.n_icb, t.G * g.KW));
    p.max_unit = std::max(p.max_unit, (p.code.size() - at) * 4);
  }
  // the distances: unit blk touches the code of unit (blk + 1) % n_icb
  if (n_pref > 0)
    for (int blk = 0; blk < t.n_icb; ++blk) {
      const int nxt = (blk + 1) % t.n_icb;
      const long long from = (long long)p.off[blk] + 4;        // what s_getpc_b64 returned
      const long long d = (long long)p.off[nxt] - from;
      p.code[patches[blk]] = (uint32_t)d;
      if (d < 0) p.code[patches[blk] + 1] = 0x82000000u | ((uint32_t)(kSPref + 1) << 16) | (0xC1u << 8) | (uint32_t)(kSPref + 1);
    }
}
}  // namespace

static Program build_pass(const ConvGeom &g, const Tiling &t, const std::vector<std::vector<int>> &rowptr,
                          const std::vector<std::vector<int>> &colidx,
                          const std::vector<std::vector<float>> &values, const Options &opt, int n_pref,
                          size_t *max_unit_bytes, const std::vector<uint32_t> &chan) {
  Program p;
  const size_t n_chains = (size_t)g.group * t.n_ocg, n_units = n_chains * t.n_icb;
  p.unit_off.assign(n_units, 0u);
  p.chan = chan;
  // the chains are independent (a unit's touches stay inside its chain): large layers generate them on up to eight
  // threads, and the blobs are put together in order -- the same bytes as one thread would write
  std::vector<ChainOut> outs(n_chains);
  auto run = [&](size_t ci) {
    const int cg = (int)(ci / t.n_ocg), ocg = (int)(ci % t.n_ocg);
    emit_chain(g, t, rowptr[cg], colidx[cg], values[cg], opt, n_pref, &p.chan[ci * t.G], ocg, &outs[ci]);
  };
  size_t nnz = 0;
  for (const auto &c : colidx) nnz += c.size();
  const size_t hw = (size_t)allowed_cores();     // (the cores this thread may use, not the machine's: thread_place.h)
  const size_t n_thr = nnz >= 20000 && n_chains > 1 ? std::min<size_t>(std::min<size_t>(8, hw), n_chains) : 1;
  // (parallel_for.h: the calling thread works too, a thread the system refuses is one worker fewer, an exception on
  //  any thread -- a code vector that cannot grow -- is rethrown here after the helpers were joined)
  parallel_for(n_chains, n_thr, run);
  size_t total = 0;
  for (const ChainOut &c : outs) total += c.code.size() + kUnitAlign / 4;
  p.code.reserve(total + 64 + (size_t)n_pref * 1024);
  for (size_t ci = 0; ci < n_chains; ++ci) {
    const ChainOut &c = outs[ci];
    while ((p.code.size() * 4) % kUnitAlign) enc_nop(p.code);
    const uint32_t base = (uint32_t)(p.code.size() * 4);
    p.code.insert(p.code.end(), c.code.begin(), c.code.end());
    for (int blk = 0; blk < t.n_icb; ++blk) p.unit_off[ci * t.n_icb + blk] = base + c.off[blk];
    p.n_rows += c.n_rows; p.n_records += c.n_records; p.n_dma += c.n_dma;
    p.overflow = p.overflow || c.overflow;
    *max_unit_bytes = std::max(*max_unit_bytes, c.max_unit);
  }
  // How well the channel deal balanced the waves: the waves of a workgroup column meet at a barrier after every
  // block, so a block costs its SLOWEST wave.  sum over (conv group, column, block) of the slowest wave's
  // instructions / sum of the mean wave's = what the barriers make a launch wait for (1.0 = perfectly even);
  // the worst single block is reported too (blocks under 64 instructions per wave are noise and left out of it).
  {
    double sum_max = 0, sum_mean = 0, worst = 1.0;
    const int ow = std::max(1, t.oc_waves);
    for (int cg = 0; cg < g.group; ++cg)
      for (int col = 0; col * ow < t.n_ocg; ++col)
        for (int blk = 0; blk < t.n_icb; ++blk) {
          double mx = 0, tot = 0;
          int n = 0;
          for (int w = 0; w < ow && col * ow + w < t.n_ocg; ++w, ++n) {
            const double c = outs[(size_t)cg * t.n_ocg + col * ow + w].blk_cost[blk];
            mx = std::max(mx, c);
            tot += c;
          }
          if (n == 0 || tot <= 0) continue;
          sum_max += mx;
          sum_mean += tot / n;
          if (tot / n >= 64.0) worst = std::max(worst, mx / (tot / n));
        }
    p.deal_slowest_over_mean = sum_mean > 0 ? sum_max / sum_mean : 1.0;
    p.deal_worst_block = worst;
  }
  // instruction prefetch and the code touches run past the last unit: keep them inside the blob
  for (int i = 0; i < 64 + n_pref * 1024; ++i) enc_nop(p.code);
  p.n_pref = n_pref;
  p.chained = opt.chain.on;
  return p;
}

Program build_program(const ConvGeom &g, const Tiling &t, const std::vector<std::vector<int>> &rowptr,
                      const std::vector<std::vector<int>> &colidx,
                      const std::vector<std::vector<float>> &values, const Options &opt_in) {
  Options opt = opt_in;
  if (!opt.dma.on || t.n_ocg % std::max(1, t.oc_waves) != 0 || t.pix_waves != 1) opt.chain.on = false;
  // Tile B (the lane's second quad: flattened rows rows_per_slab .. 2 rows_per_slab - 1) lies past the
  // workgroup's last row when all of them fit tile A -- every small pointwise image walked one (or a
  // few) to a workgroup: no reads, no FMAs for it (the kernel's epilogue stores none of its lanes)
  // ... and a tiling with one quad per lane (stream_builder.h, Tiling::tpl) has no tile B at all: its
  // accumulators hold channels 24 .. 47 of the wave
  if (t.tpl == 1 || (t.pix_waves == 1 && t.tr * t.nseg <= t.rows_per_slab && opt.one_tile >= 0)) opt.one_tile = 1;
  else opt.one_tile = 0;
  // tile B's accumulators as input registers: only where they hold nothing (two quads per lane in the tiling,
  // none of tile B's rows in the image) -- with one quad per lane they carry channels 24 .. 47
  if (!(opt.one_tile && t.tpl == 2)) opt.hi_sets = 0;
  // the channel deal (once: both passes below generate code for the same deal)
  std::vector<uint32_t> chan;
  chan.reserve((size_t)g.group * t.n_ocg * t.G);
  for (int cg = 0; cg < g.group; ++cg) {
    // instructions of generated code per nonempty row (two reads, a wait) and per nonzero
    const std::vector<uint32_t> sl = balance_channels(g, t, rowptr[cg], colidx[cg], 3.0, 5.0);
    chan.insert(chan.end(), sl.begin(), sl.end());
  }
  size_t max_unit = 0;
  if (!opt.prefetch) return build_pass(g, t, rowptr, colidx, values, opt, 0, &max_unit, chan);
  // Every unit carries the same number of code touches: enough for the longest unit, its own touches included.
  // The longest unit's size comes from an upper bound on its instruction bytes (a pass of its own over the
  // nonzeros, no code emitted: round 3 generated every layer twice to learn it) -- at worst one touch too many.
  size_t bound = 0;
  {
    const int rows_per_blk = t.icb * g.KH;
    std::vector<uint32_t> row_mark(rows_per_blk, 0u);
    uint32_t stamp = 0;
    const size_t pieces_per_unit = opt.dma.on ? (size_t)(((long)t.icb * opt.dma.qpc + 63) / 64 + opt.dma.waves - 1) / opt.dma.waves : 0;
    for (int cg = 0; cg < g.group; ++cg)
      for (int ocg = 0; ocg < t.n_ocg; ++ocg) {
        std::vector<size_t> recs(t.n_icb, 0), nrows(t.n_icb, 0);
        for (int blk = 0; blk < t.n_icb; ++blk) {
          ++stamp;
          const int ic_lo = blk * t.icb, ic_hi = std::min(g.Cg, ic_lo + t.icb);
          for (int gl = 0; gl < t.G && ocg * t.G + gl < g.Mg; ++gl) {
            const int m = (int)chan[((size_t)cg * t.n_ocg + ocg) * t.G + gl];
            // (columns ascend within a CSR row: the block's nonzeros are one contiguous run)
            const int *b = colidx[cg].data() + rowptr[cg][m], *e = colidx[cg].data() + rowptr[cg][m + 1];
            const int *lo = std::lower_bound(b, e, ic_lo * g.KH * g.KW), *hi = std::lower_bound(lo, e, ic_hi * g.KH * g.KW);
            recs[blk] += (size_t)(hi - lo);
            for (const int *c = lo; c < hi; ++c) {
              const int r = (*c / g.KW) - ic_lo * g.KH;      // (ic - ic_lo) * KH + kr
              if (row_mark[r] != stamp) { row_mark[r] = stamp; ++nrows[blk]; }
            }
          }
          // bytes: per row two reads and a wait (+ a priority switch), per nonzero a move and four FMAs, per piece
          // table read, wait, M0, offset, EXEC dance, nop, load; prologue / chain transition / alignment
          const size_t b_unit = 256 + nrows[blk] * 24 + recs[blk] * 40 + pieces_per_unit * 56;
          bound = std::max(bound, b_unit);
        }
      }
  }
  int n_pref = 1;
  while ((size_t)n_pref * 4096 < bound + (size_t)n_pref * 20) ++n_pref;
  Program p = build_pass(g, t, rowptr, colidx, values, opt, n_pref, &max_unit, chan);
  if (!p.overflow && max_unit > (size_t)n_pref * 4096) {
    // (the bound was not one: generate again with what the longest unit really needs)
    while ((size_t)n_pref * 4096 < max_unit + 16 + (size_t)n_pref * 20) ++n_pref;
    max_unit = 0;
    p = build_pass(g, t, rowptr, colidx, values, opt, n_pref, &max_unit, chan);
  }
  return p;
}

}  // namespace jit
}  // namespace escoin
