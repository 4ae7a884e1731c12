// jit_codegen.cpp -- see jit_codegen.h
#include "jit_codegen.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace escoin {
namespace jit {

Options options_from_env() {
  Options o;
  if (const char *e = getenv("ESCOIN_JIT_DEPTH")) o.depth = std::max(1, std::min(2, atoi(e)));
  if (const char *e = getenv("ESCOIN_JIT_HOIST")) o.hoist_weight = atoi(e) != 0;
  if (const char *e = getenv("ESCOIN_JIT_PRIO_ROWS")) o.prio_rows = std::max(0, atoi(e));
  if (const char *e = getenv("ESCOIN_JIT_ABL")) o.ablate = atoi(e);
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

// The walk of one unit.  Rows are read `depth` ahead into the three input sets in rotation; LDS
// returns data in order, so "all but the reads of the rows still ahead" is a counted wait.  A
// set is overwritten by the read issued two rows after the one that used it: its FMAs were
// issued (in order) before that read was.
void emit_unit(std::vector<uint32_t> &c, const std::vector<Row> &rows, const Options &opt) {
  const int n = (opt.ablate & 8) ? 0 : (int)rows.size();
  const int depth = opt.depth;
  auto issue = [&](int k) {
    if (opt.ablate & 2) return;
    const int base = kVIn0 + 8 * (k % kInSets);
    enc_ds_read_b128(c, base, kVAddrA, rows[k].lds_off);
    enc_ds_read_b128(c, base + 4, kVAddrB, rows[k].lds_off);
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
  if (opt.hoist_weight && !flat.empty() && !(opt.ablate & 4)) enc_s_mov_lit(c, sreg(0), flat[0]->bits);
  int prio = 0;
  for (int k = 0; k < n; ++k) {
    if (k + depth < n) issue(k + depth);
    if (opt.prio_rows > 0 && k % opt.prio_rows == 0) {
      prio ^= 1;
      enc_setprio(c, prio);
    }
    const int ahead = std::min(n - 1, k + depth) - k;
    enc_waitcnt_lgkm(c, (opt.ablate & 2) ? 0 : 2 * ahead);
    const int xa = kVIn0 + 8 * (k % kInSets), xb = xa + 4;
    for (int j = first_of_row[k]; j < first_of_row[k + 1]; ++j) {
      if (opt.ablate & 4) {
      } else if (opt.hoist_weight) {
        if (j + 1 < (int)flat.size()) enc_s_mov_lit(c, sreg(j + 1), flat[j + 1]->bits);
      } else {
        enc_s_mov_lit(c, sreg(j), flat[j]->bits);
      }
      if (opt.ablate & 1) continue;
      const int a = 4 * flat[j]->idx;
      enc_pk_fma(c, kAccA + a, sreg(j), xa);
      enc_pk_fma(c, kAccA + a + 2, sreg(j), xa + 2);
      enc_pk_fma(c, kAccB + a, sreg(j), xb);
      enc_pk_fma(c, kAccB + a + 2, sreg(j), xb + 2);
    }
  }
  if (opt.prio_rows > 0 && prio) enc_setprio(c, 0);
  enc_setpc_return(c);
}

}  // namespace

Program build_program(const ConvGeom &g, const Tiling &t, const std::vector<std::vector<int>> &rowptr,
                      const std::vector<std::vector<int>> &colidx,
                      const std::vector<std::vector<float>> &values, const Options &opt) {
  Program p;
  const size_t n_units = (size_t)g.group * t.n_ocg * t.n_icb;
  p.unit_off.assign(n_units, 0u);
  p.chan.reserve((size_t)g.group * t.n_ocg * t.G);
  for (int cg = 0; cg < g.group; ++cg) {
    // instructions of generated code per nonempty row (two reads, a wait) and per nonzero
    const std::vector<uint32_t> sl = balance_channels(g, t, rowptr[cg], colidx[cg], 3.0, 5.0);
    p.chan.insert(p.chan.end(), sl.begin(), sl.end());
  }
  const int rows_per_blk = t.icb * g.KH;
  std::vector<Row> rows(rows_per_blk), live;
  for (int cg = 0; cg < g.group; ++cg)
    for (int ocg = 0; ocg < t.n_ocg; ++ocg)
      for (int blk = 0; blk < t.n_icb; ++blk) {
        for (auto &r : rows) r.recs.clear();
        const int ic_lo = blk * t.icb, ic_hi = std::min(g.Cg, ic_lo + t.icb);
        for (int gl = 0; gl < t.G; ++gl) {
          if (ocg * t.G + gl >= g.Mg) break;
          const int m = (int)p.chan[((size_t)cg * t.n_ocg + ocg) * t.G + gl];
          for (int j = rowptr[cg][m]; j < rowptr[cg][m + 1]; ++j) {
            const int col = colidx[cg][j];
            const int kc = col % g.KW, kr = (col / g.KW) % g.KH, ic = col / (g.KW * g.KH);
            if (ic < ic_lo || ic >= ic_hi) continue;
            Rec rec;
            std::memcpy(&rec.bits, &values[cg][j], 4);
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
          row.recs = rows[r].recs;
          p.n_records += (long)row.recs.size();
          live.push_back(std::move(row));
        }
        p.n_rows += (long)live.size();
        while ((p.code.size() * 4) % kUnitAlign) enc_nop(p.code);
        p.unit_off[((size_t)cg * t.n_ocg + ocg) * t.n_icb + blk] = (uint32_t)(p.code.size() * 4);
        emit_unit(p.code, live, opt);
      }
  // instruction prefetch runs past the last unit's return: keep it inside the blob
  for (int i = 0; i < 64; ++i) enc_nop(p.code);
  return p;
}

}  // namespace jit
}  // namespace escoin
