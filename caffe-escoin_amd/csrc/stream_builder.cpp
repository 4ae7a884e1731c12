// stream_builder.cpp -- see stream_builder.h
#include "stream_builder.h"

#include "knobs.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include "parallel_for.h"
#include <thread>
#include <cstdlib>
#include <cstring>

namespace escoin {

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

namespace {

// One candidate tiling of a concrete (H, W) cut of the image: `passes` workgroup columns per conv
// group (0: as few as the accumulator file allows) and, for layers whose images fit a workgroup
// tile, `nseg` whole images per workgroup (0: as many as fit).
Tiling tile_with(const ConvGeom &g, int waves_per_wg, int lds_budget_bytes, int passes, int nseg, int tpl = kTilesPerLane) {
  Tiling t;
  t.tpl = tpl;
  t.H = g.H; t.W = g.W; t.OH = g.OH; t.OW = g.OW;
  // epilogue shifts: s = kc - pad_w must satisfy |s| <= 4 (one neighbouring quad)
  if (g.KW < 1 || g.KW > 5 || g.pad_w > 4 || g.KW - 1 - g.pad_w > 4) return t;
  if (g.W > 256 || g.OW > 256 || g.OH < 1 || g.OW < 1) return t;
  if (waves_per_wg != 1 && waves_per_wg != 2 && waves_per_wg != 4 && waves_per_wg != 8) return t;
  t.KW = g.KW;
  t.KH = g.KH;
  // a lane quad is an input quad AND an output quad: the row must hold the wider of the two
  // (OW > W when pad > (KW - 1) / 2); rows are >= 32 bytes (the stream stores offsets / 32)
  t.S4 = std::max(2, next_pow2((std::max(g.W, g.OW) + 3) / 4));
  t.RS = 4 * t.S4;
  t.rows_per_slab = 64 / t.S4;
  // Output channels per wave.  The accumulator file bounds it (KW=1:24 2:12 3:8 4:6 5:4); within
  // that bound the channels are spread over as many waves of the workgroup as there are channels
  // and balanced over the passes: a layer with few output channels then gets a small pixel tile
  // per workgroup (more workgroups for the same batch) and every wave of it shares one staged
  // input tile, instead of a few workgroups whose waves each own 24 channels' worth of nothing.
  const int gmax = kAccRegsPerTile * (kTilesPerLane / tpl) / (4 * g.KW);
  const int mg = std::max(1, g.Mg);
  t.waves = waves_per_wg;
  t.oc_waves = 1;
  while (t.oc_waves * 2 <= waves_per_wg && t.oc_waves * 2 <= mg) t.oc_waves *= 2;
  const int min_passes = (mg + gmax * t.oc_waves - 1) / (gmax * t.oc_waves);
  passes = std::max(passes, min_passes);
  t.G = (mg + t.oc_waves * passes - 1) / (t.oc_waves * passes);
  t.n_ocg = (g.Mg + t.G - 1) / t.G;
  t.pix_waves = waves_per_wg / t.oc_waves;
  t.n_ocblk = (t.n_ocg + t.oc_waves - 1) / t.oc_waves;
  t.rows_per_wg = t.pix_waves * tpl * t.rows_per_slab;
  if (g.OH <= t.rows_per_wg) {
    t.band_mode = false;
    t.tr = g.OH;
    const int fit = t.rows_per_wg / g.OH;
    t.nseg = nseg > 0 ? std::min(nseg, fit) : fit;
    t.bands = 1;
  } else {
    // bands of equal height: 28 rows on 16 row slots are two bands of 14, not 16 + 12 -- the same
    // lanes do the same work, but no band stages (or zero-fills) rows past the image, the planes are
    // smaller (more channels per block) and every tile costs the same
    t.band_mode = true;
    t.nseg = 1;
    t.bands = (g.OH + t.rows_per_wg - 1) / t.rows_per_wg;
    t.tr = (g.OH + t.bands - 1) / t.bands;
  }
  t.plane_rows = t.tr + g.KH - 1;
  t.plane_seg_floats = t.plane_rows * t.RS;
  t.plane_ch_floats = t.nseg * t.plane_seg_floats;
  // A channel plane just short of 64 / 128 / 256 quads is padded up to it (13 x 13 with its halo:
  // 120 -> 128 quads): whole 1 KiB DMA instructions per channel, a wave's instructions an
  // arithmetic progression with one table entry -- the fast issue path and, for 3x3 layers, the
  // kernel that issues them from inside the stream walk.  (The padding quads are filled like any
  // row past the image: zeros, or rows of the image nobody reads.)
  for (int p2 = 64; p2 <= 256; p2 *= 2) {
    const int q = t.plane_ch_floats / 4;
    if (q < p2 && q * 16 >= p2 * 15) t.plane_ch_floats = p2 * 4;
  }
  // A pointwise image walked as ONE row by a workgroup of its own (14 x 14 as 1 x 196: 49 quads on a
  // 64-quad row) needs no row pitch: the planes are packed to the quads that exist.  Lanes past the row
  // read the next channel's quads (theirs are results nobody stores); the fill moves 23 % fewer bytes.
  if (g.sub == 1 && g.KH == 1 && g.KW == 1 && g.pad_h == 0 && g.pad_w == 0 && !t.band_mode && t.nseg == 1 && t.tr == 1 &&
      (ESC_KNOB("PACK_ROW", 1) != 0))
    t.plane_ch_floats = std::min(t.plane_ch_floats, (g.W + 7) / 8 * 8);   // (whole 32 bytes: the stream's row offsets)
  const int per_ch = t.plane_ch_floats * 4;
  // (row offsets travel as offset / 32 in 11-bit fields of the stream: 64 KiB per plane buffer)
  lds_budget_bytes = std::min(lds_budget_bytes, 64 * 1024);
  int icb = lds_budget_bytes / per_ch;
  if (icb < 1) return t;                       // one channel does not fit
  icb = std::min(icb, g.Cg);
  // balance the blocks: same number of blocks, evenly sized
  t.n_icb = (g.Cg + icb - 1) / icb;
  t.icb = (g.Cg + t.n_icb - 1) / t.n_icb;
  t.planes_bytes = t.icb * per_ch;
  t.ok = true;
  return t;
}

// Estimated microseconds per launch (constants measured on MI355X): a workgroup
// pays ~1.2 us per block (barrier, DMA issue, loop prologue), ~60 ns per nonempty input row and
// ~18 ns per nonzero in its waves' stream walk -- the same whether its lanes are full or not --
// stages its planes at ~50 GB/s under that walk, and the launch takes as many rounds as there are
// workgroups per CU.
double launch_cost_us(const ConvGeom &g, const Tiling &t, int n_cu) {
  const double d = g.density > 0.f ? g.density : 0.1;
  const double rows = (double)g.Cg * g.KH;
  const double nonempty = 1.0 - std::pow(1.0 - d, (double)t.G * g.KW);
  const double walk = rows * nonempty * 0.060 + rows * g.KW * t.G * d * 0.018;
  // (planes land at 18-20 GB/s per CU with every CU staging: GoogLeNet's 28 x 28 layers read 154 MB in
  // 34.6 us with the walk and the stores switched off; rounds 1-2 assumed 50)
  static const double dma_bytes_per_us = (ESC_KNOB_F("DMA_GBPS", 19.0)) * 1e3;
  const double dma = (double)g.Cg * t.plane_ch_floats * 4.0 / dma_bytes_per_us;
  const double epilogue = 0.15 * t.G * t.tpl;
  const double per_tile = t.n_icb * 1.2 + std::max(walk, dma) + epilogue;
  const long tiles = t.band_mode ? (long)g.N * t.bands : ((long)g.N + t.nseg - 1) / t.nseg;
  const long cus = std::max(1, n_cu);
  // persistent workgroups: grid.x = CUs / grid.y of them walk the tiles; when grid.y alone exceeds
  // the CUs the workgroups run in several waves
  const long gy = (long)g.group * t.n_ocblk;
  const long gx = std::min(tiles, std::max<long>(1, cus / gy));
  const long tiles_per_wg = (tiles + gx - 1) / gx;
  const long waves_of_wgs = (gx * gy + cus - 1) / cus;
  // ... and no faster than its HBM traffic: every workgroup column stages the whole input once
  // more.  The first read comes from HBM (~4.5 TB/s through the LDS-DMA path, measured on GoogLeNet's
  // 28 x 28 layers), the columns' re-reads mostly from the Infinity Cache (counted at half price).
  // Without this term 160 output channels at 95 % were cut into two passes of 10 channels per wave
  // (one pass of 20 runs in 2/3 of the time).
  const double in_bytes = 4.0 * g.N * g.C * g.H * g.W, out_bytes = 4.0 * g.N * g.M * g.OH * g.OW;
  const double hbm_us = (in_bytes * (1.0 + 0.5 * (t.n_ocblk - 1)) + out_bytes) / 4.5e6;
  return std::max((double)tiles_per_wg * per_tile * (double)waves_of_wgs, hbm_us);
}

// The tiling of one concrete (H, W) cut: the cheapest of the candidates (more passes = fewer
// output channels and a shorter stream per wave, but the input staged once more; fewer images per
// workgroup = more workgroups for small batches, but emptier lanes).
Tiling tile_for(const ConvGeom &g, int waves_per_wg, int lds_budget_bytes, int n_cu, bool one_tile_ok) {
  Tiling best = tile_with(g, waves_per_wg, lds_budget_bytes, 0, 0);
  if (!best.ok) return best;
  // Generated code chains a tile's blocks only where every wave of every workgroup column has an oc-group
  // (jit_codegen.h ChainPlan: n_ocg a multiple of oc_waves); a tiling that leaves a column's last waves without one
  // falls back to a call per block and keeps those waves idle.  Measured on every candidate of the batch sweep's
  // cliffs (tools/tiling_oracle.py, profiles/r05_batch_sweep.md): res4 with 5 columns of G = 7 (37 oc-groups) 160-170 us
  // where 4 columns of G = 8 take 123-154; 6 and 7 columns 180-215.  The model prices such tilings a quarter up
  // (one_tile_ok is what the caller passes for generated code).
  auto cost_of = [&](const Tiling &t) {
    return launch_cost_us(g, t, n_cu) * ((one_tile_ok && t.oc_waves > 1 && t.n_ocg % t.oc_waves != 0) ? 1.25 : 1.0);
  };
  // (experiments flavour: ESCOIN_FORCE_PASSES / ESCOIN_FORCE_NSEG pin the workgroup columns per conv group / the images per
  //  tile, for sweeps of what the cost model could have chosen: tools/tiling_oracle.py)
  if (ESC_KNOB_SET("FORCE_PASSES") || ESC_KNOB_SET("FORCE_NSEG") || ESC_KNOB_SET("FORCE_TPL")) {
    // (ESCOIN_FORCE_TPL=1: one quad per lane, up to 48 channels per wave -- generated code on pointwise layers whose tile
    //  rows fit tile A, the same conditions as the search below)
    const int tpl = (int)ESC_KNOB("FORCE_TPL", kTilesPerLane);
    const Tiling t = tile_with(g, waves_per_wg, lds_budget_bytes, (int)ESC_KNOB("FORCE_PASSES", 0), (int)ESC_KNOB("FORCE_NSEG", 0), tpl);
    const bool tpl_ok = tpl == kTilesPerLane || (tpl == 1 && one_tile_ok && g.KH == 1 && g.KW == 1 && !t.band_mode && t.pix_waves == 1 &&
                                                 t.tr * t.nseg <= t.rows_per_slab);
    if (t.ok && t.G >= 1 && tpl_ok) return t;
  }
  double best_cost = cost_of(best);
  const int base_passes = best.n_ocblk;
  const int fit = best.band_mode ? 1 : best.nseg;
  for (int passes = base_passes; passes <= 8 * base_passes; ++passes) {
    for (int nseg = fit; nseg >= 1; --nseg) {
      const Tiling t = tile_with(g, waves_per_wg, lds_budget_bytes, passes, nseg);
      if (!t.ok || t.G < 1) continue;
      const double c = cost_of(t);
      // more passes stage the input more often: only for a clear win; fewer images per tile at
      // the same number of passes cost nothing (fewer, larger blocks): any win counts
      if (c < best_cost * (passes > best.n_ocblk ? 0.97 : 0.999)) {
        best = t;
        best_cost = c;
      }
    }
    if (best.G == 1) break;
  }
  // One quad per lane (generated code, pointwise layers): where every row of the tile fits tile A
  // anyway, the accumulators of tile B can hold more channels instead -- up to 48 per wave, 384 per
  // workgroup column -- and a layer with 193 .. 384 output channels stages its input once, not twice.
  // Taken only when it saves workgroup columns.
  if (one_tile_ok && g.KH == 1 && g.KW == 1 && !best.band_mode && best.pix_waves == 1 &&
      best.tr * best.nseg <= best.rows_per_slab && best.n_ocblk > 1) {
    for (int passes = 1; passes < best.n_ocblk; ++passes) {
      const Tiling t = tile_with(g, waves_per_wg, lds_budget_bytes, passes, best.nseg, 1);
      if (!t.ok || t.G < 1 || t.n_ocblk >= best.n_ocblk || t.band_mode || t.pix_waves != 1) continue;
      const double c = cost_of(t);
      if (c < best_cost * 0.97) {
        best = t;
        best_cost = c;
        break;
      }
    }
  }
  return best;
}

}  // namespace

Tiling choose_tiling(const ConvGeom &g, int waves_per_wg, int lds_budget_bytes, int n_cu, bool one_tile_ok) {
  if (n_cu < 1) n_cu = 256;
  Tiling best = tile_for(g, waves_per_wg, lds_budget_bytes, n_cu, one_tile_ok);
  // A pointwise layer (1x1, no padding) does not care where the rows of an image break: its
  // H*W pixels are one contiguous run per channel.  These layers are bound by the LDS-DMA fill rate
  // (DESIGN.md 4.1), and a fill instruction costs the same whether its 16-byte slots carry pixels,
  // straddle a row end (W not a multiple of 4) or are zero-filled padding (RS > W, band rows past the
  // image): re-cut the image into rows of W' | H*W whose tiles stage the fewest slots per image
  // (then: fewest copies that straddle, best lane use, longest rows).  56 x 56 is walked as 49 x 64,
  // 28 x 28 as 98 x 8, 14 x 14 as 7 x 28: no or 1/8 padding instead of 1/8 .. 1/4.  Input and output
  // blobs are the same memory either way.
  if (g.sub == 1 && g.KH == 1 && g.KW == 1 && g.pad_h == 0 && g.pad_w == 0 && g.OH == g.H && g.OW == g.W) {
    const int hw = g.H * g.W;
    auto slots = [](const Tiling &t) {   // 16-byte LDS slots staged per image and channel
      return t.band_mode ? (long)t.bands * t.plane_rows * t.S4 : (long)t.plane_ch_floats / 4 / t.nseg;
    };
    auto copies = [](const Tiling &t) { return (long)t.H * ((t.W + 3) / 4); };
    auto lane_use = [](const Tiling &t) {
      const double rows = t.band_mode ? (double)t.bands * t.rows_per_wg : (double)t.rows_per_wg;
      const double used = t.band_mode ? (double)t.H : (double)t.H * (t.rows_per_wg / t.H);
      return used * t.W / (rows * t.RS);
    };
    static const bool no_recut = ESC_KNOB_SET("NORECUT");
    for (int w = 1; w <= 256 && w <= hw; ++w) {
      if (hw % w != 0 || w == g.W) continue;
      if (g.W % 4 == 0 && (w % 4 != 0 || no_recut)) continue;   // never trade aligned rows for straddling ones
      ConvGeom c = g;
      c.W = c.OW = w;
      c.H = c.OH = hw / w;
      const Tiling t = tile_for(c, waves_per_wg, lds_budget_bytes, n_cu, one_tile_ok);
      if (!t.ok) continue;
      bool better = !best.ok;
      if (!better) {
        const long sa = slots(t), sb = slots(best);
        const long ca = copies(t), cb = copies(best);
        const double ua = lane_use(t), ub = lane_use(best);
        const bool tie = ua > ub + 1e-9 || (ua > ub - 1e-9 && t.W > best.W);
        if (g.W % 4 != 0) better = ca < cb || (ca == cb && tie);   // fewest straddling copies first
        else better = sa < sb || (sa == sb && tie);                  // aligned rows: fewest staged slots
      }
      if (better) best = t;
    }
  }
  return best;
}

Tiling repad_planes(const ConvGeom &g, Tiling t, int qpc, int lds_budget_bytes) {
  if (!t.ok || qpc * 4 < t.plane_ch_floats) return t;
  t.plane_ch_floats = qpc * 4;
  const int per_ch = t.plane_ch_floats * 4;
  lds_budget_bytes = std::min(lds_budget_bytes, 64 * 1024);
  int icb = lds_budget_bytes / per_ch;
  if (icb < 1) {
    t.ok = false;
    return t;
  }
  icb = std::min(icb, g.Cg);
  t.n_icb = (g.Cg + icb - 1) / icb;
  t.icb = (g.Cg + t.n_icb - 1) / t.n_icb;
  t.planes_bytes = t.icb * per_ch;
  return t;
}

namespace {
struct Rec {
  float val;
  uint8_t idx;
};
struct Group {
  uint32_t lds_off;
  std::vector<Rec> recs;
};
}  // namespace

std::vector<uint32_t> balance_channels(const ConvGeom &g, const Tiling &t, const std::vector<int> &rowptr,
                                       const std::vector<int> &colidx, double group_cost, double record_cost) {
  const int G = t.G, n_ocg = t.n_ocg, Mg = g.Mg;
  std::vector<uint32_t> slot(static_cast<size_t>(n_ocg) * G);
  for (int o = 0; o < n_ocg; ++o)
    for (int gl = 0; gl < G; ++gl) slot[(size_t)o * G + gl] = (uint32_t)std::min(o * G + gl, Mg - 1);
  static const bool enabled = (ESC_KNOB("BALANCE", 1) != 0);
  if (!enabled || g.KW == 1 || t.oc_waves < 2 || Mg < 2 * G) return slot;
  const double kGroupCost = group_cost, kRecordCost = record_cost;
  const int rows_per_blk = t.icb * g.KH;
  const int words = (rows_per_blk + 63) / 64;
  const int nb = t.n_icb;
  // per channel and block: which input rows it touches, how many nonzeros
  std::vector<uint64_t> mask((size_t)Mg * nb * words, 0ull);
  std::vector<int> recs((size_t)Mg * nb, 0);
  for (int m = 0; m < Mg; ++m)
    for (int j = rowptr[m]; j < rowptr[m + 1]; ++j) {
      const int col = colidx[j];
      const int kr = (col / g.KW) % g.KH, ic = col / (g.KW * g.KH);
      const int b = ic / t.icb, r = (ic - b * t.icb) * g.KH + kr;
      mask[((size_t)m * nb + b) * words + (r >> 6)] |= 1ull << (r & 63);
      ++recs[(size_t)m * nb + b];
    }
  auto count_of = [&](int o) { return std::max(0, std::min(G, Mg - o * G)); };
  // Where the refinement below starts from.  Rounds 2-4 started from the channels' natural order and only ever
  // swapped channels INSIDE a workgroup column: fine for uniformly pruned weights, not for a pruned model's -- a filter
  // at four times the mean density next to all-zero ones (synth "filters_tail") left the slowest wave of a block 24 %
  // over the mean on AlexNet's conv2 and the step 15 % slower than with uniform weights (profiles/r05_skew.md).
  // So first a global deal: channels by descending total cost (rows touched and nonzeros over all blocks), each to
  // the oc-group -- of ANY column -- with the least cost so far that still has a free slot (longest processing time
  // first).  Heavy filters end up on different waves and different workgroups, empty ones fill the gaps.
  // Deterministic (stable sort, first minimum).  ESCOIN_DEAL_LPT=0 (experiments flavour): the natural order.
  static const bool lpt = (ESC_KNOB("DEAL_LPT", 1) != 0);
  if (lpt) {
    std::vector<double> cm(Mg, 0.0);
    for (int m = 0; m < Mg; ++m) {
      int rows = 0, rc = 0;
      for (int b = 0; b < nb; ++b) {
        const uint64_t *mk = &mask[((size_t)m * nb + b) * words];
        for (int w = 0; w < words; ++w) rows += __builtin_popcountll(mk[w]);
        rc += recs[(size_t)m * nb + b];
      }
      cm[m] = kGroupCost * rows + kRecordCost * rc;
    }
    std::vector<int> order(Mg);
    for (int m = 0; m < Mg; ++m) order[m] = m;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cm[a] > cm[b]; });
    std::vector<double> load(n_ocg, 0.0);
    std::vector<int> fill(n_ocg, 0);
    for (int m : order) {
      int best = -1;
      for (int o = 0; o < n_ocg; ++o)
        if (fill[o] < count_of(o) && (best < 0 || load[o] < load[best])) best = o;
      slot[(size_t)best * G + fill[best]] = (uint32_t)m;
      ++fill[best];
      load[best] += cm[m];
    }
    // (slots past a partly filled oc-group's last channel repeat a valid channel, as before)
    for (int o = 0; o < n_ocg; ++o)
      for (int gl = count_of(o); gl < G; ++gl) slot[(size_t)o * G + gl] = slot[(size_t)o * G + std::max(0, count_of(o) - 1)];
  }
  // (one workgroup column at a time; the columns touch disjoint slots and run on threads of their own below)
  auto deal_column = [&](int blk0) {
  std::vector<uint64_t> acc(words);
  auto wave_cost = [&](int o, int b) {      // cost of oc-group o in block b under `slot`
    std::fill(acc.begin(), acc.end(), 0ull);
    int rc = 0;
    const int n = count_of(o);
    for (int gl = 0; gl < n; ++gl) {
      const int m = (int)slot[(size_t)o * G + gl];
      const uint64_t *mk = &mask[((size_t)m * nb + b) * words];
      for (int w = 0; w < words; ++w) acc[w] |= mk[w];
      rc += recs[(size_t)m * nb + b];
    }
    int rows = 0;
    for (int w = 0; w < words; ++w) rows += __builtin_popcountll(acc[w]);
    return kGroupCost * rows + kRecordCost * rc;
  };
  {
    const int nw = std::min(t.oc_waves, n_ocg - blk0);
    if (nw < 2) return;
    std::vector<double> cost((size_t)nw * nb);
    for (int w = 0; w < nw; ++w)
      for (int b = 0; b < nb; ++b) cost[(size_t)w * nb + b] = wave_cost(blk0 + w, b);
    // A trial swaps ONE channel of wave wa with one of wave wb: per block the rows of wa become (rows of its
    // other channels) | (rows of the incoming channel), its nonzeros change by the difference of the two -- so per
    // wave, slot and block the OR of the OTHER slots' masks is kept (`exc`), and per wave and block the nonzero
    // sum (`rsum`); per block the three costliest waves (`top`), so that the slowest wave outside {wa, wb} is a
    // lookup.  Same arithmetic, same comparisons, same result as recomputing every cost from the masks (which is
    // what this did until round 4: res5 126 ms per layer on one core per column), at a fifth of the time.
    std::vector<uint64_t> exc((size_t)nw * G * nb * words, 0ull);
    std::vector<int> rsum((size_t)nw * nb, 0);
    auto refresh_wave = [&](int w) {
      const int o = blk0 + w, n = count_of(o);
      for (int b = 0; b < nb; ++b) {
        int rc = 0;
        for (int gl = 0; gl < n; ++gl) rc += recs[(size_t)slot[(size_t)o * G + gl] * nb + b];
        rsum[(size_t)w * nb + b] = rc;
        for (int ge = 0; ge < n; ++ge) {
          uint64_t *e = &exc[(((size_t)w * G + ge) * nb + b) * words];
          for (int wd = 0; wd < words; ++wd) e[wd] = 0ull;
          for (int gl = 0; gl < n; ++gl) {
            if (gl == ge) continue;
            const uint64_t *mk = &mask[((size_t)slot[(size_t)o * G + gl] * nb + b) * words];
            for (int wd = 0; wd < words; ++wd) e[wd] |= mk[wd];
          }
        }
      }
    };
    for (int w = 0; w < nw; ++w) refresh_wave(w);
    std::vector<int> top((size_t)nb * 3, -1);
    auto refresh_top = [&]() {
      for (int b = 0; b < nb; ++b) {
        int i0 = -1, i1 = -1, i2 = -1;
        for (int w = 0; w < nw; ++w) {
          const double c = cost[(size_t)w * nb + b];
          if (i0 < 0 || c > cost[(size_t)i0 * nb + b]) { i2 = i1; i1 = i0; i0 = w; }
          else if (i1 < 0 || c > cost[(size_t)i1 * nb + b]) { i2 = i1; i1 = w; }
          else if (i2 < 0 || c > cost[(size_t)i2 * nb + b]) i2 = w;
        }
        top[(size_t)b * 3] = i0; top[(size_t)b * 3 + 1] = i1; top[(size_t)b * 3 + 2] = i2;
      }
    };
    refresh_top();
    double best = 0;
    for (int b = 0; b < nb; ++b) best += std::max(0.0, cost[(size_t)top[(size_t)b * 3] * nb + b]);
    std::vector<double> ca(nb), cb(nb);
    for (int pass = 0; pass < 3; ++pass) {
      bool improved = false;
      for (int wa = 0; wa < nw; ++wa)
        for (int ga = 0; ga < count_of(blk0 + wa); ++ga)
          for (int wb = wa + 1; wb < nw; ++wb)
            for (int gb = 0; gb < count_of(blk0 + wb); ++gb) {
              uint32_t &sa = slot[(size_t)(blk0 + wa) * G + ga], &sb = slot[(size_t)(blk0 + wb) * G + gb];
              const size_t ma = sa, mb = sb;      // channel ma leaves wave wa for wb, mb the other way
              double obj = 0;
              for (int b = 0; b < nb; ++b) {
                const uint64_t *ea = &exc[(((size_t)wa * G + ga) * nb + b) * words], *eb = &exc[(((size_t)wb * G + gb) * nb + b) * words];
                const uint64_t *ka = &mask[(ma * nb + b) * words], *kb = &mask[(mb * nb + b) * words];
                int ra = 0, rb = 0;
                for (int wd = 0; wd < words; ++wd) {
                  ra += __builtin_popcountll(ea[wd] | kb[wd]);
                  rb += __builtin_popcountll(eb[wd] | ka[wd]);
                }
                const int da = recs[mb * nb + b] - recs[ma * nb + b];
                ca[b] = kGroupCost * ra + kRecordCost * (rsum[(size_t)wa * nb + b] + da);
                cb[b] = kGroupCost * rb + kRecordCost * (rsum[(size_t)wb * nb + b] - da);
                // the slowest wave of the block: the two that changed, or the costliest of the others
                double mx = std::max(std::max(0.0, ca[b]), cb[b]);
                for (int k = 0; k < 3; ++k) {
                  const int w = top[(size_t)b * 3 + k];
                  if (w < 0) break;
                  if (w == wa || w == wb) continue;
                  mx = std::max(mx, cost[(size_t)w * nb + b]);
                  break;
                }
                obj += mx;
              }
              if (obj < best * (1.0 - 1e-9)) {
                best = obj;
                std::swap(sa, sb);
                for (int b = 0; b < nb; ++b) {
                  cost[(size_t)wa * nb + b] = ca[b];
                  cost[(size_t)wb * nb + b] = cb[b];
                }
                refresh_wave(wa);
                refresh_wave(wb);
                refresh_top();
                improved = true;
              }
            }
      if (!improved) break;
    }
  }
  };
  // WeightAlign is host time the caller waits for (res5: 70 ms of channel deal in one thread): the columns are
  // independent, so large layers deal them on up to eight threads (sixteen measured no faster).  Deterministic: a column's result does not
  // depend on the others'.
  std::vector<int> cols;
  for (int blk0 = 0; blk0 < n_ocg; blk0 += t.oc_waves) cols.push_back(blk0);
  const size_t work = (size_t)Mg * nb * words;
  const size_t hw = (size_t)allowed_cores();     // (the cores this thread may use, not the machine's: thread_place.h)
  const size_t n_thr = work >= 4096 && cols.size() > 1 ? std::min<size_t>(std::min<size_t>(8, hw), cols.size()) : 1;
  // (parallel_for.h: the calling thread works too, a thread the system refuses is one worker fewer, an exception on
  //  any thread is rethrown here after the helpers were joined)
  parallel_for(cols.size(), n_thr, [&](size_t i) { deal_column(cols[i]); });
  // slots past the last channel of a partly filled oc-group repeat a valid channel
  return slot;
}

WeightStream build_stream(const ConvGeom &g, const Tiling &t,
                            const std::vector<std::vector<int>> &rowptr,
                            const std::vector<std::vector<int>> &colidx,
                            const std::vector<std::vector<float>> &values) {
  WeightStream ws;
  const size_t n_units = (size_t)g.group * t.n_ocg * t.n_icb;
  ws.unit_hdr.assign(n_units * kUnitHdrDwords, 0u);
  const int rows_per_blk = t.icb * g.KH;
  std::vector<std::vector<Rec>> rows(rows_per_blk);   // records per (ic_local, kr); idx = quad
  auto f2u = [](float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return u;
  };
  ws.chan.reserve((size_t)g.group * t.n_ocg * t.G);
  for (int cg = 0; cg < g.group; ++cg) {
    const std::vector<uint32_t> sl = balance_channels(g, t, rowptr[cg], colidx[cg]);
    ws.chan.insert(ws.chan.end(), sl.begin(), sl.end());
  }
  for (int cg = 0; cg < g.group; ++cg)
    for (int ocg = 0; ocg < t.n_ocg; ++ocg)
      for (int blk = 0; blk < t.n_icb; ++blk) {
        for (auto &r : rows) r.clear();
        const int ic_lo = blk * t.icb, ic_hi = std::min(g.Cg, ic_lo + t.icb);
        for (int gl = 0; gl < t.G; ++gl) {
          if (ocg * t.G + gl >= g.Mg) break;       // (slots are filled in order; the rest are empty)
          const int m = (int)ws.chan[((size_t)cg * t.n_ocg + ocg) * t.G + gl];
          for (int j = rowptr[cg][m]; j < rowptr[cg][m + 1]; ++j) {
            const int col = colidx[cg][j];
            const int kc = col % g.KW, kr = (col / g.KW) % g.KH, ic = col / (g.KW * g.KH);
            if (ic < ic_lo || ic >= ic_hi) continue;
            Rec rec;
            rec.val = values[cg][j];
            rec.idx = (uint8_t)(gl * g.KW + kc);
            rows[(ic - ic_lo) * g.KH + kr].push_back(rec);
          }
        }
        std::vector<Group> groups;
        for (int r = 0; r < rows_per_blk; ++r) {
          const std::vector<Rec> &rr = rows[r];
          const int icl = r / g.KH, kr = r % g.KH;
          const uint32_t off = (uint32_t)(((size_t)icl * t.plane_ch_floats + (size_t)kr * t.nseg * t.RS) * 4);
          for (size_t b = 0; b < rr.size(); b += kMaxSlots) {
            Group gr;
            gr.lds_off = off;
            gr.recs.assign(rr.begin() + b, rr.begin() + std::min(rr.size(), b + kMaxSlots));
            groups.push_back(gr);
          }
        }
        std::stable_sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) {
          return a.recs.size() > b.recs.size();
        });
        const size_t ui = ((size_t)cg * t.n_ocg + ocg) * t.n_icb + blk;
        uint32_t *hdr = &ws.unit_hdr[ui * kUnitHdrDwords];
        const int tg = (int)groups.size();
        auto row32 = [&](int k) -> uint32_t {
          const uint32_t r = k < tg ? groups[k].lds_off / 32u : 0u;
          if (r >= 2048u) ws.overflow = true;   // does not fit the 11-bit field: the caller rejects the tiling
          return r;
        };
        auto first = [&](int k) -> uint32_t { return k < tg ? 4u * groups[k].recs[0].idx : 0u; };
        hdr[0] = first(0) | (row32(0) << 8) | (row32(1) << 21);
        for (int n = kMaxSlots; n >= 1; --n) {
          int cum = 0;
          for (const Group &gr : groups) cum += ((int)gr.recs.size() >= n) ? 1 : 0;
          hdr[1 + (kMaxSlots - n)] = (uint32_t)cum;
        }
        hdr[7] = (uint32_t)(ws.words.size() * 4);
        const size_t body = ws.words.size();
        // the body starts with a copy of the header: it reaches the wave's staging area with the
        // quads, one block ahead, and is read from there (a scalar load per block sat on the
        // critical path)
        ws.words.insert(ws.words.end(), hdr, hdr + kUnitHdrDwords);
        for (int k = 0; k < tg; ++k) {
          const Group &gr = groups[k];
          const int n = (int)gr.recs.size();
          uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          q[0] = (row32(k + 2) << 21) | first(k + 1);
          if (n > 1) q[0] |= 4u * gr.recs[1].idx << 7;
          if (n > 2) q[0] |= 4u * gr.recs[2].idx << 14;
          for (int s = 0; s < n && s < 3; ++s) q[1 + s] = f2u(gr.recs[s].val);
          for (int s = 3; s < n; ++s) {
            q[4] |= 4u * gr.recs[s].idx << (7 * (s - 3));
            q[5 + (s - 3)] = f2u(gr.recs[s].val);
          }
          ws.words.insert(ws.words.end(), q, q + (n > 3 ? 8 : 4));
          ws.n_records += n;
        }
        ws.n_groups += tg;
        ws.max_body_bytes = std::max(ws.max_body_bytes, (int)((ws.words.size() - body) * 4));
      }
  // the staging copy of the last unit reads up to one staging area past its start
  ws.words.resize(ws.words.size() + (size_t)stage_bytes_for(ws.max_body_bytes) / 4, 0u);
  return ws;
}

}  // namespace escoin
