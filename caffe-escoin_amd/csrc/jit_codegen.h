// jit_codegen.h -- WeightAlign's code generator: the sparsity pattern compiled into gfx950 code.
//
// The weight stream of the LDS-staged kernel (stream_builder.h) is DATA that a generic loop
// decodes at run time: per input row a meta word through v_readfirstlane, a row offset and up to
// six accumulator indices extracted with scalar shifts, the accumulator selected through M0
// (GPR-index mode) -- 13.7 instructions around the 10.4 packed FMAs of an average row at 90 %
// sparsity, and the walk is bound by instruction issue.  A pruned layer's pattern
// is fixed once WeightAlign has run (the reference builds its CSR exactly once per weight load,
// base_conv_layer.cpp:46-273), so here WeightAlign goes one step further and emits the walk of
// every (conv group, oc-group, input-channel block) unit as STRAIGHT-LINE machine code:
//
//   per nonempty input row      ds_read_b128 x2     the lane's tile-A / tile-B quads at an
//                                                  IMMEDIATE LDS offset (no address arithmetic)
//                               s_waitcnt lgkmcnt  counted: rows are read two ahead
//   per nonzero                 s_mov_b32 s, value  the weight as a 32-bit literal
//                               v_pk_fma_f32 x4     into STATICALLY numbered accumulators
//   at the end of the unit      s_setpc_b64 s[30:31]
//
// = 5 instructions per nonzero + 2.x per row, no meta words, no index mode, no LDS staging of the
// weights (the nonzero stream reaches the wave through instruction fetch: one cache line of code
// carries 1.6 nonzeros; every workgroup column runs the same code, so it comes from L2).  The
// code is megabytes per layer and never repeats inside a tile; tools/probes/gen_probe_istream.py
// measured that instruction mix at 95-104 packed-FMA TFLOP/s from 8 MiB of straight-line code
// per workgroup, every wave in its own part, against 52 in the decoded walk.
//
// Pure C++ (no HIP): unit-testable on a CPU-only box (tests/cpp/emulate_tiled.cpp interprets the
// generated code; tests/test_jit_codegen.py checks the encoders against llvm-mc).
//
// Register contract with sconv_tiled.hip (the compiled part owns v[0:31]):
//   v32  LDS byte address of the lane's tile-A quad (plane row 0, channel 0 of the block being
//        walked), v33 = v32 + 1024 (tile B)
//   v[36:43], v[44:51], v[52:59]   three sets of input quads (A: +0..3, B: +4..7)
//   v[64:159] / v[160:255]         tile-A / tile-B accumulators, as in the LDS-staged kernel
//   s40 / s42 (pairs s[40:41], s[42:43])   weights, alternating
//   s[30:31] return address (the unit is entered with s_swappc_b64)
// Plane DMA from inside the code (DmaPlan, layers with one wave per oc-group): while a unit walks
// block k, ITS wave's share of the LDS-DMA instructions that stage block k + 1 (or the next tile's
// block 0) is part of the unit's code, spread over its first rows -- in a burst at the block top an
// LDS-DMA instruction holds its wave for ~100 cycles, among FMAs for 5-20 -- with
// everything static: which 1 KiB piece, its LDS address, its channel (the scalar offset), its
// table entry (a ds_read_b32 at an immediate offset).  Extra registers:
//   v34  LDS byte address of this lane's entry 0 in the quad table of the tile being STAGED
//   v35, v60, v61   table entries (in rotation), s[44:47] the bottom blob's buffer descriptor (zero
//   records when nothing is left to stage), s48 LDS byte address of the buffer being filled, s49 scratch
#ifndef ESCOIN_JIT_CODEGEN_H_
#define ESCOIN_JIT_CODEGEN_H_

#include <cstdint>
#include <vector>

#include "stream_builder.h"

namespace escoin {
namespace jit {

constexpr int kVAddrA = 32, kVAddrB = 33;
constexpr int kVIn0 = 36;          // input sets at 36, 44, 52
constexpr int kInSets = 3;
constexpr int kAccA = 64, kAccB = 160;
constexpr int kSWeight0 = 40, kSWeight1 = 42;
constexpr int kVTabAddr = 34, kVTab0 = 35, kVTab1 = 60;
constexpr int kSRsrc = 44, kSFillBase = 48, kSSoff = 49;
constexpr int kUnitAlign = 64;     // bytes: a unit starts on an instruction-cache line

// ---- instruction encoders (gfx950; checked against llvm-mc in tests/test_jit_codegen.py) ----
// ds_read_b128 v[vdst:vdst+3], v<vaddr> offset:<off>
inline void enc_ds_read_b128(std::vector<uint32_t> &c, int vdst, int vaddr, unsigned off) {
  c.push_back(0xD9FE0000u | (off & 0xFFFFu));
  c.push_back(((uint32_t)vdst << 24) | (uint32_t)vaddr);
}
// s_waitcnt lgkmcnt(n)   (vmcnt, expcnt not waited for)
inline void enc_waitcnt_lgkm(std::vector<uint32_t> &c, int n) { c.push_back(0xBF8CC07Fu | ((uint32_t)(n & 15) << 8)); }
// s_mov_b32 s<sdst>, <32-bit literal>
inline void enc_s_mov_lit(std::vector<uint32_t> &c, int sdst, uint32_t lit) {
  c.push_back(0xBE8000FFu | ((uint32_t)sdst << 16));
  c.push_back(lit);
}
// v_pk_fma_f32 v[acc:acc+1], s[sw:sw+1], v[x:x+1], v[acc:acc+1] op_sel_hi:[0,1,1]
// (both halves multiply by the LOW dword of the SGPR pair: one weight, two pixels)
inline void enc_pk_fma(std::vector<uint32_t> &c, int acc, int sw, int x) {
  c.push_back(0xD3B04000u | (uint32_t)acc);
  c.push_back((uint32_t)sw | ((256u + (uint32_t)x) << 9) | ((256u + (uint32_t)acc) << 18) | (2u << 27));
}
// ds_read_b32 v<vdst>, v<vaddr> offset:<off>
inline void enc_ds_read_b32(std::vector<uint32_t> &c, int vdst, int vaddr, unsigned off) {
  c.push_back(0xD86C0000u | (off & 0xFFFFu));
  c.push_back(((uint32_t)vdst << 24) | (uint32_t)vaddr);
}
// s_add_u32 m0, s<ssrc>, <32-bit literal>
inline void enc_s_add_m0_lit(std::vector<uint32_t> &c, int ssrc, uint32_t lit) {
  c.push_back(0x807CFF00u | (uint32_t)ssrc);
  c.push_back(lit);
}
constexpr int kSExecLo = 0x7E, kSExecHi = 0x7F;          // s_mov_b32 exec_lo / exec_hi through enc_s_mov_lit
inline void enc_exec_all(std::vector<uint32_t> &c) { c.push_back(0xBEFE01C1u); }       // s_mov_b64 exec, -1
// buffer_load_dwordx4 v<vaddr>, s[srsrc:srsrc+3], s<soff> offen [nt] lds   (LDS address in M0)
inline void enc_lds_dma16(std::vector<uint32_t> &c, int vaddr, int srsrc, int soff, bool nt) {
  c.push_back(nt ? 0xE05F1000u : 0xE05D1000u);
  c.push_back((uint32_t)vaddr | ((uint32_t)(srsrc >> 2) << 16) | ((uint32_t)soff << 24));
}
inline void enc_getpc(std::vector<uint32_t> &c, int sdst) { c.push_back(0xBE801C00u | ((uint32_t)sdst << 16)); }   // s_getpc_b64 s[sdst:sdst+1]
// s_add_u32 s<sd>, s<sd>, <literal> ; s_addc_u32 s<sd+1>, s<sd+1>, 0 | -1
inline void enc_s_add_lit(std::vector<uint32_t> &c, int sd, uint32_t lit) {
  c.push_back(0x8000FF00u | ((uint32_t)sd << 16) | (uint32_t)sd);
  c.push_back(lit);
}
inline void enc_s_addc(std::vector<uint32_t> &c, int sd, bool minus_one) {
  c.push_back(0x82000000u | ((uint32_t)sd << 16) | ((minus_one ? 0xC1u : 0x80u) << 8) | (uint32_t)sd);
}
// global_load_dword v<vdst>, v<vaddr>, s[saddr:saddr+1]
inline void enc_global_load_dword(std::vector<uint32_t> &c, int vdst, int vaddr, int saddr) {
  c.push_back(0xDC508000u);
  c.push_back((uint32_t)vaddr | ((uint32_t)saddr << 16) | ((uint32_t)vdst << 24));
}
inline void enc_setpc_return(std::vector<uint32_t> &c) { c.push_back(0xBE801D1Eu); }   // s_setpc_b64 s[30:31]
// ---- block-to-block chaining (ChainPlan below) ----
// s_waitcnt vmcnt(n)   (expcnt, lgkmcnt not waited for; n <= 63)
inline void enc_waitcnt_vm(std::vector<uint32_t> &c, int n) {
  c.push_back(0xBF8C0F70u | (uint32_t)(n & 15) | ((uint32_t)((n >> 4) & 3) << 14));
}
inline void enc_barrier(std::vector<uint32_t> &c) { c.push_back(0xBF8A0000u); }          // s_barrier
// s_mov_b32 s<sdst>, s<ssrc>
inline void enc_s_mov(std::vector<uint32_t> &c, int sdst, int ssrc) { c.push_back(0xBE800000u | ((uint32_t)sdst << 16) | (uint32_t)ssrc); }
// s_add_u32 s<sdst>, s<s0>, <literal>
inline void enc_s_add_u32_lit(std::vector<uint32_t> &c, int sdst, int s0, uint32_t lit) {
  c.push_back(0x8000FF00u | ((uint32_t)sdst << 16) | (uint32_t)s0);
  c.push_back(lit);
}
// s_sub_u32 s<sdst>, s<s0>, s<s1>
inline void enc_s_sub_u32(std::vector<uint32_t> &c, int sdst, int s0, int s1) {
  c.push_back(0x80800000u | ((uint32_t)sdst << 16) | ((uint32_t)s1 << 8) | (uint32_t)s0);
}
// s_cmp_lt_u32 s<s0>, <literal>
inline void enc_s_cmp_lt_u32_lit(std::vector<uint32_t> &c, int s0, uint32_t lit) {
  c.push_back(0xBF0AFF00u | (uint32_t)s0);
  c.push_back(lit);
}
// s_cselect_b32 s<sdst>, s<s0>, 0
inline void enc_s_cselect_or_zero(std::vector<uint32_t> &c, int sdst, int s0) {
  c.push_back(0x85008000u | ((uint32_t)sdst << 16) | (uint32_t)s0);
}
// v_add_u32 v<vdst>, s<ssrc>, v<vsrc>
inline void enc_v_add_u32_s(std::vector<uint32_t> &c, int vdst, int ssrc, int vsrc) {
  c.push_back(0x68000000u | ((uint32_t)vdst << 17) | ((uint32_t)vsrc << 9) | (uint32_t)ssrc);
}
inline void enc_setprio(std::vector<uint32_t> &c, int p) { c.push_back(0xBF8F0000u | (uint32_t)(p & 3)); }
// ---- weights through the scalar data cache (Options::sweights below) ----
// v_pk_fma_f32 v[acc:acc+1], s[sw:sw+1], v[x:x+1], v[acc:acc+1] op_sel:[1,0,0] op_sel_hi:[1,1,1]
// (both halves multiply by the HIGH dword of the SGPR pair)
inline void enc_pk_fma_hi(std::vector<uint32_t> &c, int acc, int sw, int x) {
  c.push_back(0xD3B04800u | (uint32_t)acc);
  c.push_back((uint32_t)sw | ((256u + (uint32_t)x) << 9) | ((256u + (uint32_t)acc) << 18) | (3u << 27));
}
// s_load_dwordx16 s[sd:sd+15], s[sbase:sbase+1], <imm offset (bytes, 21 bits)>
inline void enc_s_load_x16(std::vector<uint32_t> &c, int sd, int sbase, uint32_t off) {
  c.push_back(0xC0120000u | ((uint32_t)sd << 6) | ((uint32_t)sbase >> 1));
  c.push_back(off & 0x1FFFFFu);
}
// s_branch <simm16 dwords, relative to the next instruction>
inline void enc_s_branch(std::vector<uint32_t> &c, int dwords) { c.push_back(0xBF820000u | ((uint32_t)dwords & 0xFFFFu)); }
constexpr int kSWBuf0 = 56, kSWBuf1 = 72;   // two buffers of 16 weights: s[56:71], s[72:87]
constexpr int kSWBase = 88;                 // s[88:89]: address of the unit's weight lines
constexpr int kSWLine = 16;                 // weights per line (one s_load_dwordx16, 64 bytes)
inline void enc_nop(std::vector<uint32_t> &c) { c.push_back(0xBF800000u); }
// ---- accumulators initialised by the code itself (Options::self_zero below) ----
// v_pk_mul_f32 v[acc:acc+1], s[sw:sw+1], v[x:x+1] op_sel_hi:[0,1] -- the first product of an accumulator pair
inline void enc_pk_mul(std::vector<uint32_t> &c, int acc, int sw, int x) {
  c.push_back(0xD3B14000u | (uint32_t)acc);
  c.push_back((uint32_t)sw | ((256u + (uint32_t)x) << 9) | (2u << 27));
}
// v_pk_mul_f32 v[acc:acc+1], s[sw:sw+1], v[x:x+1] op_sel:[1,0] (the HIGH dword of the SGPR pair)
inline void enc_pk_mul_hi(std::vector<uint32_t> &c, int acc, int sw, int x) {
  c.push_back(0xD3B14800u | (uint32_t)acc);
  c.push_back((uint32_t)sw | ((256u + (uint32_t)x) << 9) | (3u << 27));
}
// v_pk_mov_b32 v[acc:acc+1], 0, 0
inline void enc_pk_zero(std::vector<uint32_t> &c, int acc) {
  c.push_back(0xD3B34000u | (uint32_t)acc);
  c.push_back(0x18010080u);
}

// What the generated code must know to stage the next block's planes itself.
struct DmaPlan {
  bool on = false;
  int qpc = 0;              // quads per channel plane in LDS (Tiling::plane_ch_floats / 4)
  int period = 0;           // lcm(qpc, 64): the quad table covers this many quads = period / qpc channels
  uint32_t chan_bytes = 0;  // bytes between two channel planes of the bottom blob (H * W * 4)
  int waves = 8;            // waves sharing the fill (piece i is issued by wave i % waves)
  bool nt = false;          // non-temporal loads (the layer's input is read by one workgroup column)
  int spread_pct = 70;      // the pieces go out over the first this many percent of a unit's rows
  int ahead = 1;            // the unit of block k stages block k + ahead (plane buffers - 1: 1, or 2 when two
                            // fills are kept in flight)
};

// Block-to-block chaining: the units of one (conv group, oc-group) lie back to back in block order, and
// with this on a unit does not return to the kernel body after its block -- it waits for its own pieces of
// the next block (a counted s_waitcnt vmcnt: the count is static), joins the workgroup barrier, moves its
// LDS addresses on to the next plane buffer and FALLS THROUGH into the next block's unit; only the last
// block's unit returns.  The body enters a tile's chain once and comes back for the epilogue.  What the
// compiled block top cost (unit offset from LDS, fill bookkeeping, two integer divisions, SGPR spills:
// ~1300 cycles per block, 9-12 % of a ResNet launch, profiles/r04_stamp_profile.md) becomes ~14
// instructions around the barrier.  Needs: the code's own plane DMA (DmaPlan), every wave of every
// workgroup with an oc-group (the barriers are counted in code).  Extra registers:
//   s52  byte offset of the plane buffer being WALKED (s48: the one being filled); v32 / v33 follow it
//   s53  num_records for fills that belong to the NEXT tile (0 when there is none), moved into s46 by the
//        first unit that stages the next tile; s54 = byte distance from this tile's quad table to the next
//        tile's (added to v34 at the same point); s55 scratch
struct ChainPlan {
  bool on = false;
  int nbuf = 2;             // plane buffers
  uint32_t buf_bytes = 0;   // bytes of one plane buffer
};
constexpr int kSWalkBase = 52, kSNextRecords = 53, kSTabDelta = 54, kSChainTmp = 55, kSRecords = 46;

// Smallest period lcm(qpc', 64) <= max_period over qpc' in [qpc, qpc * (1 + slack)]: *padded = qpc'
// (0: none).  A plane may be padded by a few quads so that the table stays small.
int dma_period(int qpc, int max_period, double slack, int *padded);

// Code prefetch: a unit's code is megabytes away from being cache resident when its layer runs once
// per forward pass (a step of the ResNet set touches 2.8 GB between two launches of a layer), and
// an instruction-cache miss that goes to HBM stalls its wave for microseconds.  Every unit therefore
// starts by touching the lines of the unit its wave runs NEXT (the next block's; after the last
// block, block 0's, for the next tile) with plain loads into a dead register -- 64 lanes x 64 bytes
// per instruction -- so that they are in L2 a block later.  Position independent: s_getpc_b64 plus the
// distance, patched in once every unit's place is known.
//   v63 = lane * 64 (set by the caller), v62 dead, s[50:51] scratch
constexpr int kVPrefDead = 62, kVPrefLane = 63, kSPref = 50;

struct Options {
  int depth = 2;          // rows read ahead (1 or 2; three input sets allow 2)
  int depth_one_tile = 5; // ... in code without a tile B (six single-quad sets: up to 5; with hi_sets up to 13)
  int hi_sets = 24;       // extra single-quad input sets in v[160:255] (tile B's accumulators, where tile B has no rows)
  int hoist_weight = 1;   // s_mov of record j+1 issued before the FMAs of record j
  int prio_rows = 0;      // > 0: s_setprio alternates 1 / 0 every this many rows (0: never) ...
  int prio_waves = 0;     // ... in the units of the first this many waves of a workgroup (0: every unit): the
                          // first-dispatched half loses to a second half that runs at a constant priority 1
                          // and wins every tie against it at 1 (oldest first), so it alternates
  int ablate = 0;         // timing experiments only (ESCOIN_JIT_ABL; wrong results): 1 no FMAs, 2 no LDS
                          // reads, 4 no weight moves (+ 16384: with two nonzero weights in the registers), 8 empty
                          // units, 16 no barrier between chained units, 32 no wait for the plane DMA there, 64 no
                          // waits for LDS reads, 512 an s_nop behind every FMA.  Builds that change the DATA the FMAs
                          // see (1, 2, 4 alone) also change the chip's clock: profiles/r04_walk_limits.md
  DmaPlan dma;
  ChainPlan chain;
  int prefetch = 1;       // touch the next unit's code (above)
  int sweights = 0;       // 1: no literal move per nonzero -- a unit's weights lie in 64-byte lines behind its code (an island the
                          // chain branches over), one s_load_dwordx16 per 16 nonzeros brings a line into one of two SGPR
                          // buffers a line ahead of use, and the FMAs read their weight from the buffer (low or high half of a
                          // pair).  Scalar loads return out of order with LDS reads (one counter): the switch to a line is an
                          // s_waitcnt lgkmcnt(0), placed in front of the row's read-ahead; the counted LDS waits ignore the
                          // load in flight, which can only make them wait longer.  4.06 instead of 5 instructions and 36 instead
                          // of 40 code bytes per nonzero; for the 3x3 / 5x5 layers (their kernel instantiation pays 34 more
                          // clobbered SGPRs around the call: sconv_tiled.hip)
  int self_zero = 0;      // 1: the kernel body does not clear the accumulators at a tile's top (192 vector moves per wave and
                          // tile: 7 us of a 120 us res2 launch) -- block 0's unit does: the FIRST product of an accumulator pair is
                          // a v_pk_mul_f32 instead of an FMA onto zero (the same value; a product of -0 keeps its sign where
                          // 0 + -0 gave +0), and the pairs block 0 never touches are cleared at its top (rare: a (channel, kernel
                          // column) without a nonzero in the block's channels)
  int one_tile = 0;       // set by build_program: the tiling leaves tile B without rows, its reads and FMAs
                          // are not generated (-1: never, ESCOIN_JIT_ONE_TILE=0)
};
Options options_from_env();

struct Program {
  std::vector<uint32_t> code;       // every unit, back to back, each aligned to kUnitAlign bytes
  std::vector<uint32_t> unit_off;   // [conv group][n_ocg][n_icb]: byte offset of the unit's entry
  std::vector<uint32_t> chan;       // slot -> output channel, as WeightStream::chan
  long n_rows = 0, n_records = 0, n_dma = 0;
  int n_pref = 0;                   // code touches (plain loads) at the start of every unit
  bool overflow = false;            // an LDS offset does not fit the instruction's 16-bit field
  bool chained = false;             // the units of an oc-group run as one chain per tile (ChainPlan)
  // balance of the channel deal (build_pass): barrier-weighted slowest / mean wave over all blocks, and the worst block
  double deal_slowest_over_mean = 1.0, deal_worst_block = 1.0;
};

Program build_program(const ConvGeom &g, const Tiling &t, const std::vector<std::vector<int>> &rowptr,
                      const std::vector<std::vector<int>> &colidx,
                      const std::vector<std::vector<float>> &values, const Options &opt);

}  // namespace jit
}  // namespace escoin
#endif
