// stream_builder.h -- host-side construction of the tiled kernel's tiling and weight stream.
// Pure C++ (no HIP): WeightAlign's MI355X-specific half, unit-testable on a CPU-only box.
//
// The tiled kernel (sconv_tiled.hip) computes, per workgroup, an output block of
//   (oc_waves * G output channels) x (pix_waves * 2 * 64 output "quads" of 4 adjacent pixels)
// Input planes are staged in LDS per block of `icb` input channels in a zero-padded layout:
//   lds[ic_local][plane_row][segment][RS floats]     RS = 4*S4 >= W, columns >= W are zero
// (segments -- the whole images of a tile -- are interleaved row by row, and a workgroup's lanes
// are numbered (row, segment, quad): the 64 quads a wave reads for one input row are then 1 KiB of
// consecutive LDS whatever the image size, so the reads are bank-conflict free and tile B is
// always tile A + 1 KiB.)
// A lane owns TWO quads (tile A and tile B: the same lane position in two consecutive
// 64-quad slabs of the workgroup's flattened (segment,row) space).  For every
// (ic_local, kernel row kr) -- an "input row" -- the wave reads its two aligned quads ONCE and
// applies every nonzero of its G output channels that lives in that input row.  The
// kernel-column shift kc is NOT applied to the input: by linearity a separate accumulator
// class is kept per kc and the classes are shifted and summed once in the epilogue.
//
// Weight stream: one "unit" per (conv group, oc-group of G channels, input-channel block); its
// layout is described at WeightStream below.
#ifndef ESCOIN_STREAM_BUILDER_H_
#define ESCOIN_STREAM_BUILDER_H_

#include <cstdint>
#include <vector>

namespace escoin {

constexpr int kTilesPerLane = 2;      // quads owned by a lane
constexpr int kAccRegsPerTile = 96;   // accumulator VGPRs per tile (192 in all)

struct ConvGeom {
  int N, C, H, W, M, KH, KW, pad_h, pad_w, group;
  int OH, OW, Cg, Mg;
  float density = 0.f;   // nonzero fraction of the weights (0: unknown); steers the tiling choice
  int sub = 1;           // strided pointwise layers (1x1, stride `sub`, no padding): H x W here is the VIEW the kernel
                         // walks -- OH rows (input rows 0, sub, 2 sub, ...) of the full input width; lanes own input quads
                         // and every sub-th element of a quad is an output.  The rows of such a view are not contiguous
                         // in memory: no re-cut, no row packing.
};

struct Tiling {
  bool ok = false;
  int KW = 0, KH = 0;
  int H = 0, W = 0, OH = 0, OW = 0;  // the image cut the kernel works on (differs from the
                                     // layer's only for re-cut pointwise layers, see choose_tiling)
  int S4 = 0;            // quads per LDS row (power of two), RS = 4*S4 floats
  int RS = 0;
  int rows_per_slab = 0; // 64 / S4: output rows covered by one 64-quad slab
  int pix_waves = 0, oc_waves = 0, waves = 0;
  int G = 0;             // output channels per wave
  int n_ocg = 0;         // oc-groups per conv group = ceil(Mg / G)
  int n_ocblk = 0;       // workgroup columns per conv group = ceil(n_ocg / oc_waves)
  int tpl = kTilesPerLane;   // quads a lane owns: 2 (tile A, tile B), or 1 -- generated code only, pointwise
                         // layers only: the whole accumulator file serves tile A, twice the channels per wave
  int rows_per_wg = 0;   // pix_waves * tpl * rows_per_slab flattened rows
  int tr = 0;            // output rows per segment
  int nseg = 0;          // segments (whole images) per workgroup; 1 in band mode
  bool band_mode = false;// true: a workgroup covers `tr` rows of ONE image
  int bands = 0;         // bands per image (band mode) else 1
  int plane_rows = 0;    // tr + KH - 1
  int plane_seg_floats = 0;  // plane_rows * RS
  int plane_ch_floats = 0;   // nseg * plane_seg_floats
  int icb = 0, n_icb = 0;    // input channels per LDS block, blocks per conv group
  int planes_bytes = 0;      // icb * plane_ch_floats * 4
};

// Picks the tiling for a geometry; .ok == false when the tiled kernel does not apply
// (stride/dilation != 1 are filtered by the caller; here: KW > 5, W > 256, ...).
// lds_budget_bytes bounds the input planes only; the stream region is added on top.
// n_cu: compute units of the device.  Among the candidate tilings (passes over the output
// channels x images per workgroup) the one with the lowest estimated launch time is taken.
// one_tile_ok: tilings with one quad per lane may be chosen (the caller runs generated code).
Tiling choose_tiling(const ConvGeom &g, int waves_per_wg, int lds_budget_bytes, int n_cu = 256, bool one_tile_ok = false);

// The same tiling with channel planes of `qpc` quads in LDS (>= the tiling's own: padding quads are
// staged like rows past the image) and the input-channel blocks recomputed for the budget.
Tiling repad_planes(const ConvGeom &g, Tiling t, int qpc, int lds_budget_bytes);

// ---- the weight stream: staged in LDS with the planes, values read as broadcast quads ---------
//
// One unit per (conv group, oc-group, input-channel block), in two parts:
//   unit_hdr[8 * unit + ...]   (also the first 32 bytes of the body: the kernel reads it from the
//                               staging area; the table itself is read once, for block 0's offset)
//     [0]     what groups 0 and 1 need before any quad has been read: accumulator of group 0's
//             record 0 (bits 0..6), row offset / 32 of group 0 (8..18) and of group 1 (21..31)
//     [1..6]  END_6, END_5, ..., END_1: END_n = number of groups with >= n records (the groups of
//             a unit are sorted by record count, descending; END_1 = number of groups)
//     [7]     byte offset of the unit's body in `words`
//   body (copied into the wave's LDS staging area by LDS-DMA one block ahead): the 8 header
//   dwords, then
//     per group one quad [meta, v0, v1, v2]; groups with more than 3 records a second quad
//     [meta2, v3, v4, v5]; the kernel reads them as broadcast ds_read_b128 and feeds the values to
//     v_pk_fma_f32 straight from the VGPR pair (op_sel picks the half).
//     meta  = (row offset of the group AFTER the next) / 32      bits 21..31
//           | accumulator of this group's record 2, 1            bits 14..20, 7..13
//           | accumulator of record 0 of the NEXT group          bits 0..6
//     meta2 = accumulators of records 3, 4, 5                    bits 0..6, 7..13, 14..20
//     accumulator = 4 * (g_local * KW + kc): the VGPR offset GPR-index mode adds, ready for
//     s_set_gpr_idx_idx (which reads bits 0..7 of its operand; bit 7 is always a zero low bit of
//     the neighbouring field) directly or after ONE scalar shift -- every instruction a wave
//     issues costs it an issue slot at 2 waves/SIMD, scalar ones included.
// A group's row offset travels two groups ahead and its first accumulator one group ahead: the LDS
// reads of group k+1's input quads are issued at the very top of group k (a whole group of FMA
// work to land in) and nothing on the path to a group's first FMA waits for its own quad's meta.
constexpr int kMaxSlots = 6;
constexpr int kUnitHdrDwords = 8;

struct WeightStream {
  std::vector<uint32_t> words;      // bodies back to back, 16-byte aligned
  std::vector<uint32_t> unit_hdr;   // [group][n_ocg][n_icb][8]
  int max_body_bytes = 0;
  long n_groups = 0, n_records = 0;
  bool overflow = false;            // an LDS row offset did not fit its field (plane buffer > 64 KiB)
  // Which output channel (within its conv group) sits in accumulator slot gl of oc-group ocg:
  // chan[(group * n_ocg + ocg) * G + gl].  The identity (ocg * G + gl) unless the channels were
  // re-dealt over the waves (balance_channels); slots past the group's last channel repeat it.
  std::vector<uint32_t> chan;
};

// Deals the output channels of one conv group over the waves (oc-groups) so that, block by
// block, the waves of a workgroup have about the same stream to walk: every block ends in a
// barrier, and with channels in their natural order the slowest wave of a block is ~9 % over the
// mean at 90 % random sparsity (much more for a pruned model whose channels differ in density).
// Cost of a wave in a block = kGroupCost * nonempty input rows + kRecordCost * nonzeros (the
// instruction counts of the stream loop).  Greedy pairwise swaps between waves of the same
// workgroup column, deterministic.  Returns slot -> channel for the group (n_ocg * G entries).
// (group_cost / record_cost: instructions per nonempty input row / per nonzero of the walk that will
// run the deal -- 12.3 / 5.75 for the LDS-staged stream loop, 2.5 / 5 for generated code.)
std::vector<uint32_t> balance_channels(const ConvGeom &g, const Tiling &t, const std::vector<int> &rowptr,
                                       const std::vector<int> &colidx, double group_cost = 12.3,
                                       double record_cost = 5.75);

WeightStream build_stream(const ConvGeom &g, const Tiling &t,
                            const std::vector<std::vector<int>> &rowptr,
                            const std::vector<std::vector<int>> &colidx,
                            const std::vector<std::vector<float>> &values);

// Bytes of one wave's staging area for a stream whose largest body is max_body_bytes: the body is
// copied in 1 KiB DMA steps and the loop prefetches one quad past the last group.
inline int stage_bytes_for(int max_body_bytes) { return (max_body_bytes + 32 + 1023) / 1024 * 1024; }

}  // namespace escoin
#endif
