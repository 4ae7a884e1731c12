// stream_builder.h -- host-side construction of the tiled kernel's tiling and weight stream.
// Pure C++ (no HIP): WeightAlign's MI355X-specific half, unit-testable on a CPU-only box.
//
// The tiled kernel (sconv_tiled.hip) computes, per workgroup, an output block of
//   (oc_waves * G output channels) x (pix_waves * 64 output "quads" of 4 adjacent pixels)
// Input planes are staged in LDS per block of `icb` input channels in a zero-padded layout:
//   lds[ic_local][segment][plane_row][RS floats]     RS = 4*S4 >= W, columns >= W are zero
// A lane owns one quad (plane row, columns 4j..4j+3).  For every (ic_local, kernel row kr) --
// an "input row" -- the wave reads its aligned quad ONCE and applies every nonzero of its G
// output channels that lives in that input row.  The kernel-column shift kc is NOT applied
// to the input: by linearity a separate accumulator class is kept per kc and the classes
// are shifted and summed once in the epilogue.
//
// Weight stream, one "unit" per (conv group, oc-group of G channels, input-channel block):
//   chunk 0            header: [0] = number of groups, [n] (n=1..7) = END_n
//   chunk 1..Tg        one group each = one input row's records, 16 dwords:
//                        [0] LDS byte offset of the row (ic_local*plane_ch_bytes + kr*RS*4)
//                        [1] record count n (informational)
//                        [2+2s], [3+2s]   slot s = (value bits, M0 word), s = 0..6; the n
//                                         records are RIGHT-aligned (slots 7-n..6)
//   chunk Tg+1, Tg+2   zero chunks (the kernel prefetches two chunks ahead)
// Groups are sorted by n descending; END_n = 64*(#groups with count >= n + 3) is the byte
// offset the kernel's prefetch cursor has when bucket n is exhausted.
// M0 word = 0xC000 | 4*(g_local*KW + kc): GPR-index mode with VDST and VSRC2 relative.
#ifndef ESCOIN_STREAM_BUILDER_H_
#define ESCOIN_STREAM_BUILDER_H_

#include <cstdint>
#include <vector>

namespace escoin {

constexpr int kChunkDwords = 16;
constexpr int kMaxSlots = 7;        // records per group
constexpr int kMaxAccRegs = 192;    // accumulator VGPRs owned by the asm loop
constexpr unsigned kM0Mode = 0xC000u;

struct ConvGeom {
  int N, C, H, W, M, KH, KW, pad_h, pad_w, group;
  int OH, OW, Cg, Mg;
};

struct Tiling {
  bool ok = false;
  int KW = 0, KH = 0;
  int S4 = 0;            // quads per LDS row (power of two), RS = 4*S4 floats
  int RS = 0;
  int rows_per_wave = 0; // 64 / S4
  int pix_waves = 0, oc_waves = 0, waves = 0;
  int G = 0;             // output channels per wave
  int n_ocg = 0;         // oc-groups per conv group = ceil(Mg / G)
  int n_ocblk = 0;       // workgroup columns per conv group = ceil(n_ocg / oc_waves)
  int tr = 0;            // output rows per segment
  int nseg = 0;          // segments (whole images) per workgroup; 1 in band mode
  bool band_mode = false;// true: a workgroup covers `tr` rows of ONE image
  int bands = 0;         // bands per image (band mode) else 1
  int plane_rows = 0;    // tr + KH - 1
  int plane_seg_floats = 0;  // plane_rows * RS
  int plane_ch_floats = 0;   // nseg * plane_seg_floats
  int icb = 0, n_icb = 0;    // input channels per LDS block, blocks per conv group
  int lds_bytes = 0;
};

// Picks the tiling for a geometry; .ok == false when the tiled kernel does not apply
// (stride/dilation != 1 are filtered by the caller; here: KW > 5, W > 256, ...).
Tiling choose_tiling(const ConvGeom &g, int waves_per_wg, int lds_budget_bytes);

struct WeightStream {
  std::vector<uint32_t> words;     // all units, chunk-aligned
  std::vector<int32_t> unit_off;   // [group][n_ocg][n_icb] -> dword offset of the unit header
  long n_groups = 0, n_records = 0, n_slots = 0;
};

// rowptr/colidx/values: per conv group CSR with UNSTRETCHED columns ic*KH*KW + kr*KW + kc.
WeightStream build_stream(const ConvGeom &g, const Tiling &t,
                          const std::vector<std::vector<int>> &rowptr,
                          const std::vector<std::vector<int>> &colidx,
                          const std::vector<std::vector<float>> &values);

}  // namespace escoin
#endif
