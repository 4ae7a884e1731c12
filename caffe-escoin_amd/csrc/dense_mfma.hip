// dense_mfma.hip -- the dense fallback: implicit-GEMM convolution on the fp32 matrix cores.
//
// The reference sends a layer whose group-0 density exceeds 0.2 to im2col + cublasSgemm
// (forward_gpu_gemm, src/caffe/layers/base_conv_layer.cpp:713-746, gate :750-755, :805-811), and
// `caffe test -conv_mode 0` (LOWERED_GEMM) sends every layer there.  Here the column matrix is never
// materialised: per conv group
//     C[Mg x P] = A[Mg x K] * B[K x P],   K = Cg*KH*KW,  P = N*OH*OW,
// A = the dense weights, B = the im2col view gathered on the fly (any stride / pad / dilation).
//
// Persistent workgroups (two per CU, 4 waves each) walk the (pixel tile, channel tile) pairs; a
// BM x 128 output tile (BM = 128: waves 2 x 2, 64 x 64 each; BM = 64 for layers with <= 64 output
// channels per group: waves 1 x 4, 64 x 32 each) is 2 x {2,1} blocks of v_mfma_f32_32x32x2_f32
// per wave (exact fp32 products and sums).  The loop runs over k-steps of 32 ACROSS tiles: while
// step s is multiplied, the operands of step s + 1 -- the next tile's first step included -- are
// copied global -> LDS by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write), one
// barrier per step.
//   A tile  [m][32 k], 16-byte chunks XOR-swizzled by (m >> 1) & 7 (a DMA lane simply fetches the
//           chunk that belongs in its slot): a lane's four consecutive k of one row come out as
//           ONE conflict-free ds_read_b128.  The MFMA's two k-slots (lane halves) take
//           k0 + 4h + t, t = 0..3: the sum over k is just taken in another order.
//   B tile  [32 k][128 p].  Pointwise layers (1x1, stride 1, no padding, H*W % 4 == 0): the column
//           matrix IS the bottom blob, 16 bytes per DMA lane.  Otherwise 4 bytes per lane, gathered:
//           the decode k -> (ic, kr, kc) is a per-layer table built at WeightAlign (element offset
//           inside the image, (dy, dx) for the border test) read with scalar loads -- k is
//           wave-uniform; a tap outside the image gets an out-of-range offset and the buffer
//           descriptor's range check delivers the zero of the padding.
// Bias and ReLU are fused in the epilogue.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <vector>

#include "escoin_plan.h"
#include "knobs.h"

#ifdef ESCOIN_ABLATIONS
#define ESC_DENSE_ABL(a, bits) ((a).abl & (bits))
// In-kernel stamp profile (ESCOIN_PROF=1 on the ablation flavour; round 6, profiles/r06_dense_stamps.md): shader cycles of
// every wave by phase, summed over its tiles and k-steps --
//   0 start-up (arguments, first tile's addressing, first fetch)    1 k-step top: wait for this wave's operand pieces
//   2 k-step top: workgroup barrier                                 3 issuing the next k-step's fetch (LDS-DMA / gather)
//   4 fragment reads + MFMAs                                        5 between tiles: coordinates, accumulator reset
//   6 epilogue: bias, transpose through LDS, stores                 7 stream-K hand-over / fix-up
#define ESC_DPROF_DECL int dpe_ = 0; unsigned long long dpt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dpl_ = __builtin_readcyclecounter(); \
  const unsigned long long dpc0_ = dpl_, dpr0_ = __builtin_amdgcn_s_memrealtime();
#define ESC_DPROF(i) do { if (a.prof) { const unsigned long long n_ = __builtin_readcyclecounter(); dpt_[i] += n_ - dpl_; dpl_ = n_; } } while (0)
// ... and, for the step between a CU's two workgroups (profiles/r06_dense_stamps.md section 5): the 100 MHz times at which the
// epilogues of a workgroup's first 24 tiles start and end (wave 0)
#define ESC_DPROF_EPI(which) do { if (a.prof && threadIdx.x == 0 && dpe_ < 24) { a.prof[(size_t)68 * 4096 + ((size_t)blockIdx.x * 24 + dpe_) * 2 + (which)] = __builtin_amdgcn_s_memrealtime(); if (which) ++dpe_; } } while (0)
#define ESC_DPROF_DUMP                                                                                              \
  if (a.prof && (threadIdx.x & 63) == 0) {                                                                          \
    for (int i_ = 0; i_ < 10; ++i_) a.prof[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + i_] = dpt_[i_];     \
    if (threadIdx.x == 0) {                                                                                         \
      a.prof[(size_t)64 * 4096 + 4 * (size_t)blockIdx.x] = __builtin_readcyclecounter() - dpc0_;                    \
      a.prof[(size_t)64 * 4096 + 4 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - dpr0_;            \
      a.prof[(size_t)64 * 4096 + 4 * (size_t)blockIdx.x + 2] = dpr0_;                                               \
    }                                                                                                               \
  }
#else
#define ESC_DENSE_ABL(a, bits) (0)
#define ESC_DPROF_DECL
#define ESC_DPROF(i)
#define ESC_DPROF_EPI(which)
#define ESC_DPROF_DUMP
#endif

namespace escoin {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4d __attribute__((ext_vector_type(4)));

constexpr int kBN = 128, kBK = 32;

struct DenseArgs {
  const float *__restrict__ in;
  const float *__restrict__ w;     // dense (M + 128) x lda, zero-padded: rows of lda = K rounded up to 32
  const float *__restrict__ bias;
  const int2 *__restrict__ ktab;   // [K rounded up to 32]: {element offset of tap k in the image, dy | dx << 16}
  float *__restrict__ out;
  int n_images, C, H, W, M, OH, OW;
  int pad_h, pad_w, stride_h, stride_w;
  int Cg, Mg, K, lda, P, relu;
  int n_ptiles, n_mtiles, n_groups;        // tiles: pixel x channel x conv group (of this launch)
  unsigned in_bytes, w_bytes;              // buffer descriptor ranges
  int vec_out;                             // OH*OW % 4 == 0 and top 16-byte aligned: 16-byte stores through LDS
  int s2_pair;                             // BMODE 2: a lane's two outputs come out of one aligned 16-byte quad (stride 2, even OW)
  unsigned long long group_mask;           // conv groups this launch covers (all ones: every group)
  int abl;                                 // ESCOIN_ABLATIONS builds: timing experiments (wrong results)
  unsigned long long *prof;                // ESCOIN_ABLATIONS builds: stamp profile [workgroup][wave][8] (ESC_DPROF)
  int fetch_slots;                         // the next k-step's LDS-DMA pieces go out between this k-step's MFMAs (1) or as a burst at its top (0)
  // stream-K (STREAMK instantiations): every workgroup takes an equal, contiguous run of (tile, k-step) units;
  // a tile cut by a run boundary is finished by the workgroup holding its last k-steps, which adds the others'
  // partial accumulators from `sk_ws` ([workgroup][wave][16 quads][64 lanes] floats x 4) once their `sk_flag`
  // words (zeroed by the launch function before every launch) carry 1
  float *sk_ws;
  unsigned *sk_flag;     // [0 .. gridDim.x): published flags; [gridDim.x]: set when a bounded spin gave up
  unsigned *sk_fail;     // the same give-up, sticky, in pinned HOST memory: the next escoin_forward on the plan fails loudly
};

__device__ __forceinline__ int a_swizzle(int row, int chunk) { return row * kBK + ((chunk ^ ((row >> 1) & 7)) << 2); }

// One LDS-DMA instruction: 64 lanes x 16 (4) bytes from buffer offsets voff + soff (range-checked
// against the descriptor: out-of-range lanes deliver zeros) to LDS bytes [lds_addr, + 1024 (256)).
// Inline asm: the compiler must not see these as pending LDS writes (it would drain the queue with
// s_waitcnt vmcnt(0) in front of the next ds_read); the kernel waits for them itself.
__device__ __forceinline__ void dma16(u32x4d rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);   // (wave-uniform by construction)
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma4(u32x4d rsrc, unsigned lds_addr, unsigned voff) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n s_nop 0\n buffer_load_dword %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}

__device__ __forceinline__ u32x4d make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long q = reinterpret_cast<unsigned long long>(p);
  u32x4d r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)q);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

typedef unsigned __attribute__((address_space(1))) gu32;

// BMODE: how the B tile (the im2col view) reaches LDS --
//   0  gathered, 4 bytes per LDS-DMA lane (any kernel / stride / padding / dilation)
//   1  pointwise (1x1, stride 1, no padding, H*W % 4 == 0): the column matrix IS the bottom blob, 16 bytes per DMA lane
//   2  (experiments flavour only: measured slower than mode 0, see launch_dense)
//      strided pointwise (1x1, stride > 1, no padding: ResNet-50's res{3,4,5}a_branch1 / branch2a): through registers.
//      A lane owns two adjacent outputs of the tile; with stride 2 and an even output width their inputs are elements
//      0 and 2 of ONE aligned 16-byte quad (a.s2_pair: one global_load_dwordx4 per k-row, the operand-side twin of
//      the sparse path's strided view, sconv_tiled.hip TiledArgs::sub), otherwise two 4-byte loads; the two values go
//      to LDS as one 8-byte write.  8 loads + 8 writes per wave and k-step where the gather issues 16 LDS-DMA
//      instructions of 256 bytes -- an LDS-DMA instruction in a burst holds its wave ~100 cycles, and 16 of them were
//      most of a k-step's 2048 MFMA cycles (profiles/r05_dense.md).
template <int WROWS, int BMODE, bool STREAMK = false>
__global__ void __launch_bounds__(256, 2) escoin_dense_mfma_kernel(DenseArgs a) {
  constexpr bool POINTWISE4 = BMODE == 1;
  constexpr bool STRIDED1 = BMODE == 2;
  constexpr int BM = 64 * WROWS;
  constexpr int WCOLS = 4 / WROWS;           // waves along the pixel axis
  constexpr int WN = kBN / WCOLS;            // columns per wave: 64 or 32
  constexpr int NB = WN / 32;                // 32-column MFMA blocks per wave
  constexpr int A_DMA = BM / 8 / 4;          // 1 KiB A pieces per wave and k-step: 4 or 2
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  // one array per buffer: A tile then B tile (the epilogue reuses a whole buffer as its staging area)
  constexpr int kBufFloats = BM * kBK + kBK * kBN;
  __shared__ __attribute__((aligned(1024))) float sAB[2][kBufFloats];
  auto sA = [&](int b) -> float * { return &sAB[b][0]; };
  auto sB = [&](int b) -> float * { return &sAB[b][BM * kBK]; };
  ESC_DPROF_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int li = lane & 31, lh = lane >> 5;
  const int ohw = a.OH * a.OW;
  const int hw = a.H * a.W;
  const int nk = (a.K + kBK - 1) / kBK;
  const long n_tiles = (long)a.n_ptiles * a.n_mtiles * a.n_groups;
  const u32x4d rA = make_rsrc(a.w, a.w_bytes), rB = make_rsrc(a.in, a.in_bytes);
  const unsigned ldsA = (unsigned)(size_t)(&sAB[0][0]), ldsB = (unsigned)(size_t)(&sAB[0][BM * kBK]);
  constexpr unsigned kBufBytes = (unsigned)kBufFloats * 4u;

  // ---- A pieces of this wave: piece i = wave + 4 q covers rows 8 i .. 8 i + 7, lane l -> row
  // 8 i + l / 8, LDS slot l % 8, which holds chunk (l % 8) ^ ((row >> 1) & 7): the same for every q
  const unsigned voffA = (unsigned)(((lane >> 3) * a.lda + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 2)) * 4);

  auto tile_coords = [&](long tile, int &cg, int &m0, int &p0) {
    const int mt = (int)(tile % a.n_mtiles);
    const long r = tile / a.n_mtiles;
    const int pt = (int)(r % a.n_ptiles);
    const int gsel = (int)(r / a.n_ptiles);
    cg = a.group_mask == ~0ull ? gsel : nth_set_bit(a.group_mask, gsel);
    m0 = mt * BM;
    p0 = pt * kBN;
  };

  // ---- B addressing of the tile being FETCHED (recomputed when the fetch moves to a new tile) ----
  //  pointwise: lane -> pixels p0 + 4 (l % 32) .. + 3 of row pair (l / 32)
  //  gathered:  lane -> pixel p0 + 64 half + l, half = 0, 1
  unsigned fb_pix[2] = {0u, 0u};     // byte offset of the lane's pixel (window origin) in the blob
  int fb_ih0[2] = {0, 0}, fb_iw0[2] = {0, 0};
  int f_cg = 0, f_m0 = 0, f_p0 = 0;
  // BMODE 2: element offsets (channel 0 of the conv group) of the lane's two outputs, the staging registers of the
  // k-step in flight and where they go
  size_t s_off[2] = {0, 0};
  float4 s_q[STRIDED1 ? 8 : 1];
  float s_d[STRIDED1 ? 16 : 1];
  int s_k0 = 0, s_buf = 0;
  bool s_pending = false;
  auto fetch_setup = [&](long tile) {
    tile_coords(tile, f_cg, f_m0, f_p0);
    if (STRIDED1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // (pairs: the lane's first output is even and the second its right neighbour, also in the tile's clamped tail)
        const int p = a.s2_pair ? min(f_p0 + 2 * lane, a.P - 2) + e : min(f_p0 + 2 * lane + e, a.P - 1);
        const int n = p / ohw, rem = p - n * ohw;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        s_off[e] = (((size_t)n * a.C + (size_t)f_cg * a.Cg) * a.H + (size_t)oh * a.stride_h) * a.W + (size_t)ow * a.stride_w;
      }
    } else if (POINTWISE4) {
      const int p = min(f_p0 + 4 * (lane & 31), a.P - 4);
      const int n = p / ohw, rem = p - n * ohw;
      fb_pix[0] = (unsigned)((((size_t)n * a.C + (size_t)f_cg * a.Cg + (lane >> 5)) * hw + rem) * 4);
    } else {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int p = min(f_p0 + 64 * half + lane, a.P - 1);
        const int n = p / ohw, rem = p - n * ohw;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        fb_ih0[half] = oh * a.stride_h - a.pad_h;
        fb_iw0[half] = ow * a.stride_w - a.pad_w;
        fb_pix[half] = (unsigned)((((long)n * a.C + (long)f_cg * a.Cg) * hw + (long)fb_ih0[half] * a.W + fb_iw0[half]) * 4);
      }
    }
  };
  auto fetch = [&](int kstep, int buf) {
    const int k0 = kstep * kBK;
    // A: rows f_m0 + 8 i .., columns k0 .. k0 + 31 of the padded matrix
    const unsigned sa = (unsigned)((((size_t)f_cg * a.Mg + f_m0) * a.lda + k0) * 4);
#pragma unroll
    for (int q = 0; q < A_DMA; ++q) {
      const int i = wave + 4 * q;
      dma16(rA, ldsA + (unsigned)buf * kBufBytes + (unsigned)(i * 1024), voffA, sa + (unsigned)(i * 8 * a.lda * 4));
    }
    if (STRIDED1) {
      // this wave's 8 rows of the k-step into registers; rows past K are read from channel K - 1 and zeroed at the
      // commit (a zero weight does not make NaNs harmless)
      s_k0 = k0;
      s_buf = buf;
      s_pending = true;
      if (a.s2_pair) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int kc = min(k0 + 8 * wave + r, a.K - 1);
          s_q[r] = *reinterpret_cast<const float4 *>(a.in + s_off[0] + (size_t)kc * hw);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int kc = min(k0 + 8 * wave + r, a.K - 1);
          s_d[2 * r] = a.in[s_off[0] + (size_t)kc * hw];
          s_d[2 * r + 1] = a.in[s_off[1] + (size_t)kc * hw];
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // (the loads stay here, ahead of the k-step's MFMAs)
    } else if (POINTWISE4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = wave + 4 * q;          // rows k0 + 2 i, k0 + 2 i + 1
        // rows past K (the last k-step when Cg % 32 != 0) are channels of the next conv group, of the
        // next image or memory past the blob: A's zero padding does not make them harmless (0 * Inf,
        // 0 * NaN = NaN), so they are staged as zeros like the gathered path's 0x7FFF taps: the marker in
        // the VGPR offset puts the address out of range whatever the scalar offset (the descriptor's range
        // check is on the sum of the two: tools/probes/probe_rsrc_range.hip, profiles/r04_probe_rsrc_range.txt)
        const unsigned vo = (k0 + 2 * i + (lane >> 5) < a.K) ? fb_pix[0] : kOOB;
        dma16(rB, ldsB + (unsigned)buf * kBufBytes + (unsigned)(i * 1024), vo, (unsigned)((size_t)(k0 + 2 * i) * hw * 4));
      }
    } else {
      // this wave's 8 rows of the k-step: their taps are 64 contiguous bytes at a wave-uniform address
      const int4 *tp = reinterpret_cast<const int4 *>(a.ktab + k0 + 8 * wave);
      int4 tt[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) tt[q] = tp[q];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int toff = (kk & 1) ? tt[kk >> 1].z : tt[kk >> 1].x;
        const int tdyx = (kk & 1) ? tt[kk >> 1].w : tt[kk >> 1].y;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int ih = fb_ih0[half] + (tdyx & 0xFFFF), iw = fb_iw0[half] + (tdyx >> 16);
          const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
          dma4(rB, ldsB + (unsigned)buf * kBufBytes + (unsigned)((8 * wave + kk) * 512 + half * 256),
               ok ? fb_pix[half] + (unsigned)(toff * 4) : kOOB);
        }
      }
    }
  };

  // The same fetch cut into 16 SLOTS, one per group of MFMAs of a k-step (kg, t): issued between the MFMAs, an LDS-DMA
  // instruction costs its wave ~60 cycles under the shadow of the matrix instructions already queued; issued as one burst
  // at the k-step's top (what `fetch` does, and what every k-step did until round 6) each holds the wave 100-270 cycles
  // with the matrix pipe draining -- 15 % (K = 64 layers) to 48 % (stride-2 gather: 20 instructions per wave and k-step)
  // of a wave's life by the stamp profile (profiles/r06_dense_stamps.md).  Pieces go out in the FIRST slots (two per
  // slot for the gather, one otherwise) so that they have most of a k-step to land before the next top's wait.
  int4 gt[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // gathered B: this wave's 8 taps of the k-step being fetched
  auto fetch_slot = [&](int kstep, int buf, int slot) {
    const int k0 = kstep * kBK;
    if (slot < A_DMA) {
      const unsigned sa = (unsigned)((((size_t)f_cg * a.Mg + f_m0) * a.lda + k0) * 4);
      const int i = wave + 4 * slot;
      dma16(rA, ldsA + (unsigned)buf * kBufBytes + (unsigned)(i * 1024), voffA, sa + (unsigned)(i * 8 * a.lda * 4));
    }
    if (POINTWISE4) {
      if (slot >= 4 && slot < 8) {
        const int i = wave + 4 * (slot - 4);
        const unsigned vo = (k0 + 2 * i + (lane >> 5) < a.K) ? fb_pix[0] : kOOB;
        dma16(rB, ldsB + (unsigned)buf * kBufBytes + (unsigned)(i * 1024), vo, (unsigned)((size_t)(k0 + 2 * i) * hw * 4));
      }
    } else if (!STRIDED1) {
      if (slot == 0) {
        const int4 *tp = reinterpret_cast<const int4 *>(a.ktab + k0 + 8 * wave);
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // (wave-uniform: scalar registers, not four vector quads kept across eight slots)
          const int4 v = tp[q];
          gt[q].x = __builtin_amdgcn_readfirstlane(v.x); gt[q].y = __builtin_amdgcn_readfirstlane(v.y);
          gt[q].z = __builtin_amdgcn_readfirstlane(v.z); gt[q].w = __builtin_amdgcn_readfirstlane(v.w);
        }
      }
      if (slot < 8) {            // tap kk = slot: both halves
        const int kk = slot;
        const int toff = (kk & 1) ? gt[kk >> 1].z : gt[kk >> 1].x;
        const int tdyx = (kk & 1) ? gt[kk >> 1].w : gt[kk >> 1].y;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int ih = fb_ih0[half] + (tdyx & 0xFFFF), iw = fb_iw0[half] + (tdyx >> 16);
          const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
          dma4(rB, ldsB + (unsigned)buf * kBufBytes + (unsigned)((8 * wave + kk) * 512 + half * 256),
               ok ? fb_pix[half] + (unsigned)(toff * 4) : kOOB);
        }
      }
    }
  };

  // BMODE 2: the staged k-step's values into its B tile (row k, columns 2 lane and 2 lane + 1: 512 consecutive bytes
  // per row and wave, conflict-free); the barrier at the next k-step's top publishes them
  auto commit = [&]() {
    if (!STRIDED1 || !s_pending) return;
    s_pending = false;
    float *dst = &sAB[s_buf][BM * kBK];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = 8 * wave + r;
      const bool live = s_k0 + row < a.K;
      float2 v;
      if (a.s2_pair) {
        v.x = live ? s_q[r].x : 0.f;
        v.y = live ? s_q[r].z : 0.f;
      } else {
        v.x = live ? s_d[2 * r] : 0.f;
        v.y = live ? s_d[2 * r + 1] : 0.f;
      }
      *reinterpret_cast<float2 *>(&dst[row * kBN + 2 * lane]) = v;
    }
  };

  f32x16 acc[2][NB];
  // Work of this workgroup.  Tile mode: tiles blockIdx.x, + gridDim.x, ..., every k-step of each.  Stream-K:
  // units [u0, u1) of the (tile, k-step) sequence -- a run starts and ends anywhere in a tile.  The run's tiles are
  // walked LAST TO FIRST: the cut tile at the run's end (whose partial sums another workgroup waits for) comes
  // first and is published at once, the cut tile at its start (which waits for the previous workgroup's partial
  // sums) last -- walked first to last, every workgroup would wait for its predecessor's whole run.
  const long U = n_tiles * nk;
  const long sk_q = STREAMK ? (U + gridDim.x - 1) / gridDim.x : 0;
  const long u0 = STREAMK ? (long)blockIdx.x * sk_q : 0;
  const long u1 = STREAMK ? min(U, u0 + sk_q) : 0;
  if (STREAMK && u0 >= u1) return;
  const long t0 = STREAMK ? u0 / nk : 0, t1 = STREAMK ? (u1 - 1) / nk : 0;
  auto seg_lo = [&](long t) { return STREAMK && t == t0 ? (int)(u0 - t0 * nk) : 0; };
  auto seg_hi = [&](long t) { return STREAMK && t == t1 ? (int)(u1 - t1 * nk) : nk; };
  const long tile_step = STREAMK ? -1 : (long)gridDim.x;
  auto in_run = [&](long t) { return STREAMK ? t >= t0 : t < n_tiles; };
  long tile = STREAMK ? t1 : (long)blockIdx.x;
  if (!in_run(tile)) return;
  fetch_setup(tile);
  fetch(seg_lo(tile), 0);
  commit();                       // (BMODE 2: the first step has no MFMAs to hide under)
  long f_tile = tile;             // tile of the step being fetched next
  int f_k = seg_lo(tile) + 1;     // ... and its k-step
  if (f_k == seg_hi(f_tile)) {
    f_tile += tile_step;
    if (in_run(f_tile)) { f_k = seg_lo(f_tile); fetch_setup(f_tile); }
  }
  int buf = 0;
  ESC_DPROF(0);
  for (; in_run(tile); tile += tile_step) {
    int cg, m0, p0;
    tile_coords(tile, cg, m0, p0);
    const int k_lo = seg_lo(tile), k_hi = seg_hi(tile);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x16{0};
    ESC_DPROF(5);
    for (int ks = k_lo; ks < k_hi; ++ks, buf ^= 1) {
      // this wave's pieces of the step have landed (and the stores of the last epilogue are out) ...
#ifdef ESCOIN_ABLATIONS
      if (!ESC_DENSE_ABL(a, 1))     // ESCOIN_DENSE_ABL: 1 no wait for the operands, 2 no operand traffic, 4 no MFMAs
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ESC_DPROF(1);
      __syncthreads();   // ... everyone's have, and everyone is done with the other buffer
      ESC_DPROF(2);
      // the next k-step's operands: as one burst here (BMODE 2 and the experiments' ESCOIN_DENSE_SLOTS=0), or slot by
      // slot between the MFMAs below
      bool do_fetch = in_run(f_tile);
#ifdef ESCOIN_ABLATIONS
      if (ESC_DENSE_ABL(a, 2)) do_fetch = false;
#endif
      const bool slotted = !STRIDED1 && a.fetch_slots;
      if (do_fetch && !slotted) fetch(f_k, buf ^ 1);
      ESC_DPROF(3);
      // fragments of k-group kg + 1 are read while the MFMAs of k-group kg run
      // lane (i, h): A rows wm * 64 + {0, 32} + i, k = 8 kg + 4 h + t; B columns wn * WN + 32 j + i
      const int ra = wm * 64 + li;
      float4 fa[2][2];
      float fb[2][4][NB];
      auto read_frags = [&](int kg, int s) {
        fa[s][0] = *reinterpret_cast<const float4 *>(&sA(buf)[a_swizzle(ra, 2 * kg + lh)]);
        fa[s][1] = *reinterpret_cast<const float4 *>(&sA(buf)[a_swizzle(ra + 32, 2 * kg + lh)]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < NB; ++j) fb[s][t][j] = sB(buf)[(8 * kg + 4 * lh + t) * kBN + wn * WN + 32 * j + li];
      };
      read_frags(0, 0);
#pragma unroll
      for (int kg = 0; kg < kBK / 8; ++kg) {
        const int s = kg & 1;
        if (kg + 1 < kBK / 8) read_frags(kg + 1, s ^ 1);
        // keep the order: next group's LDS reads first, then this group's MFMAs (left alone, the
        // scheduler sinks each read to just above its use and every 4 MFMAs wait for LDS)
        __builtin_amdgcn_sched_barrier(0);
#ifdef ESCOIN_ABLATIONS
        if (ESC_DENSE_ABL(a, 4)) continue;
#endif
        const float a0[4] = {fa[s][0].x, fa[s][0].y, fa[s][0].z, fa[s][0].w};
        const float a1[4] = {fa[s][1].x, fa[s][1].y, fa[s][1].z, fa[s][1].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (do_fetch && slotted && 4 * kg + t < 8) {
            fetch_slot(f_k, buf ^ 1, 4 * kg + t);
            __builtin_amdgcn_sched_barrier(0);      // (the piece stays in front of this group's MFMAs)
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], fb[s][t][j], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], fb[s][t][j], acc[1][j], 0, 0, 0);
          }
        }
      }
      commit();       // (BMODE 2: the next step's B rows from the staging registers into the other buffer)
      if (do_fetch) {
        if (++f_k == seg_hi(f_tile)) {
          f_tile += tile_step;
          if (in_run(f_tile)) { f_k = seg_lo(f_tile); fetch_setup(f_tile); }
        }
      }
      ESC_DPROF(4);
    }
    if (STREAMK) {
      // the lane's accumulators as 16 * NB quads: quad (i, j, r4) = acc[i][j][4 r4 .. 4 r4 + 3]
      constexpr int kQuads = 2 * NB * 4;
      constexpr size_t kWsPerWg = (size_t)4 * 16 * 64 * 4;     // floats: 4 waves x 16 quads x 64 lanes x 4
      if (k_hi < nk) {
        // a run that ends inside a tile: hand the partial sums to the workgroup holding the tile's last k-steps.
        // Write-through stores (sc1: visible beyond this XCD's L2 once they have drained), every storing wave
        // drains, the workgroup meets, ONE lane publishes (cdna_hip_programming.md Guideline 16, R1).
        float *ws = a.sk_ws + (size_t)blockIdx.x * kWsPerWg + (size_t)wave * (16 * 64 * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int qd = (i * NB + j) * 4 + r4;
              typedef float f32x4 __attribute__((ext_vector_type(4)));
              const f32x4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
              float *dst = ws + (size_t)(qd * 64 + lane) * 4;
              asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gu32 *)(a.sk_flag + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)kQuads;
        ESC_DPROF(7);
        continue;     // (the run's last tile, walked first: the others are whole, or finished below)
      }
      if (k_lo > 0) {
        // this workgroup finishes a tile others began: add their partial sums, nearest run first (a fixed
        // order: results do not depend on timing).  The runs covering the tile's earlier units are those of
        // workgroups blockIdx.x - 1, - 2, ... down to the one holding the tile's first unit.
        const long tile_lo = tile * (long)nk;
        for (long w = (long)blockIdx.x - 1; w >= 0 && (w + 1) * sk_q > tile_lo; --w) {
          if (wave == 0) {
            // ONE wave polls ONE word, relaxed, bounded (a stuck launch must end: the give-up word makes the host fail).
            // Forward progress: the producer is a workgroup with a LOWER blockIdx.x; workgroups are dispatched in
            // index order and the grid never exceeds the resident slots (2 x CUs), so every producer is resident
            // or finished when its consumer polls.  If a runtime ever dispatched out of order on a partly occupied
            // device, the bounded spin ends the launch and the sticky word below turns it into an error.
            unsigned spins = 0;
            while (__hip_atomic_load((gu32 *)(a.sk_flag + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1u << 24)) {
                if (lane == 0) {
                  __hip_atomic_store((gu32 *)(a.sk_flag + gridDim.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store((gu32 *)a.sk_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
          const float *ws = a.sk_ws + (size_t)w * kWsPerWg + (size_t)wave * (16 * 64 * 4);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const int qd = (i * NB + j) * 4 + r4;
                const float4 v = *reinterpret_cast<const float4 *>(ws + (size_t)(qd * 64 + lane) * 4);
                acc[i][j][4 * r4] += v.x; acc[i][j][4 * r4 + 1] += v.y; acc[i][j][4 * r4 + 2] += v.z; acc[i][j][4 * r4 + 3] += v.w;
              }
        }
      }
    }

    ESC_DPROF(7);
    ESC_DPROF_EPI(0);
    // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
    // The bias of the lane's 32 rows is fetched in one go (a load per output, each followed by the
    // compiler's s_waitcnt vmcnt(0), would also wait for the previous STORE every time).
    float bv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        bv[i][reg] = (a.bias && m < a.Mg) ? a.bias[cg * a.Mg + m] : 0.f;
      }
    // ... and pinned down here: loads and stores share one counter and complete out of order with
    // respect to each other, so a load still pending when the stores start makes the compiler put
    // s_waitcnt vmcnt(0) -- wait for the previous store too -- in front of every one of them
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) asm volatile("" : "+v"(bv[i][reg]));
    if (a.vec_out) {
      // Output rows of whole pixel quads (OH*OW % 4 == 0): transpose through LDS and store 16 bytes
      // per lane -- a lane of the MFMA's C layout holds ONE pixel of 16 channels, i.e. 4-byte stores
      // of 256 bytes per instruction, and a layer with few input channels (K = 64: two k-steps per
      // tile) then spends more time issuing stores than multiplying.  Staging area: the buffer the
      // tile's last k-step was read from (the other one is being filled for the next tile); a barrier
      // says everybody is done reading it, the barrier at the next k-step's top that everybody is
      // done with the staging.  A wave transposes its own 32 x WN half-tiles: no cross-wave traffic.
      __syncthreads();
#ifdef ESCOIN_ABLATIONS
      unsigned long long dps_ = 0;
      if (a.prof) { dps_ = __builtin_readcyclecounter(); dpt_[8] += dps_ - dpl_; }      // (since the last stamp: bias, pin, barrier)
#endif
      float *stage = &sAB[buf ^ 1][wave * 32 * WN];
      constexpr int kQPR = WN / 4;                     // quads per staged row
      constexpr int kRowsPerIt = 64 / kQPR;            // rows one 64-lane read covers
      const int c4 = lane % kQPR, r0 = lane / kQPR;
      const int pq = p0 + wn * WN + 4 * c4;
      const int nq = pq / ohw, rq = pq - nq * ohw;
      float *oq = a.out + ((size_t)nq * a.M + (size_t)cg * a.Mg) * ohw + rq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            float v = acc[i][j][reg] + bv[i][reg];
            if (a.relu) v = fmaxf(v, 0.f);
            stage[row * WN + 32 * j + li] = v;
          }
        }
        // (the wave reads back what it wrote itself: LDS operations of a wave complete in order; the
        // compiler must keep the order too)
        asm volatile("" ::: "memory");
        // all of the half-tile's quads first, THEN the stores: read and stored one at a time (a predicated block each: read, wait
        // for LDS, store) the LDS latency stood 16 times in every tile's epilogue -- 25 % of a K = 64 layer's wave life
        // (profiles/r06_dense_stamps.md section 6)
        float4 vq[32 / kRowsPerIt];
#pragma unroll
        for (int it = 0; it < 32 / kRowsPerIt; ++it)
          vq[it] = *reinterpret_cast<const float4 *>(&stage[(it * kRowsPerIt + r0) * WN + 4 * c4]);
        // (pinned: left alone, the compiler sinks every read into the predicated block of its store again)
#pragma unroll
        for (int it = 0; it < 32 / kRowsPerIt; it += 4)
          asm volatile("" : "+v"(vq[it].x), "+v"(vq[it].y), "+v"(vq[it].z), "+v"(vq[it].w), "+v"(vq[it + 1].x), "+v"(vq[it + 1].y),
                            "+v"(vq[it + 1].z), "+v"(vq[it + 1].w), "+v"(vq[it + 2].x), "+v"(vq[it + 2].y), "+v"(vq[it + 2].z), "+v"(vq[it + 2].w),
                            "+v"(vq[it + 3].x), "+v"(vq[it + 3].y), "+v"(vq[it + 3].z), "+v"(vq[it + 3].w));
#pragma unroll
        for (int it = 0; it < 32 / kRowsPerIt; ++it) {
          const int m = m0 + wm * 64 + 32 * i + it * kRowsPerIt + r0;
          if (m < a.Mg && pq < a.P && !ESC_DENSE_ABL(a, 8)) *reinterpret_cast<float4 *>(oq + (size_t)m * ohw) = vq[it];      // (ESCOIN_DENSE_ABL bit 3: no stores)
        }
        asm volatile("" ::: "memory");
      }
#ifdef ESCOIN_ABLATIONS
      if (a.prof) dpt_[9] += __builtin_readcyclecounter() - dps_;
#endif
      ESC_DPROF(6);
      ESC_DPROF_EPI(1);
      continue;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int pj = p0 + wn * WN + 32 * j + li;
      if (pj >= a.P) continue;
      const int nn = pj / ohw;
      const int rr = pj - nn * ohw;
      float *obase = a.out + ((size_t)nn * a.M + (size_t)cg * a.Mg) * ohw + rr;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = m0 + wm * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          if (m >= a.Mg) continue;
          float v = acc[i][j][reg] + bv[i][reg];
          if (a.relu) v = fmaxf(v, 0.f);
          if (!ESC_DENSE_ABL(a, 8)) obase[(size_t)m * ohw] = v;
        }
      }
    }
    ESC_DPROF(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ESC_DPROF(6);
  ESC_DPROF_DUMP
}

// Zeroes the stream-K flag words before a launch (see launch_dense for why this is not a memset).
__global__ void __launch_bounds__(256) escoin_sk_clear_kernel(unsigned *flag, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = 0u;
}

const char *dense_kernel_name() { return "escoin_dense_mfma_kernel"; }

// Layout of the dense weight matrix on the device: rows of dense_lda(K) floats (whole k-steps,
// zero padded), kDenseSpareRows zero rows after the last one.
int dense_lda(int K) { return (K + kBK - 1) / kBK * kBK; }

static int dense_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    hipDeviceProp_t prop;
    c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}
int dense_spare_rows() { return 128; }

// Layers whose launches may split K across workgroups (launch_dense decides per batch): pointwise, stride 1, at least
// eight k-steps -- or whatever ESCOIN_DENSE_STREAMK=1 forces in the experiments flavour.
static bool dense_streamk_eligible(const Geometry &g) {
  static const int sk_env = (int)ESC_KNOB("DENSE_STREAMK", -1);
  if (sk_env >= 0) return sk_env != 0;
  const long nk = (g.kdim + kBK - 1) / kBK;
  return g.d.KH == 1 && g.d.KW == 1 && g.d.stride_h == 1 && g.d.stride_w == 1 && nk >= 8;
}

// The im2col decode of every k (one table per layer, built in WeightAlign): element offset of tap
// (ic, kr, kc) relative to the top-left input element of an output pixel's window, and (dy, dx)
// for the border test.  Entries past K (the last k-step) can never pass the test.
int dense_build_ktab(escoin_plan *p, hipStream_t stream) {
  const Geometry &g = p->g;
  const int K = g.kdim, Kpad = (K + kBK - 1) / kBK * kBK;
  std::vector<int> tab((size_t)Kpad * 2);
  for (int k = 0; k < Kpad; ++k) {
    if (k < K) {
      const int kc = k % g.d.KW, kr = (k / g.d.KW) % g.d.KH, ic = k / (g.d.KW * g.d.KH);
      const int dy = kr * g.d.dil_h, dx = kc * g.d.dil_w;
      tab[2 * k] = (ic * g.d.H + dy) * g.d.W + dx;
      tab[2 * k + 1] = (dy & 0xFFFF) | (dx << 16);
    } else {
      tab[2 * k] = 0;
      tab[2 * k + 1] = 0x7FFF | (0x7FFF << 16);
    }
  }
  if (p->d_ktab) (void)hipFree(p->d_ktab);
  p->d_ktab = nullptr;
  ESCOIN_HIP_TRY(hipMalloc(&p->d_ktab, sizeof(int) * tab.size()));
  p->device_bytes += sizeof(int) * tab.size();
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_ktab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipStreamSynchronize(stream));
  // Stream-K workspace, here and not at the first launch: escoin_forward must neither allocate nor synchronise (a host
  // may capture its very first forward into a HIP graph).  A stream-K launch always runs
  // 2 x CUs workgroups whatever the batch, so the size is known now: the flag words, then 64 KB of partial
  // accumulators per workgroup (32 MB on an MI355X), plus the pinned host word a give-up is reported through.  Only
  // for layers launch_dense can ever split (dense_streamk_eligible).
  if (dense_streamk_eligible(g)) {
    const long n_wg = 2l * dense_device_cus();
    const size_t ws_bytes = (size_t)n_wg * 4 * 16 * 64 * 4 * sizeof(float);
    const size_t flag_bytes = ((size_t)(n_wg + 1) * 4 + 15) / 16 * 16;
    if (!p->h_sk_fail) ESCOIN_HIP_TRY(hipHostMalloc((void **)&p->h_sk_fail, 64, hipHostMallocMapped));
    *p->h_sk_fail = 0u;
    ESCOIN_HIP_TRY(hipHostGetDevicePointer((void **)&p->d_sk_fail, p->h_sk_fail, 0));
    if (p->d_sk_ws) (void)hipFree(p->d_sk_ws);
    p->d_sk_ws = nullptr;
    p->sk_ws_bytes = 0;
    ESCOIN_HIP_TRY(hipMalloc(&p->d_sk_ws, ws_bytes + flag_bytes));
    p->sk_ws_bytes = ws_bytes + flag_bytes;
    p->device_bytes += ws_bytes + flag_bytes;
    ESCOIN_HIP_TRY(hipMemset(p->d_sk_ws, 0, flag_bytes));
    p->sk_flag_words = (int)n_wg + 1;
  }
  return ESCOIN_OK;
}

int launch_dense(const escoin_plan *p, const float *bottom, const float *bias, float *top,
                 int n_images, hipStream_t stream) {
  const Geometry &g = p->g;
  if (!p->d_ktab || !p->d_dense_w) return fail(ESCOIN_ESTATE, "dense kernel: plan has no dense weights");
  DenseArgs a;
  a.in = bottom; a.w = p->d_dense_w; a.bias = bias; a.out = top;
  a.ktab = reinterpret_cast<const int2 *>(p->d_ktab);
  a.n_images = n_images; a.C = g.d.C; a.H = g.d.H; a.W = g.d.W; a.M = g.d.M; a.OH = g.OH; a.OW = g.OW;
  a.pad_h = g.d.pad_h; a.pad_w = g.d.pad_w; a.stride_h = g.d.stride_h; a.stride_w = g.d.stride_w;
  a.Cg = g.Cg; a.Mg = g.Mg; a.K = g.kdim; a.lda = dense_lda(g.kdim); a.relu = g.d.fuse_relu;
  const long P = (long)n_images * g.OH * g.OW;
  if (P >= (1l << 31)) return fail(ESCOIN_EINVAL, "dense kernel: N*OH*OW does not fit 31 bits");
  if (g.d.dil_h * (g.d.KH - 1) > 0x7FFE || g.d.dil_w * (g.d.KW - 1) > 0x7FFE)
    return fail(ESCOIN_EINVAL, "dense kernel: dilated kernel extent does not fit 15 bits");
  a.P = (int)P;
  // the operands are addressed through buffer descriptors with 32-bit byte offsets
  const size_t in_bytes = (size_t)n_images * g.d.C * g.d.H * g.d.W * 4;
  const size_t w_bytes = ((size_t)g.d.M + dense_spare_rows()) * a.lda * 4;
  if (in_bytes >= 0xFFFFFFF0ull || w_bytes >= 0xFFFFFFF0ull)
    return fail(ESCOIN_EINVAL, "dense kernel: bottom blob or weight matrix exceeds the 4 GiB descriptor range");
  a.in_bytes = (unsigned)in_bytes;
  a.w_bytes = (unsigned)w_bytes;
  // pointwise (is_1x1_, base_conv_layer.cpp:374-379) with whole quads of pixels per image: the
  // column matrix IS the bottom blob and is staged 16 bytes at a time
  const bool pointwise = g.d.KH == 1 && g.d.KW == 1 && g.d.stride_h == 1 && g.d.stride_w == 1 &&
                         g.d.pad_h == 0 && g.d.pad_w == 0;
  const bool vec_b = pointwise && (g.d.H * g.d.W) % 4 == 0 && (reinterpret_cast<uintptr_t>(bottom) & 15) == 0 && P >= 4;
  // Strided pointwise layers (1x1, stride > 1, no padding) with the B tile staged through registers (kernel, BMODE 2: with
  // stride 2, an even output width and 16-byte aligned rows a lane's two outputs are one aligned quad's elements 0 and
  // 2) -- BUILT, MEASURED, NOT SHIPPED: same-call A/B on the ResNet-50 chain's six stride-2 layers at batch 256
  // (profiles/r05_dense.md) 624-637 -> 642-678 us and 187-209 -> 188-228 us, i.e. 0-9 % SLOWER than the 4-byte LDS-DMA
  // gather (parity green: tests/test_gpu_parity.py::test_dense_strided_pointwise_through_registers runs it in the
  // experiments flavour).  Exists in the experiments flavour only, behind ESCOIN_DENSE_S2=1; the product gathers.
#ifdef ESCOIN_EXPERIMENTS
  static const bool s2_on = (ESC_KNOB("DENSE_S2", 0) != 0);
#else
  constexpr bool s2_on = false;
#endif
  const bool strided1 = s2_on && g.d.KH == 1 && g.d.KW == 1 && g.d.pad_h == 0 && g.d.pad_w == 0 && !pointwise && P >= 2;
  a.s2_pair = strided1 && g.d.stride_w == 2 && g.OW % 2 == 0 && (g.d.H * g.d.W) % 4 == 0 && (g.d.stride_h * g.d.W) % 4 == 0 &&
              (reinterpret_cast<uintptr_t>(bottom) & 15) == 0 ? 1 : 0;
  {
    static const bool vo = (ESC_KNOB("DENSE_VEC_OUT", 1) != 0);
    a.vec_out = vo && (g.OH * g.OW) % 4 == 0 && (reinterpret_cast<uintptr_t>(top) & 15) == 0;
  }
  a.group_mask = p->use_dense ? ~0ull : p->dense_mask;
  a.abl = ESC_ABL_KNOB("DENSE_ABL");
  {
    static const int slots_knob = (int)ESC_KNOB("DENSE_SLOTS", 1);
    a.fetch_slots = slots_knob != 0 ? 1 : 0;
  }
  a.prof = nullptr;
#ifdef ESCOIN_ABLATIONS
  static unsigned long long *prof_buf = nullptr;
  if (ESC_ABL_KNOB("PROF")) {
    if (!prof_buf) ESCOIN_HIP_TRY(hipMalloc(&prof_buf, sizeof(unsigned long long) * (68 * 4096 + 48 * 4096)));
    ESCOIN_HIP_TRY(hipMemsetAsync(prof_buf, 0, sizeof(unsigned long long) * (68 * 4096 + 48 * 4096), stream));
    a.prof = prof_buf;
  }
#endif
  const int bm = g.Mg <= 64 ? 64 : 128;
  a.n_ptiles = (int)((P + kBN - 1) / kBN);
  a.n_mtiles = (g.Mg + bm - 1) / bm;
  a.n_groups = p->use_dense ? g.d.group : p->n_dense_groups;
  const long tiles = (long)a.n_ptiles * a.n_mtiles * a.n_groups;
  // persistent: two 4-wave workgroups per CU walk the tiles
  const long slots = 2l * dense_device_cus();
  // Stream-K where whole tiles leave slots idle: 784 tiles on 512 slots are two rounds at 77 % occupancy, 392 tiles
  // one round at 77 % (the ResNet-50 chain's 14 x 14 and 7 x 7 1x1 layers; hipBLASLt splits K on exactly these
  // shapes and led by 13-24 %, profiles/r03_dense_vs_libraries.md).  Every workgroup then takes an equal run of
  // (tile, k-step) units and the tiles cut by run boundaries are fixed up in the launch (kernel, STREAMK).
  // Taken when the tile schedule would waste more than 15 % of the slots and a tile has at least eight k-steps
  // (same-call A/B on the chain, profiles/r04_dense_streamk.md: 784 tiles -13 %, 392 tiles -16 %, 1568 tiles -4..-5 %;
  // at 87.5 % occupancy -- 3136 tiles -- the fix-up costs more than the idle slots: +4..6 %).
  // ESCOIN_DENSE_STREAMK = 0 / 1 forces it off / on.
  static const int sk_env = (int)ESC_KNOB("DENSE_STREAMK", -1);
  const long nk = (g.kdim + kBK - 1) / kBK;
  const long rounds = (tiles + slots - 1) / slots;
  const double occupancy = (double)tiles / (double)(rounds * slots);
  // (1x1 layers of stride 1 only: that is what was measured -- the stride-2 layer res5a_branch1 went from 581 to 629 us
  //  with it, 3x3 layers have not been tried)
  const bool unit_1x1 = g.d.KH == 1 && g.d.KW == 1 && g.d.stride_h == 1 && g.d.stride_w == 1;
  bool streamk = sk_env >= 0 ? sk_env != 0 : (unit_1x1 && occupancy < 0.85 && nk >= 8 && tiles * nk >= 4 * slots);
  if (tiles * nk < slots || strided1) streamk = false;
  // (the workspace and the pinned give-up word were allocated by dense_build_ktab at WeightAlign; a launch never
  //  allocates -- a plan without them runs whole tiles)
  if (streamk && (!p->d_sk_ws || !p->h_sk_fail || !p->d_sk_fail ||
                  p->sk_ws_bytes < (size_t)slots * 4 * 16 * 64 * 4 * sizeof(float) + ((size_t)(slots + 1) * 4 + 15) / 16 * 16))
    streamk = false;
  const long n_wg = streamk ? slots : std::min<long>(tiles, slots);
  a.sk_ws = nullptr;
  a.sk_flag = nullptr;
  a.sk_fail = nullptr;
  if (streamk) {
    a.sk_fail = p->d_sk_fail;
    const size_t flag_bytes = ((size_t)(n_wg + 1) * 4 + 15) / 16 * 16;
    a.sk_flag = reinterpret_cast<unsigned *>(p->d_sk_ws);
    a.sk_ws = reinterpret_cast<float *>(reinterpret_cast<char *>(p->d_sk_ws) + flag_bytes);
    // re-arm the flags with a KERNEL, not hipMemsetAsync: captured into a HIP graph, a memset node of this size
    // (2064 bytes) left garbage in its last 16 bytes -- the give-up word -- from the second replay on (ROCm 7.2, MI355X;
    // tools/dbg/sk_graph.py: eager launches and the first replay read 0, later replays 0xDB3F....), while kernel nodes
    // replay exactly
    hipLaunchKernelGGL(escoin_sk_clear_kernel, dim3(((unsigned)(flag_bytes / 4) + 255) / 256), dim3(256), 0, stream, a.sk_flag,
                       (int)(flag_bytes / 4));
  }
  p->sk_used = streamk;
  dim3 grid((unsigned)n_wg, 1, 1);
#define ESC_DENSE_LAUNCH(WR, MODE)                                                                                 \
  do {                                                                                                             \
    if (streamk) hipLaunchKernelGGL((escoin_dense_mfma_kernel<WR, MODE, true>), grid, dim3(256), 0, stream, a);     \
    else hipLaunchKernelGGL((escoin_dense_mfma_kernel<WR, MODE, false>), grid, dim3(256), 0, stream, a);           \
  } while (0)
  if (bm == 64) {
    if (vec_b) ESC_DENSE_LAUNCH(1, 1);
#ifdef ESCOIN_EXPERIMENTS
    else if (strided1) hipLaunchKernelGGL((escoin_dense_mfma_kernel<1, 2, false>), grid, dim3(256), 0, stream, a);
#endif
    else ESC_DENSE_LAUNCH(1, 0);
  } else {
    if (vec_b) ESC_DENSE_LAUNCH(2, 1);
#ifdef ESCOIN_EXPERIMENTS
    else if (strided1) hipLaunchKernelGGL((escoin_dense_mfma_kernel<2, 2, false>), grid, dim3(256), 0, stream, a);
#endif
    else ESC_DENSE_LAUNCH(2, 0);
  }
#undef ESC_DENSE_LAUNCH
  ESCOIN_HIP_TRY(hipGetLastError());
#ifdef ESCOIN_ABLATIONS
  if (a.prof && n_wg <= 4096) {
    // (every launch while ESCOIN_PROF=1: synchronises -- a profiling run, not a timing run)
    ESCOIN_HIP_TRY(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h((size_t)68 * 4096 + 48 * 4096);
    ESCOIN_HIP_TRY(hipMemcpy(h.data(), a.prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    static const char *names[10] = {"start-up", "operand wait", "barrier", "fetch issue", "reads + MFMAs", "tile setup", "epilogue", "stream-K",
                                    "epilogue: bias + first barrier", "epilogue: the rest"};
    double cat[10] = {0}, cyc = 0, rt = 0, first = 1e30, last = 0, mx[10] = {0};
    for (long w = 0; w < n_wg; ++w) {
      for (int i = 0; i < 10; ++i) {
        double sum = 0;
        for (int wv = 0; wv < 4; ++wv) sum += (double)h[((size_t)w * 4 + wv) * 16 + i];
        cat[i] += sum / 4;
        mx[i] = std::max(mx[i], sum / 4);
      }
      cyc += (double)h[(size_t)64 * 4096 + 4 * w];
      rt += (double)h[(size_t)64 * 4096 + 4 * w + 1];
      first = std::min(first, (double)h[(size_t)64 * 4096 + 4 * w + 2]);
      last = std::max(last, (double)h[(size_t)64 * 4096 + 4 * w + 2] + (double)h[(size_t)64 * 4096 + 4 * w + 1]);
    }
    const double ghz = cyc / (rt * 10.0);
    double tot = 0;
    for (int i = 0; i < 8; ++i) tot += cat[i] / n_wg;      // (8 and 9 split category 6: they are reported, not added)
    fprintf(stderr, "[dprof] %ld workgroups (%s, BM %d, %s), %ld tiles x %ld k-steps; workgroup life %.0f cycles = %.2f us @ %.3f GHz; first start to last end %.2f us\n",
            n_wg, streamk ? "stream-K" : "tiles", bm, vec_b ? "pointwise 16-byte B" : "gathered B", tiles, nk, cyc / n_wg, rt / n_wg * 0.01, ghz, (last - first) * 0.01);
    fprintf(stderr, "[dprof] mean wave cycles per workgroup by phase:");
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %s=%.0f (%.2f us, %.0f %%)", names[i], cat[i] / n_wg, cat[i] / n_wg / ghz * 1e-3, 100.0 * cat[i] / n_wg / std::max(1.0, tot));
    fprintf(stderr, " | accounted %.0f %% of the workgroup life\n", 100.0 * tot / std::max(1.0, cyc / n_wg));
    // Are the two workgroups of a CU (i and i + n / 2: tools/probes/probe_hwid.hip) in step?  Over their tiles 4 .. 23: the time both spend in their
    // epilogues at once / the time either's epilogue lasts.  1 = lockstep, ~ epilogue share of the tile period = independent phases, 0 = alternating.
    if (n_wg % 2 == 0 && n_wg >= 2) {
      const unsigned long long *ep = h.data() + (size_t)68 * 4096;
      double both = 0, one = 0, period = 0, elen = 0;
      long np = 0;
      for (long w = 0; w < n_wg / 2; ++w) {
        const unsigned long long *A = ep + (size_t)w * 48, *B = ep + (size_t)(w + n_wg / 2) * 48;
        if (!A[2 * 23 + 1] || !B[2 * 23 + 1]) continue;
        for (int t = 4; t < 23; ++t) {
          const double a0 = (double)A[2 * t], a1 = (double)A[2 * t + 1];
          one += a1 - a0;
          for (int u = 0; u < 24; ++u) {
            const double b0 = (double)B[2 * u], b1 = (double)B[2 * u + 1];
            both += std::max(0.0, std::min(a1, b1) - std::max(a0, b0));
          }
          period += (double)A[2 * (t + 1)] - a0;
          elen += a1 - a0;
          ++np;
        }
      }
      if (np > 0)
        fprintf(stderr, "[dprof] CU partners (workgroups i, i + %ld), tiles 4-22: tile period %.2f us, epilogue %.2f us = %.0f %% of it; partner in its epilogue during %.0f %% of a workgroup's epilogue time (independent phases would give ~%.0f %%, lockstep 100 %%)\n",
                n_wg / 2, period / np * 0.01, elen / np * 0.01, 100.0 * elen / period, 100.0 * both / one, 100.0 * elen / period);
    }
  }
#endif
  return ESCOIN_OK;
}

}  // namespace escoin
