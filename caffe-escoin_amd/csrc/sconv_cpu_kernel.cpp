// sconv_cpu_kernel.cpp -- the host kernel of the CPU mode (see sconv_cpu.h), compiled twice by the Makefile:
//   -DESC_CPU_ISA=2   -mavx2 -mfma                 -> run_group_avx2<float|double>
//   -DESC_CPU_ISA=512 -mavx512f -mavx512vl ...     -> run_group_avx512<float|double>
// sconv_cpu.cpp picks one at run time from what the host CPU reports.
//
// What the reference does here: caffe_cpu_sconv (math_functions.cpp:162-174) walks pixel -> output channel -> nonzero
// with one scalar multiply-add per nonzero and a gathered load; its ICC-only fast path sconv_unit_stride
// (sconv.hpp:57-589) keeps a register tile of output pixels per output channel and streams the channel's nonzeros
// over it.  This kernel is built on that second idea, redone for the shared-halo layout as it is:
//
//   * stride 1 (any dilation): on the padded layout an output pixel q = oh * PW + ow reads in[off_j + q] for every
//     nonzero j, so the "virtual" pixels 0 .. (OH-1) * PW + OW - 1 are ONE contiguous axis: a tile is NV vector
//     registers of consecutive virtual pixels, rows and images of any width fill whole vectors (7x7 and 13x13 layers
//     do not waste lanes per row), and the pad_w virtual pixels per row that are not outputs are dropped at the store;
//   * tile outer, output channel inner: the tile's input window (all channels x NV vectors) stays in L1 while every
//     channel's nonzeros stream over it once;
//   * per nonzero: one broadcast of the value, one scalar offset, NV fused multiply-adds with a memory operand.
//   * other strides: one output row at a time, scalar lanes (the reference's fast kernels do not cover them either,
//     math_functions.cpp:201-462).
//
// Summation order per output is the CSR order from zero, fused multiply-add, bias added afterwards once, then ReLU:
// exactly the reference's, so the results are bit-identical to caffe_cpu_sconv's whatever the tile shape.
#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "sconv_cpu.h"

#ifndef ESC_CPU_ISA
#error "compile with -DESC_CPU_ISA=2 or -DESC_CPU_ISA=512"
#endif

namespace escoin {
namespace cpu {
namespace {

template <typename T> struct Vec;

#if ESC_CPU_ISA == 512
template <> struct Vec<float> {
  typedef __m512 V;
  enum { L = 16 };
  static V zero() { return _mm512_setzero_ps(); }
  static V bcast(float x) { return _mm512_set1_ps(x); }
  static V load(const float *p) { return _mm512_loadu_ps(p); }
  static V load_n(const float *p, int n) { return _mm512_maskz_loadu_ps((__mmask16)((1u << n) - 1u), p); }
  static void store(float *p, V v) { _mm512_storeu_ps(p, v); }
  static V fma(V a, V b, V c) { return _mm512_fmadd_ps(a, b, c); }
  static V add(V a, V b) { return _mm512_add_ps(a, b); }
  static V relu(V a) { return _mm512_max_ps(a, _mm512_setzero_ps()); }
};
template <> struct Vec<double> {
  typedef __m512d V;
  enum { L = 8 };
  static V zero() { return _mm512_setzero_pd(); }
  static V bcast(double x) { return _mm512_set1_pd(x); }
  static V load(const double *p) { return _mm512_loadu_pd(p); }
  static V load_n(const double *p, int n) { return _mm512_maskz_loadu_pd((__mmask8)((1u << n) - 1u), p); }
  static void store(double *p, V v) { _mm512_storeu_pd(p, v); }
  static V fma(V a, V b, V c) { return _mm512_fmadd_pd(a, b, c); }
  static V add(V a, V b) { return _mm512_add_pd(a, b); }
  static V relu(V a) { return _mm512_max_pd(a, _mm512_setzero_pd()); }
};
constexpr int kMaxVecs = 14;   // 32 vector registers: 14 accumulators + the broadcast leave room to spare
#else
static inline __m256i mask8(int n) {
  const __m256i idx = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
  return _mm256_cmpgt_epi32(_mm256_set1_epi32(n), idx);
}
static inline __m256i mask4(int n) {
  const __m256i idx = _mm256_setr_epi64x(0, 1, 2, 3);
  return _mm256_cmpgt_epi64(_mm256_set1_epi64x(n), idx);
}
template <> struct Vec<float> {
  typedef __m256 V;
  enum { L = 8 };
  static V zero() { return _mm256_setzero_ps(); }
  static V bcast(float x) { return _mm256_set1_ps(x); }
  static V load(const float *p) { return _mm256_loadu_ps(p); }
  static V load_n(const float *p, int n) { return _mm256_maskload_ps(p, mask8(n)); }
  static void store(float *p, V v) { _mm256_storeu_ps(p, v); }
  static V fma(V a, V b, V c) { return _mm256_fmadd_ps(a, b, c); }
  static V add(V a, V b) { return _mm256_add_ps(a, b); }
  static V relu(V a) { return _mm256_max_ps(a, _mm256_setzero_ps()); }
};
template <> struct Vec<double> {
  typedef __m256d V;
  enum { L = 4 };
  static V zero() { return _mm256_setzero_pd(); }
  static V bcast(double x) { return _mm256_set1_pd(x); }
  static V load(const double *p) { return _mm256_loadu_pd(p); }
  static V load_n(const double *p, int n) { return _mm256_maskload_pd(p, mask4(n)); }
  static void store(double *p, V v) { _mm256_storeu_pd(p, v); }
  static V fma(V a, V b, V c) { return _mm256_fmadd_pd(a, b, c); }
  static V add(V a, V b) { return _mm256_add_pd(a, b); }
  static V relu(V a) { return _mm256_max_pd(a, _mm256_setzero_pd()); }
};
constexpr int kMaxVecs = 12;   // 16 vector registers: 12 accumulators, the broadcast, spare
#endif

struct Seg { int dst, src, len; };   // a run of real outputs inside a tile of virtual pixels

// One tile of NV vectors starting at virtual pixel q0, every output channel of the job.
//   masked_n: valid lanes of the LAST vector when its load must not run past them (1 .. L-1), 0 = load it whole.
template <typename T, int NV>
static void tile_unit_stride(const GroupJob<T> &J, int q0, int masked_n, const Seg *segs, int nseg, bool direct,
                             int n_valid) {
  typedef Vec<T> X;
  typedef typename X::V V;
  constexpr int L = X::L;
  const T *base = J.in + q0;
  const size_t plane = (size_t)J.OH * J.OW;
  for (int m = J.m_begin; m < J.m_end; ++m) {
    V acc[NV];
    for (int t = 0; t < NV; ++t) acc[t] = X::zero();
    const int jb = J.rowptr[m], je = J.rowptr[m + 1];
    if (masked_n == 0) {
      for (int j = jb; j < je; ++j) {
        const V v = X::bcast(J.val[j]);
        const T *p = base + J.off[j];
#pragma unroll
        for (int t = 0; t < NV; ++t) acc[t] = X::fma(v, X::load(p + t * L), acc[t]);
      }
    } else {
      for (int j = jb; j < je; ++j) {
        const V v = X::bcast(J.val[j]);
        const T *p = base + J.off[j];
#pragma unroll
        for (int t = 0; t < NV - 1; ++t) acc[t] = X::fma(v, X::load(p + t * L), acc[t]);
        acc[NV - 1] = X::fma(v, X::load_n(p + (NV - 1) * L, masked_n), acc[NV - 1]);
      }
    }
    if (J.bias) {
      const V b = X::bcast(J.bias[m]);
      for (int t = 0; t < NV; ++t) acc[t] = X::add(acc[t], b);
    }
    if (J.relu)
      for (int t = 0; t < NV; ++t) acc[t] = X::relu(acc[t]);
    T *outp = J.out + (size_t)m * plane;
    if (direct && n_valid == NV * L) {
      for (int t = 0; t < NV; ++t) X::store(outp + q0 + t * L, acc[t]);
    } else {
      for (int t = 0; t < NV; ++t) X::store(J.scratch + t * L, acc[t]);
      for (int s = 0; s < nseg; ++s) memcpy(outp + segs[s].dst, J.scratch + segs[s].src, sizeof(T) * (size_t)segs[s].len);
    }
  }
}

// The same tile with the input channels going by in blocks (GroupJob::blk_ptr): block outer, output channel inner.
template <typename T, int NV>
static void tile_unit_stride_blocked(const GroupJob<T> &J, int q0, int masked_n, const Seg *segs, int nseg, bool direct,
                                     int n_valid) {
  typedef Vec<T> X;
  typedef typename X::V V;
  constexpr int L = X::L;
  const T *base = J.in + q0;
  const size_t plane = (size_t)J.OH * J.OW;
  const int nb = J.n_blk;
  for (int b = 0; b < nb; ++b) {
    const bool first = b == 0, last = b == nb - 1;
    for (int m = J.m_begin; m < J.m_end; ++m) {
      const int jb = J.blk_ptr[(size_t)m * (nb + 1) + b], je = J.blk_ptr[(size_t)m * (nb + 1) + b + 1];
      T *park = J.partial + (size_t)(m - J.m_begin) * kPartialElemsPerRow;
      if (jb == je && !first && !last) continue;             // nothing of this row in this block: its sums stay parked
      V acc[NV];
      if (first) {
        for (int t = 0; t < NV; ++t) acc[t] = X::zero();
      } else {
        for (int t = 0; t < NV; ++t) acc[t] = X::load(park + t * L);
      }
      if (masked_n == 0) {
        for (int j = jb; j < je; ++j) {
          const V v = X::bcast(J.val[j]);
          const T *p = base + J.off[j];
#pragma unroll
          for (int t = 0; t < NV; ++t) acc[t] = X::fma(v, X::load(p + t * L), acc[t]);
        }
      } else {
        for (int j = jb; j < je; ++j) {
          const V v = X::bcast(J.val[j]);
          const T *p = base + J.off[j];
#pragma unroll
          for (int t = 0; t < NV - 1; ++t) acc[t] = X::fma(v, X::load(p + t * L), acc[t]);
          acc[NV - 1] = X::fma(v, X::load_n(p + (NV - 1) * L, masked_n), acc[NV - 1]);
        }
      }
      if (!last) {
        for (int t = 0; t < NV; ++t) X::store(park + t * L, acc[t]);
        continue;
      }
      if (J.bias) {
        const V bv = X::bcast(J.bias[m]);
        for (int t = 0; t < NV; ++t) acc[t] = X::add(acc[t], bv);
      }
      if (J.relu)
        for (int t = 0; t < NV; ++t) acc[t] = X::relu(acc[t]);
      T *outp = J.out + (size_t)m * plane;
      if (direct && n_valid == NV * L) {
        for (int t = 0; t < NV; ++t) X::store(outp + q0 + t * L, acc[t]);
      } else {
        for (int t = 0; t < NV; ++t) X::store(J.scratch + t * L, acc[t]);
        for (int s = 0; s < nseg; ++s) memcpy(outp + segs[s].dst, J.scratch + segs[s].src, sizeof(T) * (size_t)segs[s].len);
      }
    }
  }
}

// NI small images at once (each one tile of NV vectors starting at virtual pixel 0), with or without channel blocks:
// one broadcast per nonzero feeds NI x NV multiply-adds.  Per output the arithmetic is what the single-image tile does.
template <typename T, int NV, int NI>
static void tile_multi(const GroupJob<T> &J, int masked_n, const Seg *segs, int nseg, bool direct, int n_valid) {
  typedef Vec<T> X;
  typedef typename X::V V;
  constexpr int L = X::L;
  const size_t plane = (size_t)J.OH * J.OW;
  const bool blocked = J.blk_ptr && J.n_blk > 1;
  const int nb = blocked ? J.n_blk : 1;
  const T *ins[NI];
  for (int i = 0; i < NI; ++i) ins[i] = J.in + (size_t)i * J.in_stride;
  for (int b = 0; b < nb; ++b) {
    const bool first = b == 0, last = b == nb - 1;
    for (int m = J.m_begin; m < J.m_end; ++m) {
      const int jb = blocked ? J.blk_ptr[(size_t)m * (nb + 1) + b] : J.rowptr[m];
      const int je = blocked ? J.blk_ptr[(size_t)m * (nb + 1) + b + 1] : J.rowptr[m + 1];
      T *park = blocked ? J.partial + (size_t)(m - J.m_begin) * kPartialElemsPerRow : nullptr;
      if (jb == je && !first && !last) continue;
      V acc[NI * NV];
      if (first) {
        for (int k = 0; k < NI * NV; ++k) acc[k] = X::zero();
      } else {
        for (int k = 0; k < NI * NV; ++k) acc[k] = X::load(park + k * L);
      }
      if (masked_n == 0) {
        for (int j = jb; j < je; ++j) {
          const V v = X::bcast(J.val[j]);
          const int o = J.off[j];
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const T *p = ins[i] + o;
#pragma unroll
            for (int t = 0; t < NV; ++t) acc[i * NV + t] = X::fma(v, X::load(p + t * L), acc[i * NV + t]);
          }
        }
      } else {
        for (int j = jb; j < je; ++j) {
          const V v = X::bcast(J.val[j]);
          const int o = J.off[j];
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const T *p = ins[i] + o;
#pragma unroll
            for (int t = 0; t < NV - 1; ++t) acc[i * NV + t] = X::fma(v, X::load(p + t * L), acc[i * NV + t]);
            acc[i * NV + NV - 1] = X::fma(v, X::load_n(p + (NV - 1) * L, masked_n), acc[i * NV + NV - 1]);
          }
        }
      }
      if (!last) {
        for (int k = 0; k < NI * NV; ++k) X::store(park + k * L, acc[k]);
        continue;
      }
      if (J.bias) {
        const V bv = X::bcast(J.bias[m]);
        for (int k = 0; k < NI * NV; ++k) acc[k] = X::add(acc[k], bv);
      }
      if (J.relu)
        for (int k = 0; k < NI * NV; ++k) acc[k] = X::relu(acc[k]);
      for (int i = 0; i < NI; ++i) {
        T *outp = J.out + (size_t)i * J.out_stride + (size_t)m * plane;
        if (direct && n_valid == NV * L) {
          for (int t = 0; t < NV; ++t) X::store(outp + t * L, acc[i * NV + t]);
        } else {
          for (int t = 0; t < NV; ++t) X::store(J.scratch + t * L, acc[i * NV + t]);
          for (int s = 0; s < nseg; ++s) memcpy(outp + segs[s].dst, J.scratch + segs[s].src, sizeof(T) * (size_t)segs[s].len);
        }
      }
    }
  }
}

constexpr int kMaxImages = 3;                 // images per job at most
template <typename T> struct MultiTable {
  typedef void (*Fn)(const GroupJob<T> &, int, const Seg *, int, bool, int);
  Fn fn[kMaxImages + 1][kMaxVecs / 2 + 1];
  MultiTable() {
    for (auto &row : fn) for (auto &f : row) f = nullptr;
#define ESC_MULTI(NV) fn[2][NV] = &tile_multi<T, NV, 2>; if (3 * NV <= kMaxVecs) fn[3][NV] = &tile_multi<T, NV, (3 * NV <= kMaxVecs ? 3 : 2)>;
    ESC_MULTI(1) ESC_MULTI(2) ESC_MULTI(3) ESC_MULTI(4) ESC_MULTI(5) ESC_MULTI(6)
#if ESC_CPU_ISA == 512
    ESC_MULTI(7)
#endif
#undef ESC_MULTI
  }
};

template <typename T>
static int images_per_job(int OH, int OW, int PW, int stride_h, int stride_w) {
  constexpr int L = Vec<T>::L;
  if (stride_h != 1 || stride_w != 1 || OH < 1 || OW < 1) return 1;
  const int Q = (OH - 1) * PW + OW;
  const int nvec = (Q + L - 1) / L;
  if (2 * nvec > kMaxVecs) return 1;
  return std::min(kMaxImages, kMaxVecs / nvec);
}

template <typename T, int NV>
struct TileTable {
  static void fill(void (**tab)(const GroupJob<T> &, int, int, const Seg *, int, bool, int)) {
    tab[NV] = &tile_unit_stride<T, NV>;
    tab[kMaxVecs + 1 + NV] = &tile_unit_stride_blocked<T, NV>;
    TileTable<T, NV - 1>::fill(tab);
  }
};
template <typename T>
struct TileTable<T, 0> {
  static void fill(void (**)(const GroupJob<T> &, int, int, const Seg *, int, bool, int)) {}
};

template <typename T>
static void run_unit_stride(const GroupJob<T> &J) {
  constexpr int L = Vec<T>::L;
  typedef void (*TileFn)(const GroupJob<T> &, int, int, const Seg *, int, bool, int);
  struct Table {
    TileFn fn[2 * (kMaxVecs + 1)];       // [nv] = all channels at once, [kMaxVecs + 1 + nv] = channel blocks
    Table() { fn[0] = fn[kMaxVecs + 1] = nullptr; TileTable<T, kMaxVecs>::fill(fn); }
  };
  static const Table table;   // (a function-local static: initialised once, thread-safe)
  const int Q = (J.OH - 1) * J.PW + J.OW;               // virtual pixels
  const int nvec = (Q + L - 1) / L;
  if (J.n_img > 1 && (J.n_img > kMaxImages || J.n_img * nvec > kMaxVecs)) {      // (more images than fit: one after the other)
    GroupJob<T> one = J;
    one.n_img = 1;
    for (int i = 0; i < J.n_img; ++i) {
      one.in = J.in + (size_t)i * J.in_stride;
      one.out = J.out + (size_t)i * J.out_stride;
      run_unit_stride<T>(one);
    }
    return;
  }
  const int ntiles = (nvec + kMaxVecs - 1) / kMaxVecs;
  const int per_tile = (nvec + ntiles - 1) / ntiles;     // evenly sized tiles (14 x 14: 7 + 7 vectors, not 12 + 2)
  const bool direct = J.PW == J.OW;                      // no dropped columns: virtual pixel == output index
  // the runs of real outputs of one tile (at most rows-in-a-tile + 1 of them)
  Seg segs[kMaxVecs * 16 + 8];
  for (int t0 = 0; t0 < nvec; t0 += per_tile) {
    const int nv = std::min(per_tile, nvec - t0);
    const int q0 = t0 * L, q1 = std::min(Q, q0 + nv * L);
    // The runs of real outputs inside the tile, row by row: output row oh owns the virtual pixels [oh * PW, oh * PW + OW).
    // (With pad_w > (KW - 1) * dil_w / 2 a row has MORE outputs than the padded pitch, OW > PW: its last outputs are the
    //  virtual pixels the next row starts with -- the same addresses, the same sums: the shared-halo layout's own
    //  wrap-around, math_functions.cpp:162-174 -- so rows overlap on the virtual axis and each takes its own copy.)
    int nseg = 0;
    const int oh_lo = std::max(0, (q0 - J.OW + J.PW) / J.PW), oh_hi = std::min(J.OH - 1, (q1 - 1) / J.PW);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int lo = std::max(q0, oh * J.PW), hi = std::min(q1, oh * J.PW + J.OW);
      if (hi <= lo) continue;
      segs[nseg].dst = oh * J.OW + (lo - oh * J.PW);
      segs[nseg].src = lo - q0;
      segs[nseg].len = hi - lo;
      ++nseg;
    }
    const int tail = q1 - (q0 + (nv - 1) * L);           // valid lanes of the last vector, 1 .. L
    const int masked_n = (J.exact_reads && tail < L) ? tail : 0;
    if (J.n_img > 1) {
      static const MultiTable<T> multi;
      multi.fn[J.n_img][nv](J, masked_n, segs, nseg, direct, q1 - q0);     // (one tile per image: q0 = 0)
    } else {
      table.fn[(J.blk_ptr && J.n_blk > 1 ? kMaxVecs + 1 : 0) + nv](J, q0, masked_n, segs, nseg, direct, q1 - q0);
    }
  }
}

// Any stride: one output row at a time, kB scalar accumulators along the row so that a nonzero's value and offset
// are loaded once per kB outputs.
template <typename T>
static void run_any_stride(const GroupJob<T> &J) {
  constexpr int kB = 8;
  const size_t plane = (size_t)J.OH * J.OW;
  for (int m = J.m_begin; m < J.m_end; ++m) {
    const int jb = J.rowptr[m], je = J.rowptr[m + 1];
    const T b = J.bias ? J.bias[m] : T(0);
    T *outp = J.out + (size_t)m * plane;
    for (int oh = 0; oh < J.OH; ++oh) {
      const T *row = J.in + (size_t)oh * J.stride_h * J.PW;
      for (int ow0 = 0; ow0 < J.OW; ow0 += kB) {
        const int nb = std::min(kB, J.OW - ow0);
        T acc[kB];
        for (int e = 0; e < kB; ++e) acc[e] = T(0);
        const T *p0 = row + (size_t)ow0 * J.stride_w;
        if (nb == kB) {
          for (int j = jb; j < je; ++j) {
            const T v = J.val[j];
            const T *p = p0 + J.off[j];
            for (int e = 0; e < kB; ++e) acc[e] = std::fma(v, p[(size_t)e * J.stride_w], acc[e]);
          }
        } else {
          for (int j = jb; j < je; ++j) {
            const T v = J.val[j];
            const T *p = p0 + J.off[j];
            for (int e = 0; e < nb; ++e) acc[e] = std::fma(v, p[(size_t)e * J.stride_w], acc[e]);
          }
        }
        for (int e = 0; e < nb; ++e) {
          T r = acc[e];
          if (J.bias) r += b;
          if (J.relu) r = r > T(0) ? r : T(0);
          outp[(size_t)oh * J.OW + ow0 + e] = r;
        }
      }
    }
  }
}

// Channels per block: as many as keep a tile's input window (the rows the tile's pixels span + the kernel's rows, whole
// padded rows) inside kL1Window bytes, in blocks of equal size -- and none at all where a row would then have fewer than
// kMinBlockNnz nonzeros per block: parking and fetching a tile's sums is 2 x NV vector moves per (row, block), and a
// 95 %-sparse pointwise layer touches so few channels per row that nothing is reused between rows anyway (GoogLeNet's
// 1x1 layers measured -10..-25 % with blocking forced on them; ResNet / AlexNet shapes +3..+80 % by CPU).
constexpr int kL1Window = 24 * 1024, kMinBlockNnz = 12;
template <typename T>
static long window_bytes(int OH, int OW, int PW, int span_rows) {
  constexpr int L = Vec<T>::L;
  if (OH < 1 || OW < 1 || PW < 1) return 0;
  const int Q = (OH - 1) * PW + OW;
  const int nvec = (Q + L - 1) / L;
  const int ntiles = (nvec + kMaxVecs - 1) / kMaxVecs;
  const int per_tile = (nvec + ntiles - 1) / ntiles;
  const long rows = (long)(per_tile * L + PW - 1) / PW + 1 + span_rows;
  return rows * PW * (long)sizeof(T);
}
template <typename T>
static int channel_block(int OH, int OW, int PW, int span_rows, int Cg, double avg_row_nnz) {
  if (OH < 1 || OW < 1 || PW < 1 || Cg < 2) return 0;
  const long per_channel = window_bytes<T>(OH, OW, PW, span_rows) * images_per_job<T>(OH, OW, PW, 1, 1);
  if (per_channel * Cg <= 2 * kL1Window) return 0;                     // the whole window (nearly) fits as it is
  const long cb_l1 = std::max<long>(1, kL1Window / per_channel);
  if (cb_l1 >= Cg || avg_row_nnz * (double)cb_l1 / Cg < kMinBlockNnz) return 0;
  const long n_blk = (Cg + cb_l1 - 1) / cb_l1;
  return (int)((Cg + n_blk - 1) / n_blk);
}

template <typename T>
static void run_group(const GroupJob<T> &J) {
  if (J.m_end <= J.m_begin || J.OH < 1 || J.OW < 1) return;
  if (J.stride_h == 1 && J.stride_w == 1) {
    run_unit_stride<T>(J);
  } else {
    GroupJob<T> one = J;
    one.n_img = 1;
    for (int i = 0; i < std::max(1, J.n_img); ++i) {
      one.in = J.in + (size_t)i * J.in_stride;
      one.out = J.out + (size_t)i * J.out_stride;
      run_any_stride<T>(one);
    }
  }
}

}  // namespace

#if ESC_CPU_ISA == 512
template <typename T> void run_group_avx512(const GroupJob<T> &job) { run_group<T>(job); }
template void run_group_avx512<float>(const GroupJob<float> &);
template void run_group_avx512<double>(const GroupJob<double> &);
template <typename T> int images_per_job_avx512(int OH, int OW, int PW, int stride_h, int stride_w) {
  return images_per_job<T>(OH, OW, PW, stride_h, stride_w);
}
template int images_per_job_avx512<float>(int, int, int, int, int);
template int images_per_job_avx512<double>(int, int, int, int, int);
template <typename T> long window_bytes_avx512(int OH, int OW, int PW, int span_rows) { return window_bytes<T>(OH, OW, PW, span_rows); }
template long window_bytes_avx512<float>(int, int, int, int);
template long window_bytes_avx512<double>(int, int, int, int);
template <typename T> int channel_block_avx512(int OH, int OW, int PW, int span_rows, int Cg, double avg_row_nnz) {
  return channel_block<T>(OH, OW, PW, span_rows, Cg, avg_row_nnz);
}
template int channel_block_avx512<float>(int, int, int, int, int, double);
template int channel_block_avx512<double>(int, int, int, int, int, double);
#else
size_t scratch_elems(int /*OH*/, int /*PW*/) { return 16 * 16; }   // one tile of the widest flavour (14 x 16 floats), rounded up
template <typename T> void run_group_avx2(const GroupJob<T> &job) { run_group<T>(job); }
template void run_group_avx2<float>(const GroupJob<float> &);
template void run_group_avx2<double>(const GroupJob<double> &);
template <typename T> int images_per_job_avx2(int OH, int OW, int PW, int stride_h, int stride_w) {
  return images_per_job<T>(OH, OW, PW, stride_h, stride_w);
}
template int images_per_job_avx2<float>(int, int, int, int, int);
template int images_per_job_avx2<double>(int, int, int, int, int);
template <typename T> long window_bytes_avx2(int OH, int OW, int PW, int span_rows) { return window_bytes<T>(OH, OW, PW, span_rows); }
template long window_bytes_avx2<float>(int, int, int, int);
template long window_bytes_avx2<double>(int, int, int, int);
template <typename T> int channel_block_avx2(int OH, int OW, int PW, int span_rows, int Cg, double avg_row_nnz) {
  return channel_block<T>(OH, OW, PW, span_rows, Cg, avg_row_nnz);
}
template int channel_block_avx2<float>(int, int, int, int, int, double);
template int channel_block_avx2<double>(int, int, int, int, int, double);
#endif

}  // namespace cpu
}  // namespace escoin
