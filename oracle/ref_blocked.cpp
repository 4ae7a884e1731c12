/*
 * oracle/ref_blocked.cpp -- TEST INFRASTRUCTURE ONLY (second translation unit of
 * oracle/_ref/libescoin_ref.so, compiled with -DUSE_ICC).
 *
 * The reference's "best effort" CPU path: the register-blocked SIMD kernel
 *   sconv_unit_stride<WIDTH, K, FUSE_RELU, PAD>       include/caffe/util/sconv.hpp:57-589
 * compiled from the reference header where it lies (its body is guarded by USE_ICC, which g++
 * accepts: only `#pragma unroll` hints are lost), behind restatements of
 *   - the switchboard caffe_cpu_blocked_sconv          src/caffe/util/math_functions.cpp:201-462
 *     (which <WIDTH, K> instantiations exist: W in {4,7,14,28,56} x K1, {3,7,12,13,14,28,56} x K3,
 *     {7,14,27,28} x K5; stride 1, square, pad = (K-1)/2)
 *   - WeightAlign's BLOCKED_SCONV branch                src/caffe/layers/base_conv_layer.cpp:118-240
 *     (column blocks of get_col_major_ic_block() input channels, one CSR per block)
 *   - the thread grouping of cpu::OpenMpManager         src/caffe/util/cpu_info.cpp:483-605
 *     (getNumThreadGroups / getBatchThreadPartition / getSimpleGroupedThreadPartition /
 *     barrierGroup): threads are split into at most `batch` groups; a group shares one padded
 *     image, its threads split the channels of the padded copy and the blocks of 16 output
 *     channels of the compute, with a group barrier in between (forward_cpu_sconv,
 *     base_conv_layer.cpp:604-622, math_functions.cpp:216-223).
 * plus a plan object that holds the CSR so that the timed forward does not redo dense -> CSR
 * (the reference does that once, in WeightAlign).
 *
 * Reference quirk not copied: its blocked kernel starts the accumulator at bias[oc] and
 * Forward_cpu adds the bias again afterwards (conv_layer.cpp:55-58).  Here the bias goes in once.
 */
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef USE_ICC
#error "compile this file with -DUSE_ICC (it enables the body of sconv_unit_stride)"
#endif
#define NOT_IMPLEMENTED abort()
#include "caffe/util/sconv.hpp"

#include "sconv_oracle.h"

#include "ref_plan.h"

namespace {

/* math_functions.cpp:224-440: the instantiations the reference's switchboard dispatches to. */
unit_stride_fn pick_unit_stride(int width, int k) {
#define ESC_US(W, K) \
  if (width == W && k == K) return &sconv_unit_stride<W, K, false>;
  ESC_US(4, 1) ESC_US(7, 1) ESC_US(14, 1) ESC_US(28, 1) ESC_US(56, 1)
  ESC_US(3, 3) ESC_US(7, 3) ESC_US(12, 3) ESC_US(13, 3) ESC_US(14, 3) ESC_US(28, 3) ESC_US(56, 3)
  ESC_US(7, 5) ESC_US(14, 5) ESC_US(27, 5) ESC_US(28, 5)
#undef ESC_US
  return nullptr;
}

/* cpu_info.cpp:483-540 for an explicit team of T threads. */
struct Grouping {
  int T, batch, n_groups, per_group;
  Grouping(int threads, int batch_size) : T(threads), batch(batch_size) {
    n_groups = T > 2 * batch ? batch : T;                 /* getNumThreadGroups */
    per_group = (T + n_groups - 1) / n_groups;            /* getNumThreadsPerGroup */
  }
  int group_of(int tid) const { return tid / per_group; }
  int rank_in_group(int tid) const { return tid % per_group; }
  int threads_in_group(int gid) const { return std::min(per_group, T - per_group * gid); }
  void batch_range(int gid, int *b, int *e) const {       /* getBatchThreadPartition */
    const int n_per = (batch + n_groups - 1) / n_groups;
    *b = std::min(n_per * gid, batch);
    *e = std::min(*b + n_per, batch);
  }
  void split(int tid, int work, int *b, int *e) const {   /* getSimpleGroupedThreadPartition */
    const int nin = threads_in_group(group_of(tid)), r = rank_in_group(tid);
    const int per = (work + nin - 1) / nin;
    *b = std::min(per * r, work);
    *e = std::min(*b + per, work);
  }
};

/* barrierGroup (cpu_info.cpp:590-601; synk::Barrier in the reference): sense-reversing spin. */
struct GroupBarrier {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  void wait(int n, int *local_sense) {
    if (n <= 1) return;
    *local_sense ^= 1;
    if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
      count.store(0, std::memory_order_relaxed);
      sense.store(*local_sense, std::memory_order_release);
    } else {
      while (sense.load(std::memory_order_acquire) != *local_sense) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
  }
};

}  // namespace

extern "C" {

int ref_blocked_supported(const oracle_conv_geom *g) {
  if (g->dil_h != 1 || g->dil_w != 1 || g->stride_h != 1 || g->stride_w != 1) return 0;
  if (g->H != g->W || g->KH != g->KW || g->pad_h != g->pad_w) return 0;
  if (g->KH != 2 * g->pad_h + 1) return 0;   /* "matched padding", math_functions.cpp:224 */
  return pick_unit_stride(g->H, g->KH) != nullptr;
}

ref_plan *ref_plan_create(const oracle_conv_geom *g, const float *weights_dense) {
  ref_plan *p = new ref_plan();
  p->g = *g;
  const int group = g->group;
  p->Cg = g->C / group;
  p->Mg = g->M / group;
  p->kdim = p->Cg * g->KH * g->KW;
  p->OH = oracle_out_dim(g->H, g->KH, g->pad_h, g->stride_h, g->dil_h);
  p->OW = oracle_out_dim(g->W, g->KW, g->pad_w, g->stride_w, g->dil_w);
  p->plen = oracle_padded_len(g);
  const long weight_offset = (long)p->Mg * p->kdim;
  const int row_offset = p->Mg + 1;
  p->values.resize(weight_offset * group);
  p->colidx.resize(weight_offset * group);
  p->rowptr.resize(row_offset * group);
  long nnz_total = 0;
  for (int grp = 0; grp < group; ++grp) {
    nnz_total += oracle_dense2csr(p->Mg, p->kdim, weights_dense + weight_offset * grp,
                                  p->values.data() + weight_offset * grp, p->colidx.data() + weight_offset * grp,
                                  p->rowptr.data() + row_offset * grp);
    oracle_stretch(p->Mg, p->rowptr.data() + row_offset * grp, p->colidx.data() + weight_offset * grp,
                   g->KH, g->KW, g->H, g->W, g->pad_h, g->pad_w);
  }
  p->kernel = ref_blocked_supported(g) ? pick_unit_stride(g->H, g->KH) : nullptr;
  p->ncolblocks = 0;
  if (p->kernel) {
    /* base_conv_layer.cpp:118-127: one block size for the layer, from the mean nnz per group */
    const int col_block_size = get_col_major_ic_block((int)(nnz_total / group), p->Mg, p->Cg);
    const int ncolblocks = g->C / col_block_size;
    const int per_group = ncolblocks / group;
    p->ncolblocks = ncolblocks;
    p->b_rowptr.assign(ncolblocks, std::vector<int>(p->Mg + 1, 0));
    p->b_colidx.assign(ncolblocks, std::vector<int>());
    p->b_values.assign(ncolblocks, std::vector<float>());
    const int PH = g->H + g->pad_h, PW = g->W + g->pad_w;
    for (int grp = 0; grp < group; ++grp) {   /* :214-238 */
      const int *rp = p->rowptr.data() + row_offset * grp;
      const int *ci = p->colidx.data() + weight_offset * grp;
      const float *va = p->values.data() + weight_offset * grp;
      for (int oc = 0; oc < p->Mg; ++oc) {
        for (int j = rp[oc]; j < rp[oc + 1]; ++j) {
          const int c = ci[j];
          const int ic = c / PW / PH;
          const int bcol = ic / col_block_size + per_group * grp;
          p->b_colidx[bcol].push_back(c);
          p->b_values[bcol].push_back(va[j]);
        }
        for (int i = per_group * grp; i < per_group * (grp + 1); ++i)
          p->b_rowptr[i][oc + 1] = (int)p->b_colidx[i].size();
      }
    }
    for (int i = 0; i < ncolblocks; ++i) {
      p->b_rowptr_p.push_back(p->b_rowptr[i].data());
      p->b_colidx_p.push_back(p->b_colidx[i].empty() ? p->colidx.data() : p->b_colidx[i].data());
      p->b_values_p.push_back(p->b_values[i].empty() ? p->values.data() : p->b_values[i].data());
    }
  }
  return p;
}

void ref_plan_destroy(ref_plan *p) { delete p; }
int ref_plan_has_blocked(const ref_plan *p) { return p && p->kernel ? 1 : 0; }
int ref_plan_ncolblocks(const ref_plan *p) { return p ? p->ncolblocks : 0; }

/* Thread grouping alone, for the tests: fills gid / batch range / split of `work` per thread. */
void ref_thread_partition(int n_threads, int batch, int work, int *gid, int *batch_begin, int *batch_end,
                          int *work_begin, int *work_end) {
  Grouping gr(n_threads, batch);
  for (int t = 0; t < n_threads; ++t) {
    gid[t] = gr.group_of(t);
    gr.batch_range(gid[t], &batch_begin[t], &batch_end[t]);
    gr.split(t, work, &work_begin[t], &work_end[t]);
  }
}

/* Whole-batch forward from an aligned plan.  kernel 0: caffe_cpu_sconv's loop nest
 * (ref_driver.cpp: the header's caffe_cpu_sconv_default with a zero bias, bias added after), plain
 * OpenMP over the batch; kernel 1: sconv_unit_stride with the reference's thread grouping. */
int ref_plan_forward_default(const ref_plan *p, int N, const float *bottom, const float *bias, float *top,
                             int n_threads);

int ref_plan_forward(const ref_plan *p, int N, const float *bottom, const float *bias, float *top,
                     int n_threads, int kernel) {
  if (kernel == 0) return ref_plan_forward_default(p, N, bottom, bias, top, n_threads);
  if (!p->kernel) return -2;
  const oracle_conv_geom *g = &p->g;
  const int group = g->group, Cg = p->Cg, Mg = p->Mg, OH = p->OH, OW = p->OW;
  const long bottom_dim = (long)g->C * g->H * g->W, top_dim = (long)g->M * OH * OW;
  const long plen = p->plen + (VLEN - 1);                 /* base_conv_layer.cpp:73,599 */
  const bool padded = g->pad_h != 0 || g->pad_w != 0;
  if (n_threads < 1) n_threads = 1;
  const Grouping gr(n_threads, N);
  const long scratch_per_thread = (long)OC_BLOCK * OH * ((OW + 16 - 1) / 16 * 16);   /* :241 */
  float *padded_all = nullptr, *scratch_all = nullptr;
  /* one padded image per thread GROUP (input_padded_ + input_padded_len * gid), zeroed once: the
   * copy only ever rewrites the interior */
  if (posix_memalign((void **)&padded_all, 4096, sizeof(float) * (size_t)gr.n_groups * plen)) return -1;
  memset(padded_all, 0, sizeof(float) * (size_t)gr.n_groups * plen);
  if (posix_memalign((void **)&scratch_all, 4096, sizeof(float) * (size_t)n_threads * scratch_per_thread)) {
    free(padded_all);
    return -1;
  }
  std::vector<float> zero_bias(g->M, 0.f);
  const float *bias_v = bias ? bias : zero_bias.data();
  std::vector<GroupBarrier> barriers(gr.n_groups);
  const int per_group_blocks = p->ncolblocks / group;
  const int num_oc_blocks = (Mg + OC_BLOCK - 1) / OC_BLOCK;
#pragma omp parallel num_threads(n_threads)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    if (tid < gr.T) {
      const int gid = gr.group_of(tid);
      const int nin = gr.threads_in_group(gid);
      int local_sense = 0;
      int nb = 0, ne = 0;
      gr.batch_range(gid, &nb, &ne);
      float *input_padded = padded_all + (size_t)gid * plen;
      float *scratch = scratch_all + (size_t)tid * scratch_per_thread;
      for (int n = nb; n < ne; ++n) {
        const float *image = bottom + n * bottom_dim;
        float *out = top + n * top_dim;
        const float *in_p = image;
        if (padded) {   /* base_conv_layer.cpp:604-622 */
          int cb = 0, ce = 0;
          gr.split(tid, g->C, &cb, &ce);
          for (int c = cb; c < ce; ++c)
            for (int r = 0; r < g->H; ++r)
              memcpy(input_padded + ((long)c * (g->H + g->pad_h) + r + g->pad_h) * (g->W + g->pad_w) + g->pad_w,
                     image + ((long)c * g->H + r) * g->W, sizeof(float) * g->W);
          barriers[gid].wait(nin, &local_sense);
          in_p = input_padded;
        }
        int ob = 0, oe = 0;
        gr.split(tid, num_oc_blocks, &ob, &oe);   /* math_functions.cpp:216-223 */
        const int oc_begin = std::min(ob * OC_BLOCK, Mg), oc_end = std::min(oe * OC_BLOCK, Mg);
        for (int grp = 0; grp < group; ++grp) {
          const float *in_temp = in_p + (long)Cg * grp * (g->H + g->pad_h) * (g->W + g->pad_w);
          p->kernel(in_temp, const_cast<const int **>(p->b_rowptr_p.data()) + grp * per_group_blocks,
                    const_cast<const int **>(p->b_colidx_p.data()) + grp * per_group_blocks,
                    const_cast<const float **>(p->b_values_p.data()) + grp * per_group_blocks,
                    per_group_blocks, bias_v + Mg * grp, out + (long)Mg * OH * OW * grp, oc_begin, oc_end, scratch,
                    Cg, Mg);
        }
        /* the next image's copy may not start before every thread of the group has finished */
        if (padded) barriers[gid].wait(nin, &local_sense);
      }
    }
  }
  free(padded_all);
  free(scratch_all);
  return 0;
}

}  // extern "C"
