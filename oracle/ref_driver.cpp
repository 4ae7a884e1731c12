/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C-ABI driver around the REFERENCE's own CPU kernel, compiled from the
 * reference sources where they lie (-I/root/reference/include): nothing of the
 * reference is copied into this repository.  Output goes to oracle/_ref/
 * (git-ignored, travels to the GPU box as a prebuilt .so).
 *
 * What is the reference here: include/caffe/util/sconv.hpp:594-678
 * `caffe_cpu_sconv_default<FUSE_RELU>` -- the same loop nest as
 * `caffe_cpu_sconv` (src/caffe/util/math_functions.cpp:128-176, which cannot be
 * compiled on its own: that translation unit pulls glog/boost/cblas) except
 * that the accumulator starts at bias[oc]; called with a zero bias it is
 * bit-identical to the g++ path's arithmetic.
 *
 * The header needs two things its includer normally provides: the
 * NOT_IMPLEMENTED macro (caffe/common.hpp:70, only reached by the ICC-only
 * sconv_unit_stride body) and the definitions of the profiling externs declared
 * at sconv.hpp:29,42.  They are supplied below; no reference header is replaced.
 *
 * The glue around the kernel (dense->CSR, index stretch, padded copy, group
 * offsets, bias) lives in translation units that cannot be built here, so the
 * batch driver below borrows the oracle's restatement of it (sconv_oracle.c)
 * and calls the reference kernel for the arithmetic.
 */
#include <algorithm>
#include <cassert>  // sconv.hpp:657 uses assert(); its includers pull <cassert> in
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NOT_IMPLEMENTED abort()
#include "caffe/util/sconv.hpp"

unsigned long long conv_cycles_of_this_batch[1024 * 16], transpose_cycle, pool_cycle;
int flop_cnt;

#include "sconv_oracle.h"
#include "ref_plan.h"

extern "C" {

/* One image, one group: straight call of the reference kernel. */
void ref_sconv_default(const float *input_padded, int in_channels, int height, int width,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h,
                       int dilation_w, const int *rowptr, const int *colidx,
                       const float *values, int kernel_h, int kernel_w, const float *bias,
                       float *output, int out_channels, int fuse_relu) {
  if (fuse_relu)
    caffe_cpu_sconv_default<true>(input_padded, in_channels, height, width, pad_h, pad_w,
                                  stride_h, stride_w, dilation_h, dilation_w, rowptr, colidx,
                                  values, kernel_h, kernel_w, bias, output, out_channels);
  else
    caffe_cpu_sconv_default<false>(input_padded, in_channels, height, width, pad_h, pad_w,
                                   stride_h, stride_w, dilation_h, dilation_w, rowptr, colidx,
                                   values, kernel_h, kernel_w, bias, output, out_channels);
}

/* Whole-batch forward in SCONV mode with the reference kernel doing the
 * arithmetic (no density gate: always the sparse kernel).  Bias is added once
 * afterwards, as conv_layer.cpp:55-58 does on the g++ path. */
int ref_conv_forward(const oracle_conv_geom *g, int N, const float *bottom,
                     const float *weights_dense, const float *bias, float *top,
                     int n_threads) {
  const int group = g->group;
  const int Cg = g->C / group, Mg = g->M / group;
  const int kdim = Cg * g->KH * g->KW;
  const int OH = oracle_out_dim(g->H, g->KH, g->pad_h, g->stride_h, g->dil_h);
  const int OW = oracle_out_dim(g->W, g->KW, g->pad_w, g->stride_w, g->dil_w);
  const long bottom_dim = (long)g->C * g->H * g->W, top_dim = (long)g->M * OH * OW;
  const long weight_offset = (long)Mg * kdim;
  const int row_offset = Mg + 1;
  const long plen = oracle_padded_len(g);
  std::vector<float> values(weight_offset * group);
  std::vector<int> colidx(weight_offset * group), rowptr(row_offset * group);
  for (int grp = 0; grp < group; ++grp) {
    oracle_dense2csr(Mg, kdim, weights_dense + weight_offset * grp,
                     values.data() + weight_offset * grp, colidx.data() + weight_offset * grp,
                     rowptr.data() + row_offset * grp);
    oracle_stretch(Mg, rowptr.data() + row_offset * grp, colidx.data() + weight_offset * grp,
                   g->KH, g->KW, g->H, g->W, g->pad_h, g->pad_w);
  }
  std::vector<float> zero_bias(g->M, 0.f);
  const bool padded = g->pad_h != 0 || g->pad_w != 0;
  if (n_threads < 1) n_threads = 1;
  int rc = 0;
#pragma omp parallel num_threads(n_threads)
  {
    float *input_padded = NULL;
    if (padded) {
      input_padded = (float *)calloc(plen, sizeof(float));
      if (!input_padded) {
#pragma omp atomic write
        rc = -1;
      }
    }
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) {
      if (rc != 0) continue;
      const float *image = bottom + n * bottom_dim;
      float *out = top + n * top_dim;
      const float *in_p = image;
      if (padded) {
        oracle_pad_input(g, image, input_padded);
        in_p = input_padded;
      }
      for (int grp = 0; grp < group; ++grp) {
        const float *in_temp = in_p + (long)Cg * grp * (g->H + g->pad_h) * (g->W + g->pad_w);
        caffe_cpu_sconv_default<false>(
            in_temp, Cg, g->H, g->W, g->pad_h, g->pad_w, g->stride_h, g->stride_w, g->dil_h,
            g->dil_w, rowptr.data() + row_offset * grp, colidx.data() + weight_offset * grp,
            values.data() + weight_offset * grp, g->KH, g->KW, zero_bias.data(),
            out + (long)Mg * OH * OW * grp, Mg);
      }
      if (bias) oracle_bias(out, bias, g->M, OH * OW);
    }
    free(input_padded);
  }
  return rc;
}

/* The same forward from an aligned ref_plan (ref_blocked.cpp: ref_plan_create): the CSR was built
 * once, as the reference's WeightAlign does; only the per-image work is left in the call. */
int ref_plan_forward_default(const ref_plan *p, int N, const float *bottom, const float *bias, float *top,
                             int n_threads) {
  const oracle_conv_geom *g = &p->g;
  const int group = g->group, Cg = p->Cg, Mg = p->Mg, OH = p->OH, OW = p->OW;
  const long bottom_dim = (long)g->C * g->H * g->W, top_dim = (long)g->M * OH * OW;
  const long weight_offset = (long)Mg * p->kdim;
  const int row_offset = Mg + 1;
  std::vector<float> zero_bias(g->M, 0.f);
  const bool padded = g->pad_h != 0 || g->pad_w != 0;
  if (n_threads < 1) n_threads = 1;
  int rc = 0;
#pragma omp parallel num_threads(n_threads)
  {
    float *input_padded = NULL;
    if (padded) {
      input_padded = (float *)calloc(p->plen, sizeof(float));
      if (!input_padded) {
#pragma omp atomic write
        rc = -1;
      }
    }
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) {
      if (rc != 0) continue;
      const float *image = bottom + n * bottom_dim;
      float *out = top + n * top_dim;
      const float *in_p = image;
      if (padded) {
        oracle_pad_input(g, image, input_padded);
        in_p = input_padded;
      }
      for (int grp = 0; grp < group; ++grp) {
        const float *in_temp = in_p + (long)Cg * grp * (g->H + g->pad_h) * (g->W + g->pad_w);
        caffe_cpu_sconv_default<false>(
            in_temp, Cg, g->H, g->W, g->pad_h, g->pad_w, g->stride_h, g->stride_w, g->dil_h,
            g->dil_w, p->rowptr.data() + row_offset * grp, p->colidx.data() + weight_offset * grp,
            p->values.data() + weight_offset * grp, g->KH, g->KW, zero_bias.data(),
            out + (long)Mg * OH * OW * grp, Mg);
      }
      if (bias) oracle_bias(out, bias, g->M, OH * OW);
    }
    free(input_padded);
  }
  return rc;
}

}  // extern "C"
