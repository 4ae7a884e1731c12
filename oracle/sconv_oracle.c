/*
 * oracle/sconv_oracle.c -- TEST INFRASTRUCTURE ONLY (see sconv_oracle.h).
 *
 * Plain-C restatement of the reference CPU sconv forward path.  Nothing under
 * caffe-escoin_amd/ links this file; it exists so that tests, smoke() and the
 * bench's cpu_baseline leg have something to check the HIP path against.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -fopenmp, the reference's
 * g++ flags, Makefile:324,421-429 of the reference).
 */
#include "sconv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_out_dim(int in, int k, int pad, int stride, int dil) {
  /* conv_layer.cpp:16-19 */
  const int kernel_extent = dil * (k - 1) + 1;
  return (in + 2 * pad - kernel_extent) / stride + 1;
}

long oracle_padded_len(const oracle_conv_geom *g) {
  /* base_conv_layer.cpp:71 -- rows are W + pad_w long (a row's right padding is the next row's left padding), channels
   * H + pad_h rows (likewise), and the slack behind the last channel is pad_h * (W + 2 pad_w) floats.  That covers the
   * last row's right padding only when pad_h >= 1: with pad_h == 0 < pad_w the reference's kernels read pad_w floats
   * PAST its allocation at the last output columns of the last channel's last row (undefined there; none of its models
   * has such a layer).  The oracle defines those reads as the zero padding the layer specifies: pad_w floats more,
   * zero like the rest (found by tools/fuzz_parity.py: every GPU kernel agreed with each other and not with the oracle). */
  return (long)g->C * (g->H + g->pad_h) * (g->W + g->pad_w) +
         (long)g->pad_h * (g->W + 2 * g->pad_w) + (g->pad_h == 0 ? g->pad_w : 0);
}

int oracle_dense2csr(int M, int N, const float *A, float *values, int *colidx,
                     int *rowptr) {
  /* math_functions.cpp:92-105 */
  int nnz = 0;
  rowptr[0] = 0;
  for (int i = 0; i < M; i++) {
    int nnz_per_row = 0;
    for (int j = 0; j < N; j++) {
      if (A[(long)i * N + j] != 0) {
        values[nnz] = A[(long)i * N + j];
        colidx[nnz] = j;
        nnz_per_row++;
        nnz++;
      }
    }
    rowptr[i + 1] = rowptr[i] + nnz_per_row;
  }
  return nnz;
}

void oracle_stretch(int M, const int *rowptr, int *colidx, int KH, int KW,
                    int H, int W, int pad_h, int pad_w) {
  /* base_conv_layer.cpp:96-107 */
  for (int oc = 0; oc < M; ++oc) {
    for (int j = rowptr[oc]; j < rowptr[oc + 1]; ++j) {
      const int col = colidx[j];
      const int kernel_col = col % KW;
      const int kernel_row = (col / KW) % KH;
      const int in_channel = col / (KW * KH);
      colidx[j] = (in_channel * (H + pad_h) + kernel_row) * (W + pad_w) + kernel_col;
    }
  }
}

void oracle_pad_input(const oracle_conv_geom *g, const float *image,
                      float *padded) {
  /* base_conv_layer.cpp:615-620 */
  const int H = g->H, W = g->W, ph = g->pad_h, pw = g->pad_w;
  for (int c = 0; c < g->C; ++c) {
    for (int r = 0; r < H; ++r) {
      memcpy(padded + ((long)c * (H + ph) + r + ph) * (W + pw) + pw,
             image + ((long)c * H + r) * W, sizeof(float) * W);
    }
  }
}

void oracle_sconv(const float *input_padded, int in_channels, int height,
                  int width, int pad_h, int pad_w, int stride_h, int stride_w,
                  int dilation_h, int dilation_w, const int *rowptr,
                  const int *colidx, const float *values, int kernel_h,
                  int kernel_w, float *output, int out_channels) {
  /* math_functions.cpp:136-175 */
  (void)in_channels;
  const int output_h = oracle_out_dim(height, kernel_h, pad_h, stride_h, dilation_h);
  const int output_w = oracle_out_dim(width, kernel_w, pad_w, stride_w, dilation_w);
  if (dilation_h != 1 || dilation_w != 1) {
    /* :142-160 -- decodes the stretched index back to (ic, kr, kc) */
    for (int orow = 0; orow < output_h; ++orow) {
      for (int ocol = 0; ocol < output_w; ++ocol) {
        for (int oc = 0; oc < out_channels; ++oc) {
          float sum = 0;
          for (int j = rowptr[oc]; j < rowptr[oc + 1]; ++j) {
            const int col = colidx[j];
            const int kernel_col = col % (width + pad_w);
            const int kernel_row = (col / (width + pad_w)) % (height + pad_h);
            const int in_channel = col / ((width + pad_w) * (height + pad_h));
            const int input_row = kernel_row * dilation_h + orow * stride_h;
            const int input_col = kernel_col * dilation_w + ocol * stride_w;
            sum = fmaf(values[j],
                       input_padded[((long)in_channel * (height + pad_h) + input_row) *
                                        (width + pad_w) + input_col],
                       sum);
          }
          output[((long)oc * output_h + orow) * output_w + ocol] = sum;
        }
      }
    }
  } else {
    /* :162-174 */
    for (int orow = 0; orow < output_h; ++orow) {
      for (int ocol = 0; ocol < output_w; ++ocol) {
        const float *in_temp2 =
            input_padded + (long)orow * stride_h * (width + pad_w) + ocol * stride_w;
        for (int oc = 0; oc < out_channels; ++oc) {
          float sum = 0;
          for (int j = rowptr[oc]; j < rowptr[oc + 1]; ++j) {
            sum = fmaf(values[j], in_temp2[colidx[j]], sum);
          }
          output[((long)oc * output_h + orow) * output_w + ocol] = sum;
        }
      }
    }
  }
}

void oracle_bias(float *output, const float *bias, int M, int out_spatial) {
  /* base_conv_layer.cpp:663-669: C = 1*bias(Mx1)*ones(1xS) + 1*C */
  for (int oc = 0; oc < M; ++oc)
    for (int s = 0; s < out_spatial; ++s) output[(long)oc * out_spatial + s] += bias[oc];
}

void oracle_dense_gemm(const oracle_conv_geom *g, const float *image,
                       const float *weights, float *output) {
  /* base_conv_layer.cpp:532-566 with im2col.cpp:19-57 folded in: the column
   * matrix is never materialised, the k loop walks (ic, kr, kc) in im2col row
   * order and skips nothing (zeros of the halo are multiplied like im2col's). */
  const int OH = oracle_out_dim(g->H, g->KH, g->pad_h, g->stride_h, g->dil_h);
  const int OW = oracle_out_dim(g->W, g->KW, g->pad_w, g->stride_w, g->dil_w);
  const int Cg = g->C / g->group, Mg = g->M / g->group;
  for (int grp = 0; grp < g->group; ++grp) {
    for (int m = 0; m < Mg; ++m) {
      const int oc = grp * Mg + m;
      const float *w = weights + (long)oc * Cg * g->KH * g->KW;
      for (int oh = 0; oh < OH; ++oh) {
        for (int ow = 0; ow < OW; ++ow) {
          float sum = 0;
          for (int ic = 0; ic < Cg; ++ic) {
            const float *im = image + ((long)(grp * Cg + ic) * g->H) * g->W;
            for (int kr = 0; kr < g->KH; ++kr) {
              const int ir = -g->pad_h + kr * g->dil_h + oh * g->stride_h;
              for (int kc = 0; kc < g->KW; ++kc) {
                const int icol = -g->pad_w + kc * g->dil_w + ow * g->stride_w;
                float v = 0;
                if ((unsigned)ir < (unsigned)g->H && (unsigned)icol < (unsigned)g->W)
                  v = im[(long)ir * g->W + icol];
                sum = fmaf(w[(ic * g->KH + kr) * g->KW + kc], v, sum);
              }
            }
          }
          output[((long)oc * OH + oh) * OW + ow] = sum;
        }
      }
    }
  }
}

static int conv_forward_impl(const oracle_conv_geom *g, int N, const float *bottom,
                             const float *weights_dense, const float *bias, int relu,
                             float *top, int n_threads, int use_gate) {
  const int group = g->group;
  const int Cg = g->C / group, Mg = g->M / group;
  const int kdim = Cg * g->KH * g->KW;               /* kernel_dim_            */
  const int OH = oracle_out_dim(g->H, g->KH, g->pad_h, g->stride_h, g->dil_h);
  const int OW = oracle_out_dim(g->W, g->KW, g->pad_w, g->stride_w, g->dil_w);
  const long bottom_dim = (long)g->C * g->H * g->W;  /* base_conv_layer.cpp:518 */
  const long top_dim = (long)g->M * OH * OW;         /* :519                   */
  const long weight_offset = (long)Mg * kdim;        /* :60  count()/group_    */
  const int row_offset = Mg + 1;                     /* :63                    */
  const long plen = oracle_padded_len(g);

  /* --- WeightAlign(), base_conv_layer.cpp:46-107 (CPU, non-BLOCKED branch) --- */
  float *values = (float *)malloc(sizeof(float) * weight_offset * group);
  int *colidx = (int *)malloc(sizeof(int) * weight_offset * group);
  int *rowptr = (int *)malloc(sizeof(int) * row_offset * group);
  if (!values || !colidx || !rowptr) return -1;
  for (int grp = 0; grp < group; ++grp) {
    oracle_dense2csr(Mg, kdim, weights_dense + weight_offset * grp,
                     values + weight_offset * grp, colidx + weight_offset * grp,
                     rowptr + row_offset * grp);
    oracle_stretch(Mg, rowptr + row_offset * grp, colidx + weight_offset * grp, g->KH,
                   g->KW, g->H, g->W, g->pad_h, g->pad_w);
  }
  /* forward_cpu_sconv's gate looks at group 0 only, base_conv_layer.cpp:572-577 */
  const int gate_dense =
      use_gate && ((float)rowptr[Mg] / (float)((long)Mg * kdim) > 0.5f);
  const int padded = (g->pad_h != 0 || g->pad_w != 0); /* :601 */

  if (n_threads < 1) n_threads = 1;
  int rc = 0;
#pragma omp parallel num_threads(n_threads)
  {
    float *input_padded = NULL;
    if (padded && !gate_dense) {
      input_padded = (float *)calloc(plen, sizeof(float)); /* :78-80 */
      if (!input_padded) {
#pragma omp atomic write
        rc = -1;
      }
    }
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) { /* conv_layer.cpp:44 */
      const float *image = bottom + n * bottom_dim;
      float *out = top + n * top_dim;
      if (rc != 0) continue;
      if (gate_dense) {
        oracle_dense_gemm(g, image, weights_dense, out);
      } else {
        const float *in_p = image;
        if (padded) {
          oracle_pad_input(g, image, input_padded);
          in_p = input_padded;
        }
        for (int grp = 0; grp < group; ++grp) { /* base_conv_layer.cpp:626-658 */
          const float *in_temp =
              in_p + (long)Cg * grp * (g->H + g->pad_h) * (g->W + g->pad_w);
          oracle_sconv(in_temp, Cg, g->H, g->W, g->pad_h, g->pad_w, g->stride_h,
                       g->stride_w, g->dil_h, g->dil_w, rowptr + row_offset * grp,
                       colidx + weight_offset * grp, values + weight_offset * grp, g->KH,
                       g->KW, out + (long)Mg * OH * OW * grp, Mg);
        }
      }
      if (bias) oracle_bias(out, bias, g->M, OH * OW); /* conv_layer.cpp:55-58 */
      if (relu) {
        for (long i = 0; i < top_dim; ++i) out[i] = out[i] > 0.f ? out[i] : 0.f;
      }
    }
    free(input_padded);
  }
  free(values);
  free(colidx);
  free(rowptr);
  return rc;
}

int oracle_conv_forward(const oracle_conv_geom *g, int N, const float *bottom,
                        const float *weights_dense, const float *bias, int relu,
                        float *top, int n_threads) {
  return conv_forward_impl(g, N, bottom, weights_dense, bias, relu, top, n_threads, 1);
}

int oracle_conv_forward_nogate(const oracle_conv_geom *g, int N, const float *bottom,
                               const float *weights_dense, const float *bias, int relu,
                               float *top, int n_threads) {
  return conv_forward_impl(g, N, bottom, weights_dense, bias, relu, top, n_threads, 0);
}
