/*
 * oracle/sconv_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's direct-sparse-convolution forward path
 * (chenxuhao/caffe-escoin, CPU "sconv" mode).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The product library
 * (libescoin_hip.so) never does.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference tree).  Parity status: PINNED -- tests/test_oracle.py checks these
 * functions bit-for-bit against oracle/_ref (the reference's own header kernel
 * include/caffe/util/sconv.hpp:594-678 compiled in place) and against the
 * golden fixtures under tests/golden/ that oracle/_ref generated.
 */
#ifndef ESCOIN_SCONV_ORACLE_H_
#define ESCOIN_SCONV_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Convolution geometry of one layer call (NCHW fp32, int32 indices). */
typedef struct oracle_conv_geom {
  int C, H, W;          /* input channels (all groups), input height/width   */
  int M;                /* output channels (all groups)                      */
  int KH, KW;           /* kernel                                            */
  int pad_h, pad_w;
  int stride_h, stride_w;
  int dil_h, dil_w;
  int group;
} oracle_conv_geom;

/* conv_layer.cpp:8-22  O = (I + 2p - (d(k-1)+1))/s + 1 */
int oracle_out_dim(int in, int k, int pad, int stride, int dil);

/* base_conv_layer.cpp:71,596  C(H+ph)(W+pw) + ph(W+2pw) */
long oracle_padded_len(const oracle_conv_geom *g);

/* math_functions.cpp:92-105 (the non-MKL branch): row-major scan keeping
 * A[i][j] != 0 in ascending column order.  Returns nnz. */
int oracle_dense2csr(int M, int N, const float *A, float *values, int *colidx,
                     int *rowptr);

/* base_conv_layer.cpp:96-107: col -> (ic*(H+ph)+kr)*(W+pw)+kc, in place. */
void oracle_stretch(int M, const int *rowptr, int *colidx, int KH, int KW,
                    int H, int W, int pad_h, int pad_w);

/* base_conv_layer.cpp:601-620: one image C*H*W -> shared-halo padded layout.
 * `padded` must hold oracle_padded_len() floats and be zero outside the data
 * (the reference memsets it once, base_conv_layer.cpp:80). */
void oracle_pad_input(const oracle_conv_geom *g, const float *image,
                      float *padded);

/* math_functions.cpp:128-176 caffe_cpu_sconv<float>: one image, one group.
 * No bias.  Sequential fp32 accumulation from 0 in CSR order, one fused
 * multiply-add per nonzero (the reference's g++ flags -mfma contract
 * `sum += a*b`, Makefile:421-429). */
void oracle_sconv(const float *input_padded, int in_channels, int height,
                  int width, int pad_h, int pad_w, int stride_h, int stride_w,
                  int dilation_h, int dilation_w, const int *rowptr,
                  const int *colidx, const float *values, int kernel_h,
                  int kernel_w, float *output, int out_channels);

/* base_conv_layer.cpp:663-669 forward_cpu_bias: out[oc][:] += bias[oc]. */
void oracle_bias(float *output, const float *bias, int M, int out_spatial);

/* base_conv_layer.cpp:532-566 forward_cpu_gemm (LOWERED_GEMM branch) restated
 * as im2col (im2col.cpp:19-57) + a plain k-ordered fp32 GEMM.  One image. */
void oracle_dense_gemm(const oracle_conv_geom *g, const float *image,
                       const float *weights, float *output);

/* conv_layer.cpp:25-63 Forward_cpu in SCONV mode: for every image
 * forward_cpu_sconv (base_conv_layer.cpp:569-661: group-0 density gate >0.5 ->
 * dense GEMM, pad copy, per-group caffe_cpu_sconv) then bias once
 * (conv_layer.cpp:55-58).  `bias` may be NULL (no bias_term).  `relu` != 0
 * applies max(x,0) after bias (ConvolutionReLU semantics,
 * conv_relu_layer.cu:8-30 / sconv.hpp:637,666).  n_threads > 1 parallelises
 * over the batch the way the ICC build does (conv_layer.cpp:41-43) with
 * per-thread padded buffers (base_conv_layer.cpp:72-75).
 * Returns 0, or -1 on allocation failure. */
int oracle_conv_forward(const oracle_conv_geom *g, int N, const float *bottom,
                        const float *weights_dense, const float *bias,
                        int relu, float *top, int n_threads);

/* Same as oracle_conv_forward but never takes the dense gate (pure sconv for
 * any density) -- used to check kernels on dense-ish CSR. */
int oracle_conv_forward_nogate(const oracle_conv_geom *g, int N,
                               const float *bottom, const float *weights_dense,
                               const float *bias, int relu, float *top,
                               int n_threads);

#ifdef __cplusplus
}
#endif
#endif
