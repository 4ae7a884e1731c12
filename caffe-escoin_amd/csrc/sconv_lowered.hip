// sconv_lowered.hip -- the LOWERED_SPARSE comparator: im2col + sparse(CSR) x dense.
//
// The reference's conv_mode 1 lowers every image to a column matrix (im2col_gpu,
// src/caffe/util/im2col.cu:9-62; skipped for 1x1 / stride 1 / pad 0, base_conv_layer.cpp:374-379)
// and multiplies the CSR weights with it through cuSPARSE (caffe_gpu_sparse_csrmm,
// math_functions.cu:48-62, called from forward_gpu_gemm, base_conv_layer.cpp:724-736).  It is
// the "library baseline" the direct kernels were measured against (run.sh:8-12).  This file is
// that baseline on MI355X -- hand-written, no rocSPARSE dependency -- so that the direct path can
// be compared with lowering on the same chip (tools/crossover.py).  It is NOT the product path.
//
//   escoin_im2col_kernel   one thread per column-matrix element of a chunk of images; writes
//       are coalesced along the pixel axis.
//   escoin_csrmm_kernel    C[M x N] = alpha * A_csr[M x K] * B[K x N] + beta * C, row-major B/C.
//       One wave per (row, 256-column strip): a lane owns 4 adjacent columns (16-byte loads of B),
//       the CSR row is walked with wave-uniform (scalar) loads, one fmaf per nonzero in CSR
//       order from 0 -- for a convolution that is the summation order of caffe_cpu_sconv, so the
//       lowered path is bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "escoin_plan.h"

namespace escoin {

struct Im2colArgs {
  const float *__restrict__ in;   // first image of the chunk, this conv group's channels
  float *__restrict__ col;        // [image][Cg*KH*KW][OH*OW]
  int Cg, H, W, KH, KW, OH, OW;
  int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w;
  long in_stride;                 // floats between consecutive images of the bottom blob
};

__global__ void __launch_bounds__(256) escoin_im2col_kernel(Im2colArgs a) {
  const int npix = a.OH * a.OW;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int k = blockIdx.y;                 // row of the column matrix: (c, kh, kw)
  const int n = blockIdx.z;
  const int khw = a.KH * a.KW;
  const int c = k / khw;
  const int r = k - c * khw;
  const int kh = r / a.KW, kw = r - kh * a.KW;
  const int oh = p / a.OW, ow = p - oh * a.OW;
  const int ih = oh * a.stride_h - a.pad_h + kh * a.dil_h;
  const int iw = ow * a.stride_w - a.pad_w + kw * a.dil_w;
  float v = 0.f;
  if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
    v = a.in[(size_t)n * a.in_stride + ((size_t)c * a.H + ih) * a.W + iw];
  a.col[((size_t)n * gridDim.y + k) * npix + p] = v;
}

struct CsrmmArgs {
  const float *__restrict__ vals;
  const int *__restrict__ rowptr;   // M + 1 entries, offsets into vals / colidx (any base)
  const int *__restrict__ colidx;   // plain column indices, or packed taps when taps != 0
  const float *__restrict__ B;
  float *__restrict__ C;
  const float *__restrict__ bias;   // per row, nullable (convolution use)
  int M, N, K, taps, KH, KW, relu;
  float alpha, beta;
  long b_stride, c_stride;          // floats between consecutive problems of the batch (grid.z)
};

__global__ void __launch_bounds__(256) escoin_csrmm_kernel(CsrmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int m = blockIdx.y * 4 + wave;
  if (m >= a.M) return;
  const int col0 = blockIdx.x * 256 + lane * 4;
  if (col0 >= a.N) return;
  const float *__restrict__ B = a.B + (size_t)blockIdx.z * a.b_stride;
  float *__restrict__ C = a.C + (size_t)blockIdx.z * a.c_stride + (size_t)m * a.N + col0;
  const bool vec = (a.N & 3) == 0 && col0 + 3 < a.N;
  const int jb = a.rowptr[m], je = a.rowptr[m + 1];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int j = jb; j < je; ++j) {
    const float v = a.vals[j];
    int k = a.colidx[j];
    if (a.taps) k = ((k >> 16) * a.KH + ((k >> 8) & 0xFF)) * a.KW + (k & 0xFF);   // packed (ic, kr, kc)
    const float *row = B + (size_t)k * a.N + col0;
    if (vec) {
      const float4 b = *reinterpret_cast<const float4 *>(row);
      s0 = fmaf(v, b.x, s0); s1 = fmaf(v, b.y, s1); s2 = fmaf(v, b.z, s2); s3 = fmaf(v, b.w, s3);
    } else {
      s0 = fmaf(v, row[0], s0);
      if (col0 + 1 < a.N) s1 = fmaf(v, row[1], s1);
      if (col0 + 2 < a.N) s2 = fmaf(v, row[2], s2);
      if (col0 + 3 < a.N) s3 = fmaf(v, row[3], s3);
    }
  }
  float o[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (col0 + e >= a.N) break;
    float r = a.alpha == 1.f ? o[e] : a.alpha * o[e];
    if (a.beta != 0.f) r += a.beta * C[e];
    if (a.bias) r += a.bias[m];
    if (a.relu) r = fmaxf(r, 0.f);
    C[e] = r;
  }
}

const char *lowered_kernel_name() { return "escoin_csrmm_kernel"; }

static int launch_csrmm(const CsrmmArgs &a, int batch, hipStream_t stream) {
  dim3 grid((unsigned)((a.N + 255) / 256), (unsigned)((a.M + 3) / 4), (unsigned)batch);
  if (grid.y > 65535u || grid.z > 65535u) return fail(ESCOIN_EINVAL, "csrmm: grid dimension exceeds 65535");
  hipLaunchKernelGGL(escoin_csrmm_kernel, grid, dim3(256), 0, stream, a);
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

// caffe_gpu_sparse_csrmm<float>, math_functions.cu:48-62 (no transpose scratch: C comes out
// row-major directly).
int csrmm(int M, int N, int K, float alpha, const float *vals, const int *rowptr, const int *colidx,
          const float *B, float beta, float *C, hipStream_t stream) {
  CsrmmArgs a;
  a.vals = vals; a.rowptr = rowptr; a.colidx = colidx; a.B = B; a.C = C; a.bias = nullptr;
  a.M = M; a.N = N; a.K = K; a.taps = 0; a.KH = a.KW = 1; a.relu = 0; a.alpha = alpha; a.beta = beta;
  a.b_stride = a.c_stride = 0;
  return launch_csrmm(a, 1, stream);
}

// Forward_gpu in conv_mode LOWERED_SPARSE: chunks of images so that the column buffer stays
// within kColBytes (the reference lowers one image at a time into col_buffer_).
// caffe_gpu_sparse_csrmm<double>, math_functions.cu:64-78: the same product for Dtype = double (one lane per column,
// CSR order from zero, fp64 fused multiply-add).
__global__ void __launch_bounds__(256) escoin_csrmm_f64_kernel(int M, int N, double alpha, const double *__restrict__ vals,
                                                               const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                               const double *__restrict__ B, double beta, double *__restrict__ C) {
  const int m = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (m >= M || col >= N) return;
  double s = 0.0;
  for (int j = rowptr[m]; j < rowptr[m + 1]; ++j) s = fma(vals[j], B[(size_t)colidx[j] * N + col], s);
  double r = alpha == 1.0 ? s : alpha * s;
  if (beta != 0.0) r += beta * C[(size_t)m * N + col];
  C[(size_t)m * N + col] = r;
}

int csrmm_f64(int M, int N, int K, double alpha, const double *vals, const int *rowptr, const int *colidx,
              const double *B, double beta, double *C, hipStream_t stream) {
  (void)K;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)M, 1);
  if (grid.y > 65535u) return fail(ESCOIN_EINVAL, "csrmm_f64: more than 65535 rows");
  hipLaunchKernelGGL(escoin_csrmm_f64_kernel, grid, dim3(256), 0, stream, M, N, alpha, vals, rowptr, colidx, B, beta, C);
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

int launch_lowered(escoin_plan *p, const float *bottom, const float *bias, float *top, int n_images,
                   hipStream_t stream) {
  const Geometry &g = p->g;
  const int npix = g.OH * g.OW;
  const bool pointwise = g.d.KH == 1 && g.d.KW == 1 && g.d.stride_h == 1 && g.d.stride_w == 1 &&
                         g.d.pad_h == 0 && g.d.pad_w == 0;   // is_1x1_, base_conv_layer.cpp:374-379
  constexpr size_t kColBytes = 512ull << 20;
  const size_t per_image = (size_t)g.kdim * npix * sizeof(float);
  int chunk = n_images;
  if (!pointwise) {
    chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_images, kColBytes / per_image));
    const size_t need = per_image * chunk;
    if (p->col_bytes < need) {
      // the plan never holds a size for a buffer it no longer has: a failed hipMalloc leaves
      // d_col = nullptr AND col_bytes = 0, so a later, smaller forward allocates again
      if (p->d_col) (void)hipFree(p->d_col);
      p->d_col = nullptr;
      p->device_bytes -= p->col_bytes;
      p->col_bytes = 0;
      ESCOIN_HIP_TRY(hipMalloc(&p->d_col, need));
      p->device_bytes += need;
      p->col_bytes = need;
    }
  }
  const long in_image = (long)g.d.C * g.d.H * g.d.W, out_image = (long)g.d.M * npix;
  for (int n0 = 0; n0 < n_images; n0 += chunk) {
    const int nb = std::min(chunk, n_images - n0);
    for (int grp = 0; grp < g.d.group; ++grp) {
      const float *in_g = bottom + (size_t)n0 * in_image + (size_t)grp * g.Cg * g.d.H * g.d.W;
      const float *B = in_g;
      long b_stride = in_image;
      if (!pointwise) {
        Im2colArgs ia;
        ia.in = in_g; ia.col = p->d_col; ia.Cg = g.Cg; ia.H = g.d.H; ia.W = g.d.W; ia.KH = g.d.KH;
        ia.KW = g.d.KW; ia.OH = g.OH; ia.OW = g.OW; ia.pad_h = g.d.pad_h; ia.pad_w = g.d.pad_w;
        ia.stride_h = g.d.stride_h; ia.stride_w = g.d.stride_w; ia.dil_h = g.d.dil_h; ia.dil_w = g.d.dil_w;
        ia.in_stride = in_image;
        dim3 grid((unsigned)((npix + 255) / 256), (unsigned)g.kdim, (unsigned)nb);
        if (grid.y > 65535u || grid.z > 65535u) return fail(ESCOIN_EINVAL, "im2col: grid dimension exceeds 65535");
        hipLaunchKernelGGL(escoin_im2col_kernel, grid, dim3(256), 0, stream, ia);
        ESCOIN_HIP_TRY(hipGetLastError());
        B = p->d_col;
        b_stride = (long)g.kdim * npix;
      }
      CsrmmArgs a;
      a.vals = p->d_vals; a.rowptr = p->d_rowptr + (size_t)grp * g.Mg; a.colidx = p->d_taps;
      a.B = B; a.C = top + (size_t)n0 * out_image + (size_t)grp * g.Mg * npix;
      a.bias = bias ? bias + (size_t)grp * g.Mg : nullptr;
      a.M = g.Mg; a.N = npix; a.K = g.kdim; a.taps = 1; a.KH = g.d.KH; a.KW = g.d.KW;
      a.relu = g.d.fuse_relu; a.alpha = 1.f; a.beta = 0.f;
      a.b_stride = b_stride; a.c_stride = out_image;
      const int rc = launch_csrmm(a, nb, stream);
      if (rc != ESCOIN_OK) return rc;
    }
  }
  return ESCOIN_OK;
}

}  // namespace escoin
