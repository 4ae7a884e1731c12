// dense_mfma.hip -- the dense fallback: implicit-GEMM convolution on the fp32 matrix cores.
//
// The reference sends a layer whose group-0 density exceeds 0.2 to im2col + cublasSgemm
// (forward_gpu_gemm, src/caffe/layers/base_conv_layer.cpp:713-746, gate :750-755, :805-811).
// Here the column matrix is never materialised: per conv group
//     C[Mg x P] = A[Mg x K] * B[K x P],   K = Cg*KH*KW,  P = N*OH*OW,
// A = the dense weights, B = the im2col view gathered on the fly (any stride / pad / dilation).
// Workgroup = 4 waves computing a 64 x 128 tile; each wave owns 32 x 64 = two 32x32 fp32
// accumulators of v_mfma_f32_32x32x2_f32 (exact fp32: one fmaf per product, k ascending);
// operands are staged through LDS k-major so that a lane's A[i][k] / B[k][j] fragment is one
// conflict-free ds_read_b32.  Bias and ReLU are fused in the epilogue.
#include <hip/hip_runtime.h>

#include "escoin_plan.h"

namespace escoin {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 64, kBN = 128, kBK = 8;

struct DenseArgs {
  const float *__restrict__ in;
  const float *__restrict__ w;     // dense M x (Cg*KH*KW)
  const float *__restrict__ bias;
  float *__restrict__ out;
  int n_images, C, H, W, M, OH, OW, KH, KW;
  int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w;
  int Cg, Mg, K, P, relu;
};

__global__ void __launch_bounds__(256) escoin_dense_mfma_kernel(DenseArgs a) {
  __shared__ float sA[kBK][kBM + 4];   // +4: rows land on different banks for the staging writes
  __shared__ float sB[kBK][kBN + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;          // 2 x 2 waves: 32 rows x 64 cols each
  const int cg = blockIdx.z;
  const int m0 = blockIdx.y * kBM;                  // first output channel (group-local)
  const int p0 = blockIdx.x * kBN;                  // first flattened output pixel
  const int khw = a.KH * a.KW;
  const int ohw = a.OH * a.OW;

  // ---- B staging: this thread gathers column p_local for k_local = kb, kb+2, kb+4, kb+6 ----
  const int p_local = tid & (kBN - 1);
  const int kb = tid >> 7;                          // 0 or 1
  const int p = p0 + p_local;
  const bool p_ok = p < a.P;
  int n = 0, oh = 0, ow = 0;
  if (p_ok) {
    n = p / ohw;
    const int r = p - n * ohw;
    oh = r / a.OW;
    ow = r - oh * a.OW;
  }
  const int ih0 = oh * a.stride_h - a.pad_h, iw0 = ow * a.stride_w - a.pad_w;
  const float *img = a.in + ((size_t)n * a.C + (size_t)cg * a.Cg) * a.H * a.W;

  // ---- A staging: thread loads A[m0 + (tid>>2)][k0 + 2*(tid&3) + {0,1}] ----
  const int am = tid >> 2, ak = (tid & 3) * 2;
  const bool am_ok = m0 + am < a.Mg;
  const float *wrow = a.w + ((size_t)cg * a.Mg + (am_ok ? m0 + am : 0)) * a.K;

  f32x16 acc0 = {0}, acc1 = {0};
  float av0, av1, bv[4];
  // gathers one k-step's operands into registers (software pipelined: the loads of step k+1 fly
  // under the MFMAs of step k)
  auto gather = [&](int k0) {
    av0 = 0.f;
    av1 = 0.f;
    if (am_ok) {
      if (k0 + ak < a.K) av0 = wrow[k0 + ak];
      if (k0 + ak + 1 < a.K) av1 = wrow[k0 + ak + 1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + kb + 2 * q;
      float v = 0.f;
      if (p_ok && k < a.K) {
        const int ic = k / khw;
        const int r = k - ic * khw;
        const int kr = r / a.KW, kc = r - kr * a.KW;
        const int ih = ih0 + kr * a.dil_h, iw = iw0 + kc * a.dil_w;
        if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
          v = img[((size_t)ic * a.H + ih) * a.W + iw];
      }
      bv[q] = v;
    }
  };
  gather(0);
  for (int k0 = 0; k0 < a.K; k0 += kBK) {
    __syncthreads();                                // previous step's fragments are consumed
    sA[ak][am] = av0;
    sA[ak + 1][am] = av1;
#pragma unroll
    for (int q = 0; q < 4; ++q) sB[kb + 2 * q][p_local] = bv[q];
    __syncthreads();
    if (k0 + kBK < a.K) gather(k0 + kBK);
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      // 32x32x2: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
      const int ks = kk + (lane >> 5);
      const float fa = sA[ks][wm * 32 + (lane & 31)];
      const float fb0 = sB[ks][wn * 64 + (lane & 31)];
      const float fb1 = sB[ks][wn * 64 + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb1, acc1, 0, 0, 0);
    }
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int pj = p0 + wn * 64 + half * 32 + (lane & 31);
    if (pj >= a.P) continue;
    const int nn = pj / ohw;
    const int rr = pj - nn * ohw;
    float *obase = a.out + ((size_t)nn * a.M + (size_t)cg * a.Mg) * ohw + rr;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (m >= a.Mg) continue;
      float v = half ? acc1[reg] : acc0[reg];
      if (a.bias) v += a.bias[cg * a.Mg + m];
      if (a.relu) v = fmaxf(v, 0.f);
      obase[(size_t)m * ohw] = v;
    }
  }
}

const char *dense_kernel_name() { return "escoin_dense_mfma_kernel"; }

int launch_dense(const escoin_plan *p, const float *bottom, const float *bias, float *top,
                 int n_images, hipStream_t stream) {
  const Geometry &g = p->g;
  DenseArgs a;
  a.in = bottom; a.w = p->d_dense_w; a.bias = bias; a.out = top;
  a.n_images = n_images; a.C = g.d.C; a.H = g.d.H; a.W = g.d.W; a.M = g.d.M; a.OH = g.OH; a.OW = g.OW;
  a.KH = g.d.KH; a.KW = g.d.KW; a.pad_h = g.d.pad_h; a.pad_w = g.d.pad_w;
  a.stride_h = g.d.stride_h; a.stride_w = g.d.stride_w; a.dil_h = g.d.dil_h; a.dil_w = g.d.dil_w;
  a.Cg = g.Cg; a.Mg = g.Mg; a.K = g.kdim; a.relu = g.d.fuse_relu;
  const long P = (long)n_images * g.OH * g.OW;
  if (P >= (1l << 31)) return fail(ESCOIN_EINVAL, "dense kernel: N*OH*OW does not fit 31 bits");
  a.P = (int)P;
  dim3 grid((unsigned)((P + kBN - 1) / kBN), (unsigned)((g.Mg + kBM - 1) / kBM), (unsigned)g.d.group);
  if (grid.y > 65535u || grid.z > 65535u) return fail(ESCOIN_EINVAL, "dense kernel: grid too large");
  hipLaunchKernelGGL(escoin_dense_mfma_kernel, grid, dim3(256), 0, stream, a);
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

}  // namespace escoin
