// dense_mfma.hip -- the dense fallback: implicit-GEMM convolution on the fp32 matrix cores.
//
// The reference sends a layer whose group-0 density exceeds 0.2 to im2col + cublasSgemm
// (forward_gpu_gemm, src/caffe/layers/base_conv_layer.cpp:713-746, gate :750-755, :805-811).
// Here the column matrix is never materialised: per conv group
//     C[Mg x P] = A[Mg x K] * B[K x P],   K = Cg*KH*KW,  P = N*OH*OW,
// A = the dense weights, B = the im2col view gathered on the fly (any stride / pad / dilation).
//
// Workgroup = 4 waves on a (64 * WROWS) x 128 tile, WROWS = 2 (waves 2 x 2, 64 x 64 each) or 1
// (waves 1 x 4, 64 x 32 each, for layers with <= 64 output channels per group).  A wave's tile is
// 2 x {2,1} blocks of v_mfma_f32_32x32x2_f32 (exact fp32: one fmaf per product, k ascending), so
// a k-pair costs 4 (3) LDS fragment reads for 4 (2) MFMAs.  Operands are staged through LDS
// k-major -- a lane's A[i][k] / B[k][j] fragment is one conflict-free ds_read_b32 -- in k-steps of
// 16, double buffered: the global loads of step s+1 (A along k, B gathered along the pixel axis,
// coalesced) fly under the 32 (16) MFMAs of step s and there is one barrier per step.  Bias and
// ReLU are fused in the epilogue.
#include <hip/hip_runtime.h>

#include "escoin_plan.h"

namespace escoin {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 16, kLdsPad = 4;

struct DenseArgs {
  const float *__restrict__ in;
  const float *__restrict__ w;     // dense M x (Cg*KH*KW)
  const float *__restrict__ bias;
  float *__restrict__ out;
  int n_images, C, H, W, M, OH, OW, KH, KW;
  int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w;
  int Cg, Mg, K, P, relu;
  unsigned long long group_mask;   // conv groups this launch covers (all ones: every group)
};

template <int WROWS>
__global__ void __launch_bounds__(256) escoin_dense_mfma_kernel(DenseArgs a) {
  constexpr int BM = 64 * WROWS;
  constexpr int WCOLS = 4 / WROWS;           // waves along the pixel axis
  constexpr int WN = kBN / WCOLS;            // columns per wave: 64 or 32
  constexpr int NB = WN / 32;                // 32-column MFMA blocks per wave
  constexpr int A_PER = BM * kBK / 256;      // A elements staged per thread and step: 8 or 4
  constexpr int B_PER = kBN * kBK / 256;     // B elements staged per thread and step: 8
  __shared__ float sA[2][kBK][BM + kLdsPad];
  __shared__ float sB[2][kBK][kBN + kLdsPad];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int cg = a.group_mask == ~0ull ? (int)blockIdx.z : nth_set_bit(a.group_mask, blockIdx.z);
  const int m0 = blockIdx.y * BM;                   // first output channel (group-local)
  const int p0 = blockIdx.x * kBN;                  // first flattened output pixel
  const int khw = a.KH * a.KW;
  const int ohw = a.OH * a.OW;

  // ---- B staging: this thread gathers column p_local for k_local = kb + 2 q, q < B_PER ----
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

  // ---- A staging: thread loads A[m0 + am][k0 + ak .. ak + A_PER) (contiguous in memory) ----
  constexpr int A_TPR = kBK / A_PER;                // threads per A row: 2 or 4
  const int am = tid / A_TPR, ak = (tid % A_TPR) * A_PER;
  const bool am_ok = m0 + am < a.Mg;
  const float *wrow = a.w + ((size_t)cg * a.Mg + (am_ok ? m0 + am : 0)) * a.K;
  const bool a_vec = (a.K & 3) == 0;                // rows 16-byte aligned (hipMalloc'd base)

  f32x16 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x16{0};

  float av[A_PER], bv[B_PER];
  auto gather = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER; i += 4) {
      const int k = k0 + ak + i;
      if (am_ok && a_vec && k + 3 < a.K) {
        const float4 v = *reinterpret_cast<const float4 *>(wrow + k);
        av[i] = v.x; av[i + 1] = v.y; av[i + 2] = v.z; av[i + 3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[i + e] = (am_ok && k + e < a.K) ? wrow[k + e] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < B_PER; ++q) {
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
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) sA[buf][ak + i][am] = av[i];
#pragma unroll
    for (int q = 0; q < B_PER; ++q) sB[buf][kb + 2 * q][p_local] = bv[q];
  };

  gather(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < a.K; k0 += kBK, buf ^= 1) {
    const bool more = k0 + kBK < a.K;
    if (more) gather(k0 + kBK);                     // flies under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      // 32x32x2: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
      const int ks = kk + (lane >> 5);
      const float fa0 = sA[buf][ks][wm * 64 + (lane & 31)];
      const float fa1 = sA[buf][ks][wm * 64 + 32 + (lane & 31)];
      float fb[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) fb[j] = sB[buf][ks][wn * WN + 32 * j + (lane & 31)];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb[j], acc[1][j], 0, 0, 0);
      }
    }
    if (more) stage(buf ^ 1);                       // the other buffer: last read one step ago
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int pj = p0 + wn * WN + 32 * j + (lane & 31);
    if (pj >= a.P) continue;
    const int nn = pj / ohw;
    const int rr = pj - nn * ohw;
    float *obase = a.out + ((size_t)nn * a.M + (size_t)cg * a.Mg) * ohw + rr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m >= a.Mg) continue;
        float v = acc[i][j][reg];
        if (a.bias) v += a.bias[cg * a.Mg + m];
        if (a.relu) v = fmaxf(v, 0.f);
        obase[(size_t)m * ohw] = v;
      }
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
  const int bm = g.Mg <= 64 ? 64 : 128;
  a.group_mask = p->use_dense ? ~0ull : p->dense_mask;
  dim3 grid((unsigned)((P + kBN - 1) / kBN), (unsigned)((g.Mg + bm - 1) / bm),
            (unsigned)(p->use_dense ? g.d.group : p->n_dense_groups));
  if (grid.y > 65535u || grid.z > 65535u) return fail(ESCOIN_EINVAL, "dense kernel: grid too large");
  if (bm == 64) hipLaunchKernelGGL(escoin_dense_mfma_kernel<1>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(escoin_dense_mfma_kernel<2>, grid, dim3(256), 0, stream, a);
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

}  // namespace escoin
