// dense_mfma.hip -- the dense fallback: implicit-GEMM convolution on the fp32 matrix cores.
//
// The reference sends a layer whose group-0 density exceeds 0.2 to im2col + cublasSgemm
// (forward_gpu_gemm, src/caffe/layers/base_conv_layer.cpp:713-746, gate :750-755, :805-811), and
// `caffe test -conv_mode 0` (LOWERED_GEMM) sends every layer there.  Here the column matrix is never
// materialised: per conv group
//     C[Mg x P] = A[Mg x K] * B[K x P],   K = Cg*KH*KW,  P = N*OH*OW,
// A = the dense weights, B = the im2col view gathered on the fly (any stride / pad / dilation).
//
// Workgroup = 4 waves on a BM x 128 tile (BM = 128: waves 2 x 2, 64 x 64 each; BM = 64 for layers
// with <= 64 output channels per group: waves 1 x 4, 64 x 32 each), two workgroups per CU.  A
// wave's tile is 2 x {2,1} blocks of v_mfma_f32_32x32x2_f32 (exact fp32 products and sums).
// k-steps of 32, double buffered through LDS: the global loads of step s + 1 fly under the 64 (32)
// MFMAs of step s, one barrier per step.
//   A tile  [m][32 k], 16-byte chunks XOR-swizzled by (m >> 1) & 7: a lane's four consecutive k of
//           one row come out as ONE conflict-free ds_read_b128.  The MFMA's two k-slots (lane
//           halves) take k0 + 4h + t, t = 0..3: the sum over k is just taken in another order.
//   B tile  [32 k][128 p + 4]: the staging writes are 16-byte (pointwise layers) or 4-byte
//           (gathered) and conflict-free, the fragment reads ds_read_b32.
//   im2col  the decode k -> (ic, kr, kc) is a per-layer table built at WeightAlign (ktab: element
//           offset inside the image and the (dy, dx) used for the border test), read with scalar
//           loads -- k is wave-uniform --, instead of two integer divisions per gathered element.
// Bias and ReLU are fused in the epilogue.
#include <hip/hip_runtime.h>

#include <vector>

#include "escoin_plan.h"

namespace escoin {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 32, kBPad = 4;

struct DenseArgs {
  const float *__restrict__ in;
  const float *__restrict__ w;     // dense (M + 128) x lda, zero-padded: rows of lda = K rounded up to 32
  const float *__restrict__ bias;
  const int2 *__restrict__ ktab;   // [K rounded up to 32]: {element offset of tap k in the image, dy | dx << 16}
  float *__restrict__ out;
  int n_images, C, H, W, M, OH, OW;
  int pad_h, pad_w, stride_h, stride_w;
  int Cg, Mg, K, lda, P, relu;
  int vec_b;                       // pointwise layer whose pixels can be staged 16 bytes at a time
  unsigned long long group_mask;   // conv groups this launch covers (all ones: every group)
};

__device__ __forceinline__ int a_swizzle(int row, int chunk) { return row * kBK + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <int WROWS, bool POINTWISE4>
__global__ void __launch_bounds__(256, 2) escoin_dense_mfma_kernel(DenseArgs a) {
  constexpr int BM = 64 * WROWS;
  constexpr int WCOLS = 4 / WROWS;           // waves along the pixel axis
  constexpr int WN = kBN / WCOLS;            // columns per wave: 64 or 32
  constexpr int NB = WN / 32;                // 32-column MFMA blocks per wave
  constexpr int A_CHUNKS = BM * (kBK / 4) / 256;   // 16-byte chunks of A staged per thread: 4 or 2
  __shared__ __attribute__((aligned(16))) float sA[2][BM * kBK];
  __shared__ __attribute__((aligned(16))) float sB[2][kBK][kBN + kBPad];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCOLS, wn = wave % WCOLS;
  const int cg = a.group_mask == ~0ull ? (int)blockIdx.z : nth_set_bit(a.group_mask, blockIdx.z);
  const int m0 = blockIdx.y * BM;                   // first output channel (group-local)
  const int p0 = blockIdx.x * kBN;                  // first flattened output pixel
  const int ohw = a.OH * a.OW;
  const int hw = a.H * a.W;

  // ---- A staging: chunk c = tid + 256 q -> row c / 8, 16-byte chunk c % 8 of the k-step.  The
  // weight matrix is stored zero-padded to whole k-steps and with 128 spare rows, so every load is
  // an unconditional 16-byte load (rows past Mg bring in another group's weights: those
  // accumulator rows are never stored). ----
  const float *wbase = a.w + (size_t)(cg * a.Mg + m0) * a.lda;

  // ---- B staging ----
  //  pointwise, 16 bytes: thread -> pixels 4 (tid % 32) .. + 3, k = tid / 32 + 8 q, q < 4
  //  gathered,   4 bytes: thread -> pixel tid % 128,          k = 16 (tid / 128) + q, q < 16
  const int bp = POINTWISE4 ? (tid & 31) * 4 : (tid & 127);
  const int bk = POINTWISE4 ? (tid >> 5) : 16 * (wave >> 1);   // (gathered: wave-uniform)
  // Columns past P (the last tile) are clamped to the last pixel: they are computed and never
  // stored, so that every staging load below is unconditional (a conditional load makes the
  // compiler wait for it at the join, in front of the MFMAs it is supposed to fly under).
  const int p = min(p0 + bp, a.P - (POINTWISE4 ? 4 : 1));
  const int n = p / ohw;
  const int rem = p - n * ohw;
  int ih0 = 0, iw0 = 0;
  if (!POINTWISE4) {
    const int oh = rem / a.OW, ow = rem - oh * a.OW;
    ih0 = oh * a.stride_h - a.pad_h;
    iw0 = ow * a.stride_w - a.pad_w;
  }
  // pointwise: the pixel's address in channel 0 of its group; gathered: the image's first element
  // of that channel (the window's offset is added per tap, 0 for a tap outside the image)
  const float *img = a.in + ((size_t)n * a.C + (size_t)cg * a.Cg) * hw + (POINTWISE4 ? rem : 0);
  const int win = ih0 * a.W + iw0;

  f32x16 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x16{0};

  float4 av[A_CHUNKS];
  float4 bv4[POINTWISE4 ? 4 : 1];
  float bv[POINTWISE4 ? 1 : 16];
  unsigned bmask = 0u;
  auto gather = [&](int k0) {
#pragma unroll
    for (int q = 0; q < A_CHUNKS; ++q) {
      const int c = tid + 256 * q;
      av[q] = *reinterpret_cast<const float4 *>(wbase + (size_t)(c >> 3) * a.lda + k0 + ((c & 7) << 2));
    }
    if (POINTWISE4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // k past K (last k-step): any valid address will do, A holds zeros there
        const int k = min(k0 + bk + 8 * q, a.K - 1);
        bv4[q] = *reinterpret_cast<const float4 *>(img + (size_t)k * hw);
      }
    } else {
      // the 16 taps of this wave's half of the k-step: 128 contiguous bytes at a wave-uniform
      // address -> two s_load_dwordx16, issued once, ahead of the 16 gathers
      const int4 *tp = reinterpret_cast<const int4 *>(a.ktab + k0 + bk);
      int4 tt[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) tt[q] = tp[q];
      bmask = 0u;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int toff = (q & 1) ? tt[q >> 1].z : tt[q >> 1].x;
        const int tdyx = (q & 1) ? tt[q >> 1].w : tt[q >> 1].y;
        const int ih = ih0 + (tdyx & 0xFFFF), iw = iw0 + (tdyx >> 16);
        const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        bv[q] = img[ok ? win + toff : 0];      // unconditional load; zeroed when staged
        bmask |= ok ? (1u << q) : 0u;
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < A_CHUNKS; ++q) {
      const int c = tid + 256 * q;
      *reinterpret_cast<float4 *>(&sA[buf][a_swizzle(c >> 3, c & 7)]) = av[q];
    }
    if (POINTWISE4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&sB[buf][bk + 8 * q][bp]) = bv4[q];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) sB[buf][bk + q][bp] = ((bmask >> q) & 1u) ? bv[q] : 0.f;
    }
  };

  gather(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = 0; k0 < a.K; k0 += kBK, buf ^= 1) {
    const bool more = k0 + kBK < a.K;
    if (more) gather(k0 + kBK);                     // flies under the MFMAs below
    // fragments of k-group kg + 1 are read while the MFMAs of k-group kg run
    // lane (i, h): A rows wm * 64 + {0, 32} + i, k = 8 kg + 4 h + t; B columns wn * WN + 32 j + i
    const int ra = wm * 64 + li;
    float4 fa[2][2];
    float fb[2][4][NB];
    auto read_frags = [&](int kg, int s) {
      fa[s][0] = *reinterpret_cast<const float4 *>(&sA[buf][a_swizzle(ra, 2 * kg + lh)]);
      fa[s][1] = *reinterpret_cast<const float4 *>(&sA[buf][a_swizzle(ra + 32, 2 * kg + lh)]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[s][t][j] = sB[buf][8 * kg + 4 * lh + t][wn * WN + 32 * j + li];
    };
    read_frags(0, 0);
#pragma unroll
    for (int kg = 0; kg < kBK / 8; ++kg) {
      const int s = kg & 1;
      if (kg + 1 < kBK / 8) read_frags(kg + 1, s ^ 1);
      // keep the order: next group's LDS reads first, then this group's MFMAs (left alone, the
      // scheduler sinks each read to just above its use and every 4 MFMAs wait for LDS)
      __builtin_amdgcn_sched_barrier(0);
      const float a0[4] = {fa[s][0].x, fa[s][0].y, fa[s][0].z, fa[s][0].w};
      const float a1[4] = {fa[s][1].x, fa[s][1].y, fa[s][1].z, fa[s][1].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], fb[s][t][j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], fb[s][t][j], acc[1][j], 0, 0, 0);
        }
      }
    }
    if (more) stage(buf ^ 1);                       // the other buffer: last read one step ago
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
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
        float v = acc[i][j][reg];
        if (a.bias) v += a.bias[cg * a.Mg + m];
        if (a.relu) v = fmaxf(v, 0.f);
        obase[(size_t)m * ohw] = v;
      }
    }
  }
}

const char *dense_kernel_name() { return "escoin_dense_mfma_kernel"; }

// Layout of the dense weight matrix on the device: rows of dense_lda(K) floats (whole k-steps,
// zero padded), kDenseSpareRows zero rows after the last one.
int dense_lda(int K) { return (K + kBK - 1) / kBK * kBK; }
int dense_spare_rows() { return 128; }

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
  // pointwise (is_1x1_, base_conv_layer.cpp:374-379) with whole quads of pixels per image: the
  // column matrix IS the bottom blob and is staged 16 bytes at a time
  const bool pointwise = g.d.KH == 1 && g.d.KW == 1 && g.d.stride_h == 1 && g.d.stride_w == 1 &&
                         g.d.pad_h == 0 && g.d.pad_w == 0;
  a.vec_b = pointwise && (g.d.H * g.d.W) % 4 == 0 && (reinterpret_cast<uintptr_t>(bottom) & 15) == 0;
  a.group_mask = p->use_dense ? ~0ull : p->dense_mask;
  const int bm = g.Mg <= 64 ? 64 : 128;
  dim3 grid((unsigned)((P + kBN - 1) / kBN), (unsigned)((g.Mg + bm - 1) / bm),
            (unsigned)(p->use_dense ? g.d.group : p->n_dense_groups));
  if (grid.y > 65535u || grid.z > 65535u) return fail(ESCOIN_EINVAL, "dense kernel: grid too large");
  if (bm == 64) {
    if (a.vec_b) hipLaunchKernelGGL((escoin_dense_mfma_kernel<1, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((escoin_dense_mfma_kernel<1, false>), grid, dim3(256), 0, stream, a);
  } else {
    if (a.vec_b) hipLaunchKernelGGL((escoin_dense_mfma_kernel<2, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((escoin_dense_mfma_kernel<2, false>), grid, dim3(256), 0, stream, a);
  }
  ESCOIN_HIP_TRY(hipGetLastError());
  return ESCOIN_OK;
}

}  // namespace escoin
