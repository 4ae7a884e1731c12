// escoin_capi.hip -- host side of the C ABI declared in include/escoin.h:
// plan life cycle, WeightAlign (dense -> CSR -> device weight streams), dispatch.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "escoin_plan.h"
#include "knobs.h"

namespace escoin {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

// Density above which KERNEL_AUTO sends a conv group to the dense fp32-MFMA kernel: the measured
// crossover between the tiled sparse kernel and the dense kernel (profiles/r02_crossover.md).
constexpr int kDefaultDenseThresholdPct = 50;
constexpr int kGenericDenseThresholdPct = 4;

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <typename T>
int set_csr_host(escoin_plan *p, const int *rowptr, const int *colidx, const T *values, const int *nnz_per_group);
template <typename T> std::vector<std::vector<T>> &plan_vals(escoin_plan *p);
template <> std::vector<std::vector<float>> &plan_vals<float>(escoin_plan *p) { return p->values; }
template <> std::vector<std::vector<double>> &plan_vals<double>(escoin_plan *p) { return p->values64; }

static int out_dim(int in, int k, int pad, int stride, int dil) {
  // conv_layer.cpp:16-19
  return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}

static int validate(const escoin_conv_desc *d, Geometry *g) {
  if (!d) return fail(ESCOIN_EINVAL, "null descriptor");
  if (d->N < 1 || d->C < 1 || d->H < 1 || d->W < 1 || d->M < 1 || d->KH < 1 || d->KW < 1)
    return fail(ESCOIN_EINVAL, "non-positive dimension");
  if (d->pad_h < 0 || d->pad_w < 0 || d->stride_h < 1 || d->stride_w < 1 || d->dil_h < 1 ||
      d->dil_w < 1 || d->group < 1)
    return fail(ESCOIN_EINVAL, "bad pad/stride/dilation/group");
  // base_conv_layer.cpp:393-396: channels_ % group_ == 0, num_output_ % group_ == 0
  if (d->C % d->group != 0) return fail(ESCOIN_EINVAL, "channels not divisible by group");
  if (d->M % d->group != 0) return fail(ESCOIN_EINVAL, "num_output not divisible by group");
  if (d->KH > 255 || d->KW > 255 || d->C / d->group > 32767)
    return fail(ESCOIN_EINVAL, "kernel > 255 or > 32767 channels per group not supported");
  const int oh = out_dim(d->H, d->KH, d->pad_h, d->stride_h, d->dil_h);
  const int ow = out_dim(d->W, d->KW, d->pad_w, d->stride_w, d->dil_w);
  if (oh < 1 || ow < 1) return fail(ESCOIN_EINVAL, "empty output (kernel larger than padded input)");
  if (g) {
    g->d = *d;
    g->OH = oh;
    g->OW = ow;
    g->Cg = d->C / d->group;
    g->Mg = d->M / d->group;
    g->kdim = g->Cg * d->KH * d->KW;
  }
  return ESCOIN_OK;
}

static void free_device(escoin_plan *p) {
  if (p->d_rowptr) (void)hipFree(p->d_rowptr);
  if (p->d_taps) (void)hipFree(p->d_taps);
  if (p->d_vals) (void)hipFree(p->d_vals);
  if (p->d_vals64) (void)hipFree(p->d_vals64);
  p->d_vals64 = nullptr;
  tiled_release(p);
  if (p->d_col) (void)hipFree(p->d_col);
  p->d_col = nullptr;
  p->col_bytes = 0;
  if (p->d_dense_w) (void)hipFree(p->d_dense_w);
  p->d_dense_w = nullptr;
  if (p->d_ktab) (void)hipFree(p->d_ktab);
  p->d_ktab = nullptr;
  if (p->d_sk_ws) (void)hipFree(p->d_sk_ws);
  p->d_sk_ws = nullptr;
  p->sk_ws_bytes = 0;
  p->sk_flag_words = 0;
  if (p->h_sk_fail) (void)hipHostFree(p->h_sk_fail);
  p->h_sk_fail = nullptr;
  p->d_sk_fail = nullptr;
  p->sk_used = false;
  p->d_rowptr = p->d_taps = nullptr;
  p->d_vals = nullptr;
  p->device_bytes = 0;
}

// Small launches.  The LDS-tiled kernels walk a tile block by block, every block a round trip to HBM, on as many
// workgroups as the batch has tiles x columns: 11-19 us for one image of a GoogLeNet 1x1 layer; the generic kernel
// puts a lane on every output pixel of the whole chip and needs 7-12 (profiles/r04_batch_sweep.md -- the reference's
// SCONV mode calls the layer image by image, conv_layer.cu:16-26).  Round 4 TIMED both kernels; a layer's bits then
// depended on the box's noise (two ranks could settle differently).  Now a RULE decides, from what WeightAlign knows --
// the same weights, options and batch give the same kernel in every process on the same device model:
//   pointwise layers only (the 3x3 / 5x5 layers keep generated code at every batch: the generic kernel needs
//   16-81 us where code needs 12-16), a launch of one round (tiles x columns <= workgroup slots of the chip: CUs, or
//   2 x CUs for the half-workgroup tilings) under 64 MFLOP;
//   generic  ~ max(7.0, 5.8 + 0.125 r, 6.2 + waves x (0.143 + 0.0473 r) / 1000)   us; r = nonzeros per output row (the
//              CSR row a wave walks with scalar loads: a latency chain), waves = N x M x ceil(OH OW / 64)
//   code     ~ 7.6 + (0.6 chained | 1.1 one call per block) x blocks per tile + 0.1 x MB of blobs    us
//   the kernel with the lower estimate.
// Fitted to 180 cells (profiles/r05_small_launch_fit.md; tools/small_launch_fit.py: both kernels forced, 1-32 images
// of every distinct GoogLeNet 1x1 shape): the models are within 6 % (code) / 12 % (generic) rms of the measurements
// and the rule's pick is at most 12.7 % behind the faster kernel, two cells of 180 more than 10 %.
// Evaluated from the TILING, before any code is generated or loaded (round 6): a layer the rule sends to
// the generic kernel no longer pays for code generation and a module load / unload at every WeightAlign.
// Returns 0 (not considered), 1 (generated code), 2 (generic kernel).
int small_launch_rule(const escoin_plan *p, const Tiling &t, bool chained) {
  const Geometry &g = p->g;
  if (!(p->kernel_choice == ESCOIN_KERNEL_AUTO && p->n_dense_groups == 0 && (p->tiling_batch <= 0 || p->tiling_batch == g.d.N) &&
        g.d.KH == 1 && g.d.KW == 1))
    return 0;
  long nnz = 0;
  for (const auto &c : p->colidx) nnz += (long)c.size();
  const double flops = 2.0 * g.d.N * g.OH * g.OW * (double)nnz;
  const long tiles = t.band_mode ? (long)g.d.N * t.bands : ((long)g.d.N + t.nseg - 1) / t.nseg;
  const long wgs = tiles * t.n_ocblk * g.d.group;
  const long slots = (long)tiled_device_cus() * (t.waves == 4 ? 2 : 1);
  if (!(flops < 64e6 && wgs <= slots)) return 0;
  const double r = (double)nnz / (double)g.d.M;
  const double waves = (double)g.d.N * g.d.M * std::ceil((double)g.OH * g.OW / 64.0);
  const double mb = 4.0 * g.d.N * ((double)g.d.C * g.d.H * g.d.W + (double)g.d.M * g.OH * g.OW) * 1e-6;
  const double t_gen = std::max(std::max(7.0, 5.8 + 0.125 * r), 6.2 + waves * (0.143 + 0.0473 * r) * 1e-3);
  const double t_code = 7.6 + (chained ? 0.6 : 1.1) * t.n_icb + 0.1 * mb;
  const int pick = t_gen < t_code ? 2 : 1;
  if (getenv("ESCOIN_VERBOSE"))
    fprintf(stderr, "[escoin] small launch (%.1f MFLOP, %ld workgroups on %ld slots, %d blocks): code ~%.1f us, generic ~%.1f us -> %s\n",
            flops * 1e-6, wgs, slots, t.n_icb, t_code, t_gen, pick == 2 ? "generic" : "code");
  return pick;
}

// Dtype = double: rowptr / packed taps / double values for the order-preserving generic kernel -- the only device
// kernel a double plan runs, in every conv_mode (fp64 vector FMA is native on gfx950; the LDS-tiled, generated-code and
// MFMA kernels are fp32: north_star measures fp32, double is boundary completeness, conv_layer.cu:75).
static int upload_f64(escoin_plan *p, hipStream_t stream) {
  const Geometry &g = p->g;
  long nnz = 0;
  for (int grp = 0; grp < g.d.group; ++grp) nnz += (long)p->colidx[grp].size();
  std::vector<int> rowptr(g.d.M + 1), taps((size_t)(nnz > 0 ? nnz : 1));
  std::vector<double> vals((size_t)(nnz > 0 ? nnz : 1));
  long base = 0;
  for (int grp = 0; grp < g.d.group; ++grp) {
    for (int m = 0; m < g.Mg; ++m) rowptr[grp * g.Mg + m] = (int)(base + p->rowptr[grp][m]);
    const long n_g = (long)p->colidx[grp].size();
    for (long j = 0; j < n_g; ++j) {
      const int col = p->colidx[grp][j];
      taps[base + j] = pack_tap(col / (g.d.KW * g.d.KH), (col / g.d.KW) % g.d.KH, col % g.d.KW);
      vals[base + j] = p->values64[grp][j];
    }
    base += n_g;
  }
  rowptr[g.d.M] = (int)base;
  ESCOIN_HIP_TRY(hipMalloc(&p->d_rowptr, sizeof(int) * rowptr.size()));
  ESCOIN_HIP_TRY(hipMalloc(&p->d_taps, sizeof(int) * taps.size()));
  ESCOIN_HIP_TRY(hipMalloc(&p->d_vals64, sizeof(double) * vals.size()));
  p->device_bytes += sizeof(int) * (rowptr.size() + taps.size()) + sizeof(double) * vals.size();
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_rowptr, rowptr.data(), sizeof(int) * rowptr.size(), hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_taps, taps.data(), sizeof(int) * taps.size(), hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_vals64, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipStreamSynchronize(stream));  // host vectors die at scope exit
  p->tiled = TiledConfig();
  p->n_dense_groups = 0;
  p->n_sparse_groups = g.d.group;
  p->use_dense = false;
  p->dense_mask = 0;
  p->sparse_mask = ~0ull;
  p->small_rule = 0;
  p->import_fast = false;
  p->kernel_name = generic_kernel_name_f64(g.d.fuse_relu != 0);
  p->aligned = true;
  return ESCOIN_OK;
}

// Uploads the CSR held in p->rowptr/colidx/values and builds the kernel-specific
// streams.  Shared tail of escoin_weight_align and escoin_plan_set_csr.
//   jit_blob: the generated-code section of a persisted aligned form (escoin_plan_import_aligned), tried
//   where the generator would otherwise run.
static int upload(escoin_plan *p, hipStream_t stream, const char *jit_blob = nullptr, size_t jit_blob_bytes = 0) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(ESCOIN_ENODEVICE, "no HIP device: escoin_weight_align / set_csr / import_aligned prepare the GPU path (Caffe::CPU mode has its own entry points: escoin_weight_align_cpu, escoin_forward_cpu)");
  ESCOIN_HIP_TRY(hipGetDevice(&p->device));
  free_device(p);
  if (p->is_f64) return upload_f64(p, stream);
  const Geometry &g = p->g;
  long nnz = 0;
  for (int grp = 0; grp < g.d.group; ++grp) nnz += (long)p->colidx[grp].size();
  std::vector<int> rowptr(g.d.M + 1), taps((size_t)(nnz > 0 ? nnz : 1));
  std::vector<float> vals((size_t)(nnz > 0 ? nnz : 1));
  long base = 0;
  for (int grp = 0; grp < g.d.group; ++grp) {
    for (int m = 0; m < g.Mg; ++m) rowptr[grp * g.Mg + m] = (int)(base + p->rowptr[grp][m]);
    const long n_g = (long)p->colidx[grp].size();
    for (long j = 0; j < n_g; ++j) {
      const int col = p->colidx[grp][j];
      const int kc = col % g.d.KW, kr = (col / g.d.KW) % g.d.KH, ic = col / (g.d.KW * g.d.KH);
      taps[base + j] = pack_tap(ic, kr, kc);
      vals[base + j] = p->values[grp][j];
    }
    base += n_g;
  }
  rowptr[g.d.M] = (int)base;
  ESCOIN_HIP_TRY(hipMalloc(&p->d_rowptr, sizeof(int) * rowptr.size()));
  ESCOIN_HIP_TRY(hipMalloc(&p->d_taps, sizeof(int) * taps.size()));
  ESCOIN_HIP_TRY(hipMalloc(&p->d_vals, sizeof(float) * vals.size()));
  p->device_bytes += sizeof(int) * (rowptr.size() + taps.size()) + sizeof(float) * vals.size();
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_rowptr, rowptr.data(), sizeof(int) * rowptr.size(),
                                hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_taps, taps.data(), sizeof(int) * taps.size(),
                                hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_vals, vals.data(), sizeof(float) * vals.size(),
                                hipMemcpyHostToDevice, stream));
  ESCOIN_HIP_TRY(hipStreamSynchronize(stream));  // host vectors die at scope exit

  p->tiled = TiledConfig();
  // ---- which conv groups go to the dense (fp32 MFMA) kernel -------------------------------
  //  * kernel DENSE or conv_mode LOWERED_GEMM (forward_gpu_gemm, base_conv_layer.cpp:713-746): all;
  //  * dense_gate = 1: the reference's gate -- group 0's density decides for the whole layer
  //    (nz_num_[0] / (M/g * kernel_dim) > 0.2, base_conv_layer.cpp:750, :805; quirk 5);
  //  * otherwise (AUTO): each group by its own density against the measured crossover
  //    (profiles/r02_crossover.md; option "dense_threshold_pct" overrides).
  const int G = g.d.group;
  std::vector<char> dense(G, 0);
  const double per_group = (double)g.Mg * g.kdim;
  if (p->kernel_choice == ESCOIN_KERNEL_DENSE || p->conv_mode == ESCOIN_CONV_MODE_LOWERED_GEMM) {
    dense.assign(G, 1);
  } else if (p->kernel_choice == ESCOIN_KERNEL_AUTO) {
    if (p->dense_gate) {
      if ((double)p->colidx[0].size() / per_group > 0.2) dense.assign(G, 1);
    } else {
      // a geometry the tiled kernels do not cover (stride / dilation != 1) has only the generic kernel on
      // the sparse side -- one lane per output pixel, ~2 sparse TFLOP/s -- against 65-110 dense TFLOP/s:
      // the MFMA kernel wins from ~4 % density on (ResNet-50's stride-2 1x1 layers pruned @90 %:
      // 2.2-2.7 ms generic vs 0.17-0.58 ms dense, profiles/r04_chain_prune_1x1.md)
      const bool fast_sparse = tiled_supported(g);
      const int auto_thr = fast_sparse ? kDefaultDenseThresholdPct : kGenericDenseThresholdPct;
      const double thr = (p->dense_threshold_pct >= 0 ? p->dense_threshold_pct : auto_thr) / 100.0;
      // Above the cut the dense kernel used to win by rule.  With generated code on every sparse layer it
      // does not: measured crossovers (profiles/r04_crossover.md) sit between 53 % density (AlexNet conv3, whose
      // 507 output tiles fill the 512 MFMA slots exactly) and > 90 % (res2, 64 output channels: half-empty
      // 64 x 128 tiles) -- no single density separates them, what WeightAlign knows about the two kernels does:
      //   dense  = rounds of 128 x 128 (64 x 128 when a group has <= 64 channels) output tiles on 512 slots at
      //            120 TFLOP/s (x 0.7 for the narrow tiles), no faster than the blobs at 4.0 TB/s;
      //   sparse = 50 us + 2 * pixels * nonzeros at 80 TFLOP/s (3x3 / 5x5), 20 us + ... at 68 (1x1), no faster than
      //            the blobs at 4.8 TB/s.
      // Re-fitted in round 5 (profiles/r05_crossover.md; round 4's constants -- 25 us + 68 / 58 TFLOP/s against 110 --
      // were fitted to generated code that still moved a literal per nonzero, and on round 5's code sent res2 @10 %
      // sparsity to the dense kernel at 782 us against 588, res4 / res5 @20 % at 688 / 682 against 583 / 574).
      // Worst regret over the 14 shapes x 12 sparsities of the table: 7.7 % (round 4's constants on this table: 33 %;
      // a fixed 50 % cut: 42 %).  The model only ever decides ABOVE the cut; below it the sparse kernel always won, and
      // above 92 % density the dense one (an unpruned layer is never turned into megabytes of code).  An explicit
      // dense_threshold_pct option is obeyed as given.
      auto model_says_dense = [&](long nnz_g) {
        if ((double)nnz_g > 0.92 * per_group) return true;
        const double n = (double)(p->tiling_batch > 0 ? p->tiling_batch : g.d.N);
        const double pix = n * g.OH * g.OW;
        const int tm = g.Mg <= 64 ? 64 : 128;
        const double tiles = std::ceil((double)g.Mg / tm) * std::ceil(pix / 128.0);
        const double rounds = std::ceil(tiles / 512.0);
        const double byt = 4.0 * n * ((double)g.Cg * g.d.H * g.d.W + (double)g.Mg * g.OH * g.OW);
        const double t_dense = std::max(rounds * 2.0 * tm * 128.0 * g.kdim / (120e6 * (tm == 64 ? 0.7 : 1.0) / 512.0), byt / 4.0e6);
        const double t_sparse = std::max((g.d.KH * g.d.KW > 1 ? 50.0 : 20.0) + 2.0 * pix * (double)nnz_g / (g.d.KH * g.d.KW > 1 ? 80e6 : 68e6), byt / 4.8e6 + 8.0);
        return t_dense < t_sparse;
      };
      const bool use_model = p->dense_threshold_pct < 0 && fast_sparse && jit_available() &&
                             (ESC_KNOB("DENSE_MODEL", 1) != 0);
      if (G <= 64) {
        for (int grp = 0; grp < G; ++grp) {
          const long n_g = (long)p->colidx[grp].size();
          dense[grp] = (double)n_g / per_group > thr && (!use_model || model_says_dense(n_g));
        }
      } else if ((double)nnz / (per_group * G) > thr && (!use_model || model_says_dense(nnz / G))) {
        dense.assign(G, 1);
      }
    }
  }
  p->n_dense_groups = 0;
  p->dense_mask = 0;
  for (int grp = 0; grp < G; ++grp)
    if (dense[grp]) {
      ++p->n_dense_groups;
      if (grp < 64) p->dense_mask |= 1ull << grp;
    }
  p->n_sparse_groups = G - p->n_dense_groups;
  p->use_dense = p->n_dense_groups == G;
  if (p->use_dense) p->dense_mask = ~0ull;
  p->sparse_mask = p->n_dense_groups == 0 ? ~0ull : ~p->dense_mask & (G >= 64 ? ~0ull : ((1ull << G) - 1));
  if (p->n_dense_groups > 0) {
    const size_t lda = (size_t)dense_lda(g.kdim);
    std::vector<float> dw(((size_t)g.d.M + dense_spare_rows()) * lda, 0.f);
    for (int grp = 0; grp < G; ++grp)
      for (int m = 0; m < g.Mg; ++m)
        for (int j = p->rowptr[grp][m]; j < p->rowptr[grp][m + 1]; ++j)
          dw[((size_t)grp * g.Mg + m) * lda + p->colidx[grp][j]] = p->values[grp][j];
    ESCOIN_HIP_TRY(hipMalloc(&p->d_dense_w, sizeof(float) * dw.size()));
    p->device_bytes += sizeof(float) * dw.size();
    ESCOIN_HIP_TRY(hipMemcpyAsync(p->d_dense_w, dw.data(), sizeof(float) * dw.size(),
                                  hipMemcpyHostToDevice, stream));
    ESCOIN_HIP_TRY(hipStreamSynchronize(stream));
    const int rc = dense_build_ktab(p, stream);
    if (rc != ESCOIN_OK) return rc;
  }
  if (p->use_dense) {
    p->kernel_name = dense_kernel_name();
    p->aligned = true;
    return ESCOIN_OK;
  }
  p->small_rule = 0;
  const bool explicit_tiled = p->kernel_choice == ESCOIN_KERNEL_TILED || p->kernel_choice == ESCOIN_KERNEL_JIT;
  const bool want_tiled = explicit_tiled || (p->kernel_choice == ESCOIN_KERNEL_AUTO && tiled_supported(g));
  if (want_tiled) {
    if (!tiled_supported(g))
      return fail(ESCOIN_EINVAL, "tiled kernel requested for a geometry it does not support");
    // AUTO: generated code (jit_codegen.h) wherever the sparse path runs.  Round 3 cut it off at 18 %
    // density (25 % for layers of at most 100 k nonzeros) because AlexNet's 13 x 13 layers lost 9-14 % to
    // the stream kernel at 80 % sparsity and below; that loss was the eight L2s each streaming every
    // column's code, and grouping the workgroup columns by XCD (sconv_tiled.hip, xcd_q) removed it:
    // over 50-95 % sparsity on every BASELINE 3x3 / 5x5 / 1x1 shape generated code is now ahead of the
    // stream kernel at every point (profiles/r04_crossover.md; the worst point, alex_conv2 @50 %, by 14 %).
    // ESCOIN_JIT_MAX_DENSITY_PCT restores a density cut for experiments; code beyond kMaxJitBytes
    // (sconv_tiled.hip) falls back to the stream kernel by itself.
    static const int jit_max_density_pct = (int)ESC_KNOB("JIT_MAX_DENSITY_PCT", 100);
    double dens_sparse = 0;
    long nz_sparse = 0;
    {
      long ng = 0;
      for (int grp = 0; grp < G; ++grp)
        if (!dense[grp]) { nz_sparse += (long)p->colidx[grp].size(); ++ng; }
      dens_sparse = ng ? (double)nz_sparse / (per_group * ng) : 0.0;
    }
    const bool sparse_enough = dens_sparse * 100.0 <= jit_max_density_pct;
    const bool try_jit = p->kernel_choice == ESCOIN_KERNEL_JIT ||
                         (p->kernel_choice == ESCOIN_KERNEL_AUTO && jit_available() && sparse_enough);
    int rc = ESCOIN_OK;
    p->import_fast = false;
    p->small_rule = 0;     // (set by tiled_build / tiled_import from the tiling, before any code is generated or loaded)
    if (try_jit && jit_blob && jit_blob_bytes > 0) {
      rc = tiled_import(p, jit_blob, jit_blob_bytes, stream);     // (leaves tiled.enabled false when the blob does not fit)
      if (rc != ESCOIN_OK) return rc;
      p->import_fast = p->tiled.enabled;
    }
    if (try_jit && !p->tiled.enabled && p->small_rule != 2) {
      rc = tiled_build(p, stream, true);
      if (rc != ESCOIN_OK && p->kernel_choice == ESCOIN_KERNEL_JIT) return rc;
      if (!p->tiled.enabled && p->kernel_choice == ESCOIN_KERNEL_JIT)
        return fail(ESCOIN_EINVAL, "generated-code kernel requested but the layer does not fit it");
      if (rc != ESCOIN_OK) {
        // KERNEL_AUTO: a failure of the code path (code object manager, module load, an allocation)
        // is not the layer's failure -- the stream kernel runs it.  Whatever the attempt left on the
        // device is released first, and the reason is not lost.
        if (getenv("ESCOIN_VERBOSE"))
          fprintf(stderr, "[escoin] generated code unavailable for this layer (%s): falling back to the stream kernel\n",
                  g_last_error.c_str());
        const float dens = p->tiled.density;
        tiled_release(p);
        p->tiled.density = dens;
        rc = ESCOIN_OK;
      }
    }
    if (!p->tiled.enabled && p->small_rule != 2) {
      rc = tiled_build(p, stream, false);   // leaves tiled.enabled false when the stream does not fit LDS
      if (rc != ESCOIN_OK) return rc;
    }
    if (!p->tiled.enabled && p->kernel_choice == ESCOIN_KERNEL_TILED)
      return fail(ESCOIN_EINVAL, "tiled kernel requested but its weight stream does not fit the LDS budget");
  }
  p->kernel_name = p->tiled.enabled ? tiled_kernel_name(p) : generic_kernel_name(g.d.fuse_relu != 0);
  if (p->n_dense_groups > 0) p->kernel_name += std::string(" + ") + dense_kernel_name();
  p->aligned = true;
  return ESCOIN_OK;
}


template <typename T>
void csr_from_dense(escoin_plan *p, const T *w) {
  const Geometry &g = p->g;
  // caffe_cpu_sparse_dense2csr, math_functions.cpp:92-105: row-major scan, keep != 0
  const size_t weight_offset = (size_t)g.Mg * g.kdim;  // base_conv_layer.cpp:60
  p->aligned = false;
  p->host_aligned = false;
  free_device(p);
  p->is_f64 = sizeof(T) == 8;
  std::vector<std::vector<T>> &vals = plan_vals<T>(p);
  for (int grp = 0; grp < g.d.group; ++grp) {
    std::vector<int> &rp = p->rowptr[grp];
    std::vector<int> &ci = p->colidx[grp];
    std::vector<T> &va = vals[grp];
    rp.assign(g.Mg + 1, 0);
    ci.clear();
    va.clear();
    p->values[grp].clear();
    p->values64[grp].clear();
    const T *A = w + weight_offset * grp;
    for (int i = 0; i < g.Mg; ++i) {
      for (int j = 0; j < g.kdim; ++j) {
        const T v = A[(size_t)i * g.kdim + j];
        if (v != 0) {
          va.push_back(v);
          ci.push_back(j);
        }
      }
      rp[i + 1] = (int)ci.size();
    }
  }
  p->cpu_off_valid = false;
  p->host_aligned = true;
}
template void csr_from_dense<float>(escoin_plan *, const float *);
template void csr_from_dense<double>(escoin_plan *, const double *);

template <typename T>
static int weight_align_t(escoin_plan *p, const T *dense_w, int w_on_device, void *stream) {
  if (!p || !dense_w) return fail(ESCOIN_EINVAL, "null argument");
  const auto t_start = std::chrono::steady_clock::now();
  const Geometry &g = p->g;
  const size_t count = (size_t)g.d.M * g.kdim;
  std::vector<T> host;
  const T *w = dense_w;
  if (w_on_device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
      return fail(ESCOIN_ENODEVICE, "no HIP device: cannot read device weights");
    host.resize(count);
    ESCOIN_HIP_TRY(hipMemcpyAsync(host.data(), dense_w, sizeof(T) * count, hipMemcpyDeviceToHost, (hipStream_t)stream));
    ESCOIN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    w = host.data();
  }
  csr_from_dense<T>(p, w);
  const int rc = upload(p, (hipStream_t)stream);
  p->align_ms = ms_since(t_start);
  return rc;
}

template <typename T>
static int set_csr_t(escoin_plan *p, const int *rowptr, const int *colidx, const T *values, const int *nnz_per_group,
                     void *stream) {
  const auto t_start = std::chrono::steady_clock::now();
  const int rc = set_csr_host<T>(p, rowptr, colidx, values, nnz_per_group);
  if (rc != ESCOIN_OK) return rc;
  const int rc2 = upload(p, (hipStream_t)stream);
  p->align_ms = ms_since(t_start);
  return rc2;
}

template <typename T>
static int get_csr_t(const escoin_plan *p, int *rowptr, int *colidx, T *values, int stretched) {
  if (!p || !rowptr) return fail(ESCOIN_EINVAL, "null argument");
  if (values && p->host_aligned && p->is_f64 != (sizeof(T) == 8))
    return fail(ESCOIN_ESTATE, p->is_f64 ? "get_csr: the plan holds double values (use escoin_plan_get_csr_f64)"
                                         : "get_csr_f64: the plan holds float values (use escoin_plan_get_csr)");
  const Geometry &g = p->g;
  const std::vector<std::vector<T>> &vals = plan_vals<T>(const_cast<escoin_plan *>(p));
  long base = 0;
  for (int grp = 0; grp < g.d.group; ++grp) {
    memcpy(rowptr + (size_t)grp * (g.Mg + 1), p->rowptr[grp].data(), sizeof(int) * (g.Mg + 1));
    const long n_g = (long)p->colidx[grp].size();
    for (long j = 0; j < n_g; ++j) {
      int col = p->colidx[grp][j];
      if (stretched) {  // base_conv_layer.cpp:99-105
        const int kc = col % g.d.KW, kr = (col / g.d.KW) % g.d.KH, ic = col / (g.d.KW * g.d.KH);
        col = (ic * (g.d.H + g.d.pad_h) + kr) * (g.d.W + g.d.pad_w) + kc;
      }
      if (colidx) colidx[base + j] = col;
      if (values) values[base + j] = vals[grp][j];
    }
    base += n_g;
  }
  return ESCOIN_OK;
}

}  // namespace escoin

using namespace escoin;

extern "C" {

const char *escoin_last_error(void) { return g_last_error.c_str(); }

int escoin_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int escoin_out_shape(const escoin_conv_desc *desc, int *out_h, int *out_w) {
  Geometry g;
  int rc = validate(desc, &g);
  if (rc != ESCOIN_OK) return rc;
  if (out_h) *out_h = g.OH;
  if (out_w) *out_w = g.OW;
  return ESCOIN_OK;
}

long escoin_padded_len(const escoin_conv_desc *d) {
  if (!d) return fail(ESCOIN_EINVAL, "null descriptor");
  // base_conv_layer.cpp:71, plus the pad_w floats that formula forgets when pad_h == 0 < pad_w: the last row's right
  // padding is read out of the floats that follow the row, and without a bottom padding row nothing follows the last
  // channel's last row (the reference's kernels read past its allocation there; none of its models has such a layer).
  // A buffer of this length, zeroed once, is safe for every entry point of this library.
  return (long)d->C * (d->H + d->pad_h) * (d->W + d->pad_w) + (long)d->pad_h * (d->W + 2 * d->pad_w) +
         (d->pad_h == 0 ? d->pad_w : 0);
}

int escoin_plan_create(const escoin_conv_desc *desc, escoin_plan **plan) {
  return guarded([&]() -> int {
    if (!plan) return fail(ESCOIN_EINVAL, "null plan pointer");
    *plan = nullptr;
    Geometry g;
    int rc = validate(desc, &g);
    if (rc != ESCOIN_OK) return rc;
    escoin_plan *p = new (std::nothrow) escoin_plan();
    if (!p) return fail(ESCOIN_ENOMEM, "out of host memory");
    p->g = g;
    p->rowptr.assign(g.d.group, std::vector<int>(g.Mg + 1, 0));
    p->colidx.assign(g.d.group, std::vector<int>());
    p->values.assign(g.d.group, std::vector<float>());
    p->values64.assign(g.d.group, std::vector<double>());
    *plan = p;
    return ESCOIN_OK;
  });
}

int escoin_plan_destroy(escoin_plan *plan) {
  if (!plan) return ESCOIN_OK;
  free_device(plan);
  delete plan;
  return ESCOIN_OK;
}

int escoin_plan_set_option(escoin_plan *p, const char *key, int value) {
  return guarded([&]() -> int {
    if (!p || !key) return fail(ESCOIN_EINVAL, "null argument");
    if (p->aligned && strcmp(key, "conv_mode") != 0 && strcmp(key, "cpu_channel_block") != 0 && strcmp(key, "cpu_images_per_job") != 0)
      return fail(ESCOIN_ESTATE, "this option must be set before weight_align/set_csr");
    if (!strcmp(key, "tiling_batch")) {
      if (value < 0) return fail(ESCOIN_EINVAL, "tiling_batch must be >= 0");
      p->tiling_batch = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "max_launch_bytes")) {
      if (value < 0) return fail(ESCOIN_EINVAL, "max_launch_bytes must be >= 0");
      p->max_launch_bytes = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "dense_threshold_pct")) {
      if (value < -1 || value > 100) return fail(ESCOIN_EINVAL, "dense_threshold_pct must be in [-1, 100]");
      p->dense_threshold_pct = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "cpu_images_per_job")) {
      if (value < 0) return fail(ESCOIN_EINVAL, "cpu_images_per_job must be >= 0");
      p->cpu_img_force = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "cpu_channel_block")) {
      if (value < 0) return fail(ESCOIN_EINVAL, "cpu_channel_block must be >= 0");
      p->cpu_blk_force = value;
      p->cpu_blk_cb = -1;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "code_loader")) {
      if (value < 0 || value > 1) return fail(ESCOIN_EINVAL, "code_loader must be 0 or 1");
      p->code_loader = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "stream_stores")) {
      if (value < -1 || value > 1) return fail(ESCOIN_EINVAL, "stream_stores must be -1, 0 or 1");
      p->stream_stores = value;
      return ESCOIN_OK;
    }
    if (!strcmp(key, "kernel")) {
      if (value < ESCOIN_KERNEL_AUTO || value > ESCOIN_KERNEL_JIT)
        return fail(ESCOIN_EINVAL, "unknown kernel id");
      p->kernel_choice = value;
    } else if (!strcmp(key, "conv_mode")) {
      if (value < ESCOIN_CONV_MODE_LOWERED_GEMM || value > ESCOIN_CONV_MODE_SCONV_PAR)
        return fail(ESCOIN_EINVAL, "conv_mode must be one of Caffe::ConvMode's four values (0..3)");
      const bool regroup = p->aligned && (value == ESCOIN_CONV_MODE_LOWERED_GEMM) !=
                                             (p->conv_mode == ESCOIN_CONV_MODE_LOWERED_GEMM);
      p->conv_mode = value;
      // to or from LOWERED_GEMM on an aligned plan: the dense / sparse device structures are rebuilt
      // from the CSR the plan holds (the other three modes share theirs).  The plan is not aligned
      // while that happens: if an allocation fails, the next forward reports ESCOIN_ESTATE instead of
      // launching on freed pointers.  The rebuild must happen on the device the plan lives on.
      if (regroup) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != p->device)
          return fail(ESCOIN_ESTATE, "conv_mode flip on an aligned plan: the current device is not the plan's device");
        p->aligned = false;
        return upload(p, nullptr);
      }
    } else if (!strcmp(key, "dense_gate")) {
      p->dense_gate = value != 0;
    } else {
      return fail(ESCOIN_EINVAL, std::string("unknown option: ") + key);
    }
    return ESCOIN_OK;
  });
}

int escoin_weight_align(escoin_plan *p, const float *dense_w, int w_on_device, void *stream) {
  return guarded([&]() -> int { return weight_align_t<float>(p, dense_w, w_on_device, stream); });
}

int escoin_weight_align_f64(escoin_plan *p, const double *dense_w, int w_on_device, void *stream) {
  return guarded([&]() -> int { return weight_align_t<double>(p, dense_w, w_on_device, stream); });
}

int escoin_plan_set_csr(escoin_plan *p, const int *rowptr, const int *colidx, const float *values,
                        const int *nnz_per_group, void *stream) {
  return guarded([&]() -> int { return set_csr_t<float>(p, rowptr, colidx, values, nnz_per_group, stream); });
}

int escoin_plan_set_csr_f64(escoin_plan *p, const int *rowptr, const int *colidx, const double *values,
                            const int *nnz_per_group, void *stream) {
  return guarded([&]() -> int { return set_csr_t<double>(p, rowptr, colidx, values, nnz_per_group, stream); });
}

}  // extern "C"

namespace escoin {
// Validates a CSR and copies it into the plan's host vectors (shared by set_csr and import_aligned); fixes the plan's
// Dtype like a WeightAlign does.
template <typename T>
int set_csr_host(escoin_plan *p, const int *rowptr, const int *colidx, const T *values, const int *nnz_per_group) {
  if (!p || !rowptr || !nnz_per_group) return fail(ESCOIN_EINVAL, "null argument");
  const Geometry &g = p->g;
  long base = 0;
  for (int grp = 0; grp < g.d.group; ++grp) {
    const int n_g = nnz_per_group[grp];
    const int *rp = rowptr + (size_t)grp * (g.Mg + 1);
    if (n_g < 0 || rp[0] != 0 || rp[g.Mg] != n_g)
      return fail(ESCOIN_EINVAL, "set_csr: rowptr does not match nnz_per_group");
    if (n_g > 0 && (!colidx || !values)) return fail(ESCOIN_EINVAL, "set_csr: null colidx/values");
    for (int m = 0; m < g.Mg; ++m)
      if (rp[m + 1] < rp[m]) return fail(ESCOIN_EINVAL, "set_csr: rowptr not monotone");
    for (int j = 0; j < n_g; ++j)
      if (colidx[base + j] < 0 || colidx[base + j] >= g.kdim)
        return fail(ESCOIN_EINVAL, "set_csr: column index out of range");
    // caffe_cpu_sparse_dense2csr scans a row left to right (math_functions.cpp:92-105): columns are
    // strictly ascending within a row.  The stream builder's row grouping and the reference-order
    // kernels (bit-exact summation order) rely on it, so anything else is refused, not repaired.
    for (int m = 0; m < g.Mg; ++m)
      for (int j = rp[m] + 1; j < rp[m + 1]; ++j)
        if (colidx[base + j] <= colidx[base + j - 1])
          return fail(ESCOIN_EINVAL, "set_csr: column indices must be strictly ascending within a row");
    base += n_g;
  }
  // (validated as a whole first: a refused CSR leaves the plan as it was)
  p->aligned = false;
  p->host_aligned = false;
  p->is_f64 = sizeof(T) == 8;
  std::vector<std::vector<T>> &vals = plan_vals<T>(p);
  base = 0;
  for (int grp = 0; grp < g.d.group; ++grp) {
    const int n_g = nnz_per_group[grp];
    const int *rp = rowptr + (size_t)grp * (g.Mg + 1);
    p->rowptr[grp].assign(rp, rp + g.Mg + 1);
    p->colidx[grp].assign(colidx + base, colidx + base + n_g);
    p->values[grp].clear();
    p->values64[grp].clear();
    vals[grp].assign(values + base, values + base + n_g);
    base += n_g;
  }
  p->cpu_off_valid = false;
  p->host_aligned = true;
  return ESCOIN_OK;
}
template int set_csr_host<float>(escoin_plan *, const int *, const int *, const float *, const int *);
template int set_csr_host<double>(escoin_plan *, const int *, const int *, const double *, const int *);
}  // namespace escoin

extern "C" {

// ---- the persisted aligned form -----------------------------------------------------------------
// [AlignedHdr][desc][nnz_per_group][rowptr][colidx][values][generated-code section (sconv_tiled.hip)]
namespace {
constexpr uint32_t kAlignedMagic = 0x4E474C41u;   // "ALGN"
constexpr uint32_t kAlignedVersion = 2u;          // 2: content tags over the CSR and the code section (round 6)
struct AlignedHdr {
  uint32_t magic, version;
  uint64_t total_bytes, nnz, jit_bytes;
  // Content tags: 64-bit hashes of the CSR section (descriptor included) and of the generated-code section, and a
  // third one binding the two -- the code was generated FROM this CSR.  A blob whose sections come from different
  // exports (a torn or spliced broadcast, a stale cache file patched with new weights) is refused instead of running
  // code that disagrees with its CSR.  Not a security boundary: an integrity check against accidents.
  uint64_t csr_tag, jit_tag, pair_tag;
};

// 4 x 64-bit multiply-rotate lanes over 32-byte stripes (the shape of XXH64's main loop): ~10 GB/s on a host core, so
// tagging the 54 MB of the 16 ResNet layers costs ~5 ms per side.
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t content_tag(const void *data, size_t n, uint64_t seed) {
  const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull;
  const unsigned char *p = static_cast<const unsigned char *>(data);
  uint64_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
  size_t i = 0;
  for (; i + 32 <= n; i += 32)
    for (int k = 0; k < 4; ++k) {
      uint64_t w;
      memcpy(&w, p + i + 8 * k, 8);
      v[k] = rotl64(v[k] + w * P2, 31) * P1;
    }
  uint64_t h = rotl64(v[0], 1) + rotl64(v[1], 7) + rotl64(v[2], 12) + rotl64(v[3], 18) + (uint64_t)n;
  for (; i < n; ++i) h = rotl64(h ^ (p[i] * P3), 11) * P1;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}
uint64_t pair_tag_of(uint64_t a, uint64_t b) {
  const uint64_t both[2] = {a, b};
  return content_tag(both, sizeof(both), 0x6573636F696E3236ull);
}
}  // namespace

int escoin_plan_export_aligned(const escoin_plan *p, void *buf, size_t capacity, size_t *bytes) {
  return guarded([&]() -> int {
    if (!p || !bytes) return fail(ESCOIN_EINVAL, "null argument");
    if (!p->aligned) return fail(ESCOIN_ESTATE, "export_aligned before weight_align / set_csr");
    if (p->is_f64) return fail(ESCOIN_ESTATE, "export_aligned: the aligned form is defined for float plans (a double plan has no generated code to persist; hand its CSR over with escoin_plan_get_csr_f64 / set_csr_f64)");
    const Geometry &g = p->g;
    std::vector<char> jit;
    const int rc = tiled_export(p, &jit);
    if (rc != ESCOIN_OK) return rc;
    uint64_t nnz = 0;
    for (const auto &c : p->colidx) nnz += c.size();
    const size_t need = sizeof(AlignedHdr) + sizeof(escoin_conv_desc) + 4 * (size_t)g.d.group +
                        4 * (size_t)g.d.group * (g.Mg + 1) + 8 * (size_t)nnz + jit.size();
    *bytes = need;
    if (!buf) return ESCOIN_OK;                       // size query
    if (capacity < need) return fail(ESCOIN_EINVAL, "export_aligned: buffer too small");
    char *q = static_cast<char *>(buf);
    AlignedHdr h{kAlignedMagic, kAlignedVersion, (uint64_t)need, nnz, (uint64_t)jit.size(), 0, 0, 0};
    char *const hdr_at = q;
    q += sizeof(h);
    const char *const csr_at = q;
    memcpy(q, &g.d, sizeof(g.d)); q += sizeof(g.d);
    for (int grp = 0; grp < g.d.group; ++grp) { const int n = (int)p->colidx[grp].size(); memcpy(q, &n, 4); q += 4; }
    for (int grp = 0; grp < g.d.group; ++grp) { memcpy(q, p->rowptr[grp].data(), 4 * (size_t)(g.Mg + 1)); q += 4 * (size_t)(g.Mg + 1); }
    for (int grp = 0; grp < g.d.group; ++grp) { memcpy(q, p->colidx[grp].data(), 4 * p->colidx[grp].size()); q += 4 * p->colidx[grp].size(); }
    for (int grp = 0; grp < g.d.group; ++grp) { memcpy(q, p->values[grp].data(), 4 * p->values[grp].size()); q += 4 * p->values[grp].size(); }
    h.csr_tag = content_tag(csr_at, (size_t)(q - csr_at), 1);
    if (!jit.empty()) memcpy(q, jit.data(), jit.size());
    h.jit_tag = content_tag(q, jit.size(), 2);
    h.pair_tag = pair_tag_of(h.csr_tag, h.jit_tag);
    memcpy(hdr_at, &h, sizeof(h));
    return ESCOIN_OK;
  });
}

int escoin_plan_import_aligned(escoin_plan *p, const void *buf, size_t bytes, void *stream) {
  return guarded([&]() -> int {
    if (!p || !buf) return fail(ESCOIN_EINVAL, "null argument");
    const auto t_start = std::chrono::steady_clock::now();
    const Geometry &g = p->g;
    AlignedHdr h;
    if (bytes < sizeof(h) + sizeof(escoin_conv_desc)) return fail(ESCOIN_EINVAL, "import_aligned: truncated blob");
    const char *q = static_cast<const char *>(buf);
    memcpy(&h, q, sizeof(h)); q += sizeof(h);
    if (h.magic != kAlignedMagic || h.version != kAlignedVersion || h.total_bytes != bytes)
      return fail(ESCOIN_EINVAL, "import_aligned: not an aligned-form blob of this library build");
    escoin_conv_desc d;
    memcpy(&d, q, sizeof(d)); q += sizeof(d);
    // the weights' own geometry must match; batch, bias and ReLU are the importing plan's business
    if (d.C != g.d.C || d.M != g.d.M || d.KH != g.d.KH || d.KW != g.d.KW || d.group != g.d.group)
      return fail(ESCOIN_EINVAL, "import_aligned: the blob was exported for other weights (C / M / kernel / group differ)");
    // (bounded before it sizes anything: a layer has at most group * Mg * kdim weights)
    if (h.nnz > (uint64_t)g.d.group * (uint64_t)g.Mg * (uint64_t)g.kdim || h.jit_bytes > bytes)
      return fail(ESCOIN_EINVAL, "import_aligned: nnz or code section larger than the layer / the blob");
    const size_t csr_bytes = 4 * (size_t)g.d.group + 4 * (size_t)g.d.group * (g.Mg + 1) + 8 * (size_t)h.nnz;
    if (sizeof(h) + sizeof(d) + csr_bytes + h.jit_bytes != bytes) return fail(ESCOIN_EINVAL, "import_aligned: section sizes do not add up");
    {
      // the content tags, before a single byte of either section is trusted
      const char *csr_at = static_cast<const char *>(buf) + sizeof(h);
      const size_t csr_sec = sizeof(d) + csr_bytes;
      const uint64_t ct = content_tag(csr_at, csr_sec, 1), jt = content_tag(csr_at + csr_sec, (size_t)h.jit_bytes, 2);
      if (ct != h.csr_tag || jt != h.jit_tag || pair_tag_of(ct, jt) != h.pair_tag)
        return fail(ESCOIN_EINVAL, "import_aligned: content tag mismatch -- the blob is torn, or its code section does not belong to its CSR section");
    }
    const double ms_tags = ms_since(t_start);
    std::vector<int> ng(g.d.group), rp((size_t)g.d.group * (g.Mg + 1)), ci((size_t)h.nnz);
    std::vector<float> va((size_t)h.nnz);
    memcpy(ng.data(), q, 4 * ng.size()); q += 4 * ng.size();
    memcpy(rp.data(), q, 4 * rp.size()); q += 4 * rp.size();
    memcpy(ci.data(), q, 4 * ci.size()); q += 4 * ci.size();
    memcpy(va.data(), q, 4 * va.size()); q += 4 * va.size();
    uint64_t sum = 0;
    for (int n : ng) sum += (uint64_t)std::max(0, n);
    if (sum != h.nnz) return fail(ESCOIN_EINVAL, "import_aligned: nnz_per_group does not match the blob's nnz");
    const int rc = set_csr_host<float>(p, rp.data(), ci.data(), va.data(), ng.data());
    if (rc != ESCOIN_OK) return rc;
    p->aligned = false;
    // the code section only counts for the geometry it was generated for (the LDS offsets in the code
    // are this H x W's) and for the same epilogue flags
    const bool same_geom = d.H == g.d.H && d.W == g.d.W && d.pad_h == g.d.pad_h && d.pad_w == g.d.pad_w &&
                           d.stride_h == g.d.stride_h && d.stride_w == g.d.stride_w && d.dil_h == g.d.dil_h &&
                           d.dil_w == g.d.dil_w && d.N == g.d.N;
    const double ms_csr = ms_since(t_start);
    const int rc2 = upload(p, (hipStream_t)stream, same_geom && h.jit_bytes ? q : nullptr, same_geom ? (size_t)h.jit_bytes : 0);
    p->align_ms = ms_since(t_start);
    if (getenv("ESCOIN_VERBOSE"))
      fprintf(stderr, "[escoin] import_aligned: %zu bytes (code %llu): tags %.2f ms, CSR checks %.2f ms, upload + code load %.2f ms\n",
              bytes, (unsigned long long)h.jit_bytes, ms_tags, ms_csr - ms_tags, p->align_ms - ms_csr);
    return rc2;
  });
}

// The same from a DEVICE buffer (where an RCCL broadcast leaves the blob): one copy into a host staging area that the
// calling thread keeps between calls (grow-only; pageable -- pinning 13 MB costs more than the copy saves), then the
// host import: the CSR and the code object are parsed and loaded from host memory either way (hipModuleLoadData takes a
// host image).
namespace {
thread_local std::vector<char> g_stage;
}  // namespace

int escoin_plan_import_aligned_dev(escoin_plan *p, const void *dev_buf, size_t bytes, void *stream) {
  return guarded([&]() -> int {
    if (!p || !dev_buf) return fail(ESCOIN_EINVAL, "null argument");
    if (bytes < sizeof(AlignedHdr) || bytes > (size_t)1 << 36) return fail(ESCOIN_EINVAL, "import_aligned: implausible blob size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ESCOIN_ENODEVICE, "no HIP device");
    if (g_stage.size() < bytes) g_stage.resize(bytes + (bytes >> 2));     // (grows by a quarter: layers come in rising sizes)
    ESCOIN_HIP_TRY(hipMemcpyAsync(g_stage.data(), dev_buf, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    ESCOIN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return escoin_plan_import_aligned(p, g_stage.data(), bytes, stream);
  });
}

long escoin_plan_stat(const escoin_plan *p, const char *key) {
  if (!p || !key) return fail(ESCOIN_EINVAL, "null argument");
  if (!strcmp(key, "align_us")) return (long)(p->align_ms * 1e3);
  if (!strcmp(key, "code_bytes")) return (long)(p->tiled.enabled && p->tiled.jit ? p->jit_module.code_bytes : 0);
  if (!strcmp(key, "device_bytes")) return (long)p->device_bytes;
  if (!strcmp(key, "import_fast")) return p->import_fast ? 1 : 0;
  if (!strcmp(key, "cpu_images_per_job")) return p->cpu_img_last;   // images per job of the last escoin_forward_cpu
  if (!strcmp(key, "cpu_channel_block")) return p->cpu_blk_cb;     // channels per block of the last escoin_forward_cpu (0: unblocked, -1: none yet)
  if (!strcmp(key, "code_direct")) return p->jit_module.direct ? 1 : 0;     // the plan's code sits in executable memory the library filled itself
  if (!strcmp(key, "small_launch_rule")) return p->small_rule;
  if (!strcmp(key, "jit_rows")) return p->tiled.jit ? p->tiled.jit_rows : 0;
  if (!strcmp(key, "jit_records")) return p->tiled.jit ? p->tiled.jit_records : 0;
  // balance of the channel deal, x 1000 (1000 = every wave of every block costs the same; generated-code plans
  // built from weights -- 0 for an imported code object, which does not carry the figure)
  if (!strcmp(key, "deal_slowest_over_mean_x1000")) return p->tiled.jit ? (long)(p->tiled.deal_slowest_over_mean * 1000.f + 0.5f) : 0;
  if (!strcmp(key, "deal_worst_block_x1000")) return p->tiled.jit ? (long)(p->tiled.deal_worst_block * 1000.f + 0.5f) : 0;
  if (!strcmp(key, "lds_bytes")) return p->tiled.enabled ? (long)p->tiled.lds_bytes : 0;
  if (!strcmp(key, "workgroup_columns")) return p->tiled.enabled ? p->tiled.tiling.n_ocblk : 0;
  if (!strcmp(key, "streamk_gave_up")) {
    // dense kernel, stream-K launches: 1 if a workgroup's bounded wait for another one's partial sums ran out in
    // the last launch (its results are then wrong); synchronises with the device.  0 for plans that never split K.
    if (!p->d_sk_ws || p->sk_flag_words < 1 || !p->sk_used) return 0;
    if (p->h_sk_fail && *(volatile unsigned *)p->h_sk_fail != 0u) return 1;      // (sticky: any launch since WeightAlign)
    unsigned v = 0;
    if (hipMemcpy(&v, static_cast<const unsigned *>(p->d_sk_ws) + (p->sk_flag_words - 1), 4, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(ESCOIN_EHIP, "streamk_gave_up: device read failed");
    return (long)v;
  }
  if (!strcmp(key, "streamk")) return p->sk_used ? 1 : 0;
  if (!strcmp(key, "is_f64")) return p->is_f64 ? 1 : 0;
  if (!strcmp(key, "host_aligned")) return p->host_aligned ? 1 : 0;
  if (!strcmp(key, "kernel_choice")) {
    if (!p->aligned) return fail(ESCOIN_ESTATE, "kernel_choice before weight_align / set_csr");
    if (p->is_f64) return ESCOIN_KERNEL_GENERIC;
    if (p->use_dense) return ESCOIN_KERNEL_DENSE;
    if (!p->tiled.enabled) return ESCOIN_KERNEL_GENERIC;
    return p->tiled.jit ? ESCOIN_KERNEL_JIT : ESCOIN_KERNEL_TILED;
  }
  return fail(ESCOIN_EINVAL, std::string("unknown stat: ") + key);
}

long escoin_plan_nnz(const escoin_plan *p, int group) {
  if (!p) return fail(ESCOIN_EINVAL, "null plan");
  if (group >= p->g.d.group) return fail(ESCOIN_EINVAL, "group out of range");
  if (group >= 0) return (long)p->colidx[group].size();
  long n = 0;
  for (const auto &c : p->colidx) n += (long)c.size();
  return n;
}

int escoin_plan_get_csr(const escoin_plan *p, int *rowptr, int *colidx, float *values, int stretched) {
  return guarded([&]() -> int { return get_csr_t<float>(p, rowptr, colidx, values, stretched); });
}

int escoin_plan_get_csr_f64(const escoin_plan *p, int *rowptr, int *colidx, double *values, int stretched) {
  return guarded([&]() -> int { return get_csr_t<double>(p, rowptr, colidx, values, stretched); });
}

size_t escoin_plan_workspace_bytes(const escoin_plan *p) { return p ? p->device_bytes : 0; }

const char *escoin_plan_kernel_name(const escoin_plan *p) {
  if (p && p->aligned && !p->is_f64 && p->conv_mode == ESCOIN_CONV_MODE_LOWERED_SPARSE &&
      p->kernel_choice != ESCOIN_KERNEL_DENSE)
    return lowered_kernel_name();
  return p ? p->kernel_name.c_str() : "";
}

const char *escoin_plan_tiling_info(const escoin_plan *p) {
  return (p && p->aligned && p->tiled.enabled) ? p->tiled.info.c_str() : "";
}

int escoin_forward(escoin_plan *p, const float *bottom_dev, const float *bias_dev, float *top_dev,
                   int n_images, void *stream) {
  return guarded([&]() -> int {
    if (!p || !bottom_dev || !top_dev) return fail(ESCOIN_EINVAL, "null argument");
    if (!p->aligned) return fail(ESCOIN_ESTATE, "forward called before weight_align / set_csr");
    if (p->is_f64) return fail(ESCOIN_ESTATE, "forward: the plan holds double weights (use escoin_forward_f64)");
    if (n_images < 0 || n_images > p->g.d.N)
      return fail(ESCOIN_EINVAL, "n_images outside [0, desc.N]");
    if (n_images == 0) return ESCOIN_OK;
    {
      int dev = -1;
      if (hipGetDevice(&dev) != hipSuccess || dev != p->device)
        return fail(ESCOIN_ESTATE, "forward: the current device is not the device the plan was aligned on");
    }
    // dense kernel, stream-K: a fix-up wait that ran out in an EARLIER launch of this plan left wrong results in that
    // launch's top blob.  The word lives in pinned host memory (no synchronisation here) and stays set until the
    // next WeightAlign: the caller hears about it at the next call at the latest (dense_mfma.hip).
    if (p->h_sk_fail && *(volatile unsigned *)p->h_sk_fail != 0u)
      return fail(ESCOIN_EHIP, "dense kernel (stream-K): a workgroup gave up waiting for another one's partial sums in an "
                               "earlier launch of this plan; that launch's results are wrong");
    hipStream_t s = (hipStream_t)stream;
    // LOWERED_SPARSE lowers every group, whatever AUTO decided for the direct path
    if (p->conv_mode == ESCOIN_CONV_MODE_LOWERED_SPARSE && p->kernel_choice != ESCOIN_KERNEL_DENSE)
      return launch_lowered(p, bottom_dev, bias_dev, top_dev, n_images, s);
    if (p->n_dense_groups > 0) {
      const int rc = launch_dense(p, bottom_dev, bias_dev, top_dev, n_images, s);
      if (rc != ESCOIN_OK || p->use_dense) return rc;
    }
    if (p->tiled.enabled) return launch_tiled(p, bottom_dev, bias_dev, top_dev, n_images, s);
    return launch_generic(p, bottom_dev, bias_dev, top_dev, n_images, s);
  });
}

// Forward_gpu for Dtype = double (conv_layer.cu:75 instantiates the layer for both types): the order-preserving
// generic kernel in fp64, whatever the conv_mode.
int escoin_forward_f64(escoin_plan *p, const double *bottom_dev, const double *bias_dev, double *top_dev, int n_images,
                       void *stream) {
  return guarded([&]() -> int {
    if (!p || !bottom_dev || !top_dev) return fail(ESCOIN_EINVAL, "null argument");
    if (!p->aligned) return fail(ESCOIN_ESTATE, "forward called before weight_align / set_csr");
    if (!p->is_f64) return fail(ESCOIN_ESTATE, "forward_f64: the plan holds float weights (use escoin_forward)");
    if (n_images < 0 || n_images > p->g.d.N) return fail(ESCOIN_EINVAL, "n_images outside [0, desc.N]");
    if (n_images == 0) return ESCOIN_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != p->device)
      return fail(ESCOIN_ESTATE, "forward: the current device is not the device the plan was aligned on");
    return launch_generic_f64(p, bottom_dev, bias_dev, top_dev, n_images, (hipStream_t)stream);
  });
}

int escoin_gpu_sparse_csrmm(int M, int N, int K, int nnz, float alpha, const float *values,
                            const int *rowptr, const int *colidx, const float *B, float beta, float *C,
                            void *stream) {
  if (M < 0 || N < 0 || K < 0 || nnz < 0) return fail(ESCOIN_EINVAL, "negative dimension");
  if (M == 0 || N == 0) return ESCOIN_OK;
  if (!rowptr || !C || (nnz > 0 && (!values || !colidx || !B))) return fail(ESCOIN_EINVAL, "null argument");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(ESCOIN_ENODEVICE, "no HIP device (this entry point is the GPU one)");
  return csrmm(M, N, K, alpha, values, rowptr, colidx, B, beta, C, (hipStream_t)stream);
}

int escoin_gpu_sparse_csrmm_f64(int M, int N, int K, int nnz, double alpha, const double *values, const int *rowptr,
                                const int *colidx, const double *B, double beta, double *C, void *stream) {
  if (M < 0 || N < 0 || K < 0 || nnz < 0) return fail(ESCOIN_EINVAL, "negative dimension");
  if (M == 0 || N == 0) return ESCOIN_OK;
  if (!rowptr || !C || (nnz > 0 && (!values || !colidx || !B))) return fail(ESCOIN_EINVAL, "null argument");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(ESCOIN_ENODEVICE, "no HIP device (this entry point is the GPU one)");
  return csrmm_f64(M, N, K, alpha, values, rowptr, colidx, B, beta, C, (hipStream_t)stream);
}

}  // extern "C"
