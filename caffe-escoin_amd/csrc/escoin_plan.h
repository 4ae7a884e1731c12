// Internal plan structure of libescoin_hip.so (not part of the C ABI).
#ifndef ESCOIN_PLAN_H_
#define ESCOIN_PLAN_H_

#include <hip/hip_runtime.h>

#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "escoin.h"
#include "jit_module.h"
#include "stream_builder.h"

namespace escoin {

// Thread-local error message behind escoin_last_error().
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define ESCOIN_HIP_TRY(expr)                                                          \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return ::escoin::fail(ESCOIN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// No exception may cross the C ABI (std::bad_alloc from the host vectors, std::system_error from a thread team):
// every entry point that allocates runs its body through this.
template <typename F>
inline int guarded(F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(ESCOIN_ENOMEM, "out of host memory");
  } catch (const std::exception &e) {
    return fail(ESCOIN_EINVAL, std::string("internal error: ") + e.what());
  } catch (...) {
    return fail(ESCOIN_EINVAL, "internal error");
  }
}

// A nonzero's kernel tap packed for the generic kernel: ic << 16 | kr << 8 | kc
// (ic group-local).
inline int pack_tap(int ic, int kr, int kc) { return (ic << 16) | (kr << 8) | kc; }

struct Geometry {
  escoin_conv_desc d;
  int OH, OW;
  int Cg, Mg;   // channels per group
  int kdim;     // kernel_dim_ = Cg*KH*KW
};

// Parameters of the tiled kernel chosen in weight_align (see sconv_tiled.hip).
struct TiledConfig {
  bool enabled = false;
  int s4 = 0;          // quads (4 floats) per LDS row, power of two
  int kw_classes = 0;  // KW
  int tile_rows = 0;   // output rows (over the flattened (image,row) axis) per workgroup
  int imgs_per_wg = 0; // whole images per workgroup (0: row bands of one image)
  int pix_waves = 0;   // waves along the pixel axis
  int oc_waves = 0;    // waves along the output-channel axis
  int G = 0;           // output channels per wave
  int icb = 0;         // input channels per LDS block
  int n_icb = 0;       // blocks per group
  int slots = 0;       // records per row group in the weight stream
  size_t lds_bytes = 0;
  int lds_budget = 0;  // plane-buffer budget the tiling was chosen with
  float density = 0.f; // nonzero fraction of the weights the tiling was chosen with
  Tiling tiling;       // the tiling itself (chosen once in WeightAlign; launches only read it)
  int stage_bytes = 0; // LDS bytes of one wave's weight-stream staging area
  int nbuf = 2;        // plane / staging buffers per workgroup
  bool jit = false;    // the walk is generated code (jit_codegen.h) instead of the LDS-staged stream
  long jit_rows = 0, jit_records = 0;
  bool jit_dma = false;      // ... and its units stage the next block's planes themselves (jit_codegen.h DmaPlan)
  int dma_period = 0;        // quads covered by the quad table (lcm of the plane size and 64)
  int jit_pref = 0;          // code touches at the start of every unit
  bool jit_chain = false;    // a tile's units run as one chain (jit_codegen.h ChainPlan)
  bool jit_self_zero = false;  // block 0's unit initialises the accumulators (jit_codegen.h Options::self_zero)
  float deal_slowest_over_mean = 1.f, deal_worst_block = 1.f;   // balance of the channel deal (jit_codegen.h Program)
  std::string info;          // escoin_plan_tiling_info
};

// One host thread's buffers of the CPU mode (sconv_cpu.cpp): the shared-halo padded image and the store scratch.
struct CpuWorkspace {
  std::vector<char> pad, scratch, partial;   // padded image, one tile for the stores, parked sums of channel blocking
  const void *src = nullptr;     // the bottom image the padded buffer currently holds ...
  unsigned long call = 0;        // ... as of this escoin_forward_cpu call
};

}  // namespace escoin

struct escoin_plan {
  escoin::Geometry g;
  int kernel_choice = ESCOIN_KERNEL_AUTO;
  int conv_mode = ESCOIN_CONV_MODE_SCONV_PAR;
  int dense_gate = 0;
  long max_launch_bytes = 0;   // option "max_launch_bytes": bottom-blob bytes one tiled launch may cover (0: the 4 GiB descriptor range)
  int tiling_batch = 0;   // option "tiling_batch": choose the tiled kernel's tiling as for this batch (0: desc.N)
  bool aligned = false;
  int device = -1;

  // host CSR, per group (unstretched column indices), exactly what
  // caffe_cpu_sparse_dense2csr produces (math_functions.cpp:92-105)
  std::vector<std::vector<int>> rowptr;   // [group][Mg+1]
  std::vector<std::vector<int>> colidx;   // [group][nnz_g]
  std::vector<std::vector<float>> values; // [group][nnz_g]   (Dtype = float plans)
  // Dtype = double (conv_layer.cpp:102 INSTANTIATE_CLASS): the align call fixes a plan's type; a double plan keeps its
  // values here, runs the order-preserving generic kernel on the device (no fast path) and the same host kernel
  std::vector<std::vector<double>> values64;
  bool is_f64 = false;
  double *d_vals64 = nullptr;
  // Caffe::CPU mode (sconv_cpu.cpp): true once a CSR is on the host, whether or not a device was there to upload to;
  // cpu_off = the nonzeros' offsets into the shared-halo padded image for THIS geometry (dilation folded in), built on
  // the first escoin_forward_cpu after an align
  bool host_aligned = false;
  std::vector<std::vector<int>> cpu_off;
  bool cpu_off_valid = false;
  // channel blocking of the CPU mode (sconv_cpu.h GroupJob::blk_ptr): per conv group, Mg x (cpu_blk_n + 1) nonzero indices
  // for blocks of cpu_blk_cb input channels (-1: not built; 0: this geometry runs unblocked)
  std::vector<std::vector<int>> cpu_blk;
  int cpu_blk_cb = -1, cpu_blk_n = 0, cpu_blk_isa = 0, cpu_blk_elem = 0;
  int cpu_img_force = 0;          // option "cpu_images_per_job": 0 = chosen from the geometry, n = at most n images per job
  int cpu_img_last = 0;           // what the last escoin_forward_cpu used
  int cpu_blk_force = 0;          // option "cpu_channel_block": 0 = chosen from the geometry, > 0 = this many channels per block
  std::vector<escoin::CpuWorkspace> cpu_ws;   // per team thread: padded image + store scratch
  unsigned long cpu_calls = 0;

  // device arrays for the generic kernel
  int *d_rowptr = nullptr;   // [M+1] absolute offsets into d_taps/d_vals
  int *d_taps = nullptr;     // [nnz] packed (ic,kr,kc)
  float *d_vals = nullptr;   // [nnz]

  // device arrays for the tiled kernel
  escoin::TiledConfig tiled;
  unsigned *d_stream = nullptr;   // unit bodies of the weight stream (stream_builder.h)
  unsigned *d_unit_hdr = nullptr; // 8 dwords per (conv group, oc group, ic block); generated code: 1 (code offset)
  escoin::JitModule jit_module;   // generated-code kernel: where the plan's code lives on the device (jit_module.h)
  unsigned *d_chan = nullptr;     // slot -> output channel (WeightStream::chan)
  size_t stream_words = 0;
  size_t tiled_device_bytes = 0;  // device bytes of the five members above (part of device_bytes)
  // host copies of what a generated-code plan loaded, kept for escoin_plan_export_aligned: the code
  // (position-independent words, jit_codegen.h), the unit table and the channel deal
  std::vector<uint32_t> jit_code;
  int code_loader = 0;            // option "code_loader": 0 executable device memory first, 1 the code object loader (jit_module.h)
  std::vector<uint32_t> h_unit_off, h_chan;
  double align_ms = 0.0;          // wall time of the last weight_align / set_csr / import_aligned
  bool import_fast = false;       // the last import_aligned loaded a persisted code object as it was
  int small_rule = 0;              // KERNEL_AUTO's small-launch rule applied to this plan: 0 not considered, 1 kept generated code, 2 took the generic kernel

  // dense fallback (fp32 MFMA implicit GEMM), chosen per conv group: bit g of dense_mask = group g
  // goes to the MFMA kernel, bit g of sparse_mask = to the sparse kernels (layers with more than 64
  // groups take one decision for all of them: both masks are then all-ones / zero)
  float *d_dense_w = nullptr;     // [M + dense_spare_rows()][dense_lda(Cg*KH*KW)], zero padded
  int *d_ktab = nullptr;          // im2col decode per k: {image offset, dy | dx << 16} (dense_mfma.hip)
  bool use_dense = false;         // every group dense
  unsigned long long dense_mask = 0, sparse_mask = ~0ull;
  int n_dense_groups = 0, n_sparse_groups = 0;
  int dense_threshold_pct = -1;   // option "dense_threshold_pct" (-1: the measured default)
  int stream_stores = -1;         // option "stream_stores": pointwise layers write the top blob with non-temporal stores (1), never (0), by size (-1)

  // stream-K workspace of the dense kernel (dense_mfma.hip): flag words, then the partial accumulators; grown on
  // allocated at WeightAlign for layers whose launches may split K (dense_build_ktab): a launch never allocates
  mutable void *d_sk_ws = nullptr;
  mutable size_t sk_ws_bytes = 0;
  mutable int sk_flag_words = 0;
  mutable bool sk_used = false;            // the last dense launch of this plan split K (escoin_plan_stat "streamk")
  unsigned *d_sk_fail = nullptr;           // device address of h_sk_fail (looked up once, at WeightAlign)
  mutable unsigned *h_sk_fail = nullptr;   // pinned host word the kernel sets when a fix-up wait gave up (sticky until the next WeightAlign)

  // LOWERED_SPARSE comparator (sconv_lowered.hip): column buffer, grown on demand
  float *d_col = nullptr;
  size_t col_bytes = 0;

  size_t device_bytes = 0;
  std::string kernel_name = "(not aligned)";
};

namespace escoin {

// escoin_capi.hip: caffe_cpu_sparse_dense2csr over every conv group of a dense blobs_[0] (math_functions.cpp:92-105,
// base_conv_layer.cpp:55-66) into the plan's host CSR; fixes the plan's Dtype, releases what an earlier align left on
// the device and leaves the plan host_aligned (and not device-aligned).  T = float | double.
template <typename T> void csr_from_dense(escoin_plan *p, const T *w);

// sconv_generic.hip
int launch_generic(const escoin_plan *p, const float *bottom, const float *bias, float *top,
                   int n_images, hipStream_t stream);
int launch_generic_f64(const escoin_plan *p, const double *bottom, const double *bias, double *top,
                       int n_images, hipStream_t stream);
const char *generic_kernel_name(bool relu);
const char *generic_kernel_name_f64(bool relu);

// sconv_tiled.hip
bool tiled_supported(const Geometry &g);
int tiled_device_cus();             // compute units of the current device
int tiled_build(escoin_plan *p, hipStream_t stream, bool jit);  // fills p->tiled, uploads streams / loads generated code
void tiled_release(escoin_plan *p);   // frees what tiled_build put on the device (and its share of device_bytes)
// The fast half of escoin_plan_import_aligned: a generated-code plan restored from what
// escoin_plan_export_aligned wrote (tiling, channel deal, unit table, code object) -- no channel
// deal, no generator pass, no assembler.  `blob` points behind the CSR section.
int tiled_export(const escoin_plan *p, std::vector<char> *out);
int tiled_import(escoin_plan *p, const char *blob, size_t bytes, hipStream_t stream);
int launch_tiled(const escoin_plan *p, const float *bottom, const float *bias, float *top,
                 int n_images, hipStream_t stream);
const char *tiled_kernel_name(const escoin_plan *p);

// escoin_capi.hip: KERNEL_AUTO's rule for pointwise launches of one round of workgroups under 64 MFLOP, evaluated from
// the tiling before any code is generated or loaded: 0 not considered, 1 generated code, 2 the generic kernel.
int small_launch_rule(const escoin_plan *p, const Tiling &t, bool chained);

// sconv_lowered.hip (conv_mode LOWERED_SPARSE: im2col + CSR x dense, the lowering baseline)
int launch_lowered(escoin_plan *p, const float *bottom, const float *bias, float *top, int n_images,
                   hipStream_t stream);
int csrmm(int M, int N, int K, float alpha, const float *vals, const int *rowptr, const int *colidx,
          const float *B, float beta, float *C, hipStream_t stream);
int csrmm_f64(int M, int N, int K, double alpha, const double *vals, const int *rowptr, const int *colidx,
              const double *B, double beta, double *C, hipStream_t stream);
const char *lowered_kernel_name();

// dense_mfma.hip
int launch_dense(const escoin_plan *p, const float *bottom, const float *bias, float *top,
                 int n_images, hipStream_t stream);
const char *dense_kernel_name();
int dense_build_ktab(escoin_plan *p, hipStream_t stream);
int dense_lda(int K);          // floats per row of d_dense_w (K rounded up to whole k-steps)
int dense_spare_rows();       // zero rows after row M - 1

// Index of the k-th set bit of `mask` (k < popcount): blockIdx -> conv group when only some groups
// of a layer run in a launch.  Wave-uniform scalar code on the device.
__host__ __device__ inline int nth_set_bit(unsigned long long mask, int k) {
  for (; k > 0; --k) mask &= mask - 1;
  int i = 0;
  while (i < 63 && !((mask >> i) & 1ull)) ++i;
  return i;
}

}  // namespace escoin

#endif  // ESCOIN_PLAN_H_
