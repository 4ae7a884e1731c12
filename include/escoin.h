/*
 * include/escoin.h -- C ABI of libescoin_hip.so
 *
 * MI355X (gfx950) direct sparse convolution forward: the drop-in for the path
 *   ConvolutionLayer<Dtype>::Forward_gpu            src/caffe/layers/conv_layer.cu:8-40
 *     -> BaseConvolutionLayer::forward_gpu_sconv[_par]   base_conv_layer.cpp:748-848
 *        -> caffe_gpu_sconv + forward_gpu_bias           math_functions.cu:590-704,
 *                                                        base_conv_layer.cpp:850-856
 *   fed by BaseConvolutionLayer::WeightAlign             base_conv_layer.cpp:46-273
 * of chenxuhao/caffe-escoin (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / STL types cross this boundary;
 *   - every entry point returns 0 on success and a negative ESCOIN_E* code on
 *     failure; escoin_last_error() returns the message of the calling thread's last
 *     failure.  The reference aborts on every error (glog CHECK / CUDA_CHECK,
 *     include/caffe/util/device_alternate.hpp:51-78); the C++ Layer shim
 *     (caffe-escoin_amd/caffe_shim/) turns a non-zero return into that abort;
 *   - all tensors fp32 NCHW contiguous, all indices int32, as in the reference;
 *   - device pointers are HIP device pointers on the current device; `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream, which is what every
 *     reference kernel launch uses);
 *   - the caller owns bottom / top / bias / dense-weight memory; a plan owns its CSR
 *     arrays, the blocked weight streams and its scratch, and frees them in
 *     escoin_plan_destroy (reference: layer dtor, base_conv_layer.cpp:16-42);
 *   - thread-compatible: one plan per (host thread, device), like a Caffe layer
 *     instance (common.cpp:13-19 keeps the Caffe singleton thread-local).
 *   - no SILENT CPU fallback: the GPU entry points (escoin_weight_align, escoin_forward, the escoin_gpu_* helpers)
 *     fail with ESCOIN_ENODEVICE when no HIP device is visible, they never compute on the host.  Caffe::CPU mode
 *     (ConvolutionLayer::Forward_cpu, conv_layer.cpp:25-63) is a separate, explicit set of entry points --
 *     escoin_weight_align_cpu / escoin_forward_cpu / escoin_cpu_* below -- implemented in this library
 *     (csrc/sconv_cpu*.cpp) and usable on a machine without a GPU;
 *   - Dtype: the reference instantiates the path for float and double (conv_layer.cpp:102, conv_layer.cu:75,
 *     math_functions.cu:696-704,765-766, math_functions.cpp:178-199).  Every unsuffixed entry point is the float
 *     one; the `_f64` twins take double.  A plan's type is fixed by the align call (weight_align[_cpu][_f64] /
 *     set_csr[_f64]); calling the other type's forward on it is ESCOIN_ESTATE.  north_star measures fp32: double
 *     plans run the order-preserving generic kernel on the device (fp64 vector FMA is native on gfx950) and the same
 *     host kernel as float in CPU mode -- no LDS-tiled / generated-code / MFMA fast path.
 */
#ifndef ESCOIN_H_
#define ESCOIN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Symbol visibility: the library is built with -fvisibility=hidden. */
#if defined(ESCOIN_BUILD) && defined(__GNUC__)
#define ESCOIN_API __attribute__((visibility("default")))
#else
#define ESCOIN_API
#endif

#define ESCOIN_OK 0
#define ESCOIN_EINVAL (-1)    /* bad argument / unsupported geometry            */
#define ESCOIN_ENOMEM (-2)    /* host or device allocation failed               */
#define ESCOIN_EHIP (-3)      /* a HIP runtime call or kernel launch failed     */
#define ESCOIN_ESTATE (-4)    /* forward before weight_align / set_csr          */
#define ESCOIN_ENODEVICE (-5) /* no HIP device visible                          */

/* Caffe::ConvMode, include/caffe/common.hpp:112.  The two direct-sparse modes differ in the
 * reference only by launch granularity (per image vs whole batch, conv_layer.cu:16-26) and produce
 * the same numbers; LOWERED_GEMM is the dense MFMA kernel, LOWERED_SPARSE the lowering comparator. */
#define ESCOIN_CONV_MODE_LOWERED_GEMM 0
#define ESCOIN_CONV_MODE_LOWERED_SPARSE 1
#define ESCOIN_CONV_MODE_SCONV 2
#define ESCOIN_CONV_MODE_SCONV_PAR 3

/* Which kernel family a plan uses (escoin_plan_set_option("kernel", ...)). */
#define ESCOIN_KERNEL_AUTO 0
#define ESCOIN_KERNEL_GENERIC 1 /* one lane = one output pixel, CSR order kept    */
#define ESCOIN_KERNEL_TILED 2   /* LDS-staged tiles, row-grouped weight stream    */
#define ESCOIN_KERNEL_DENSE 3   /* implicit-GEMM on the fp32 matrix cores (MFMA)  */
#define ESCOIN_KERNEL_JIT 4     /* LDS-staged tiles; the weight walk is machine code
                                   WeightAlign generated from the sparsity pattern
                                   (what AUTO runs on every stride-1 layer -- and on
                                   1x1 / stride-2 layers -- that it keeps on the sparse
                                   path; the one exception is the small-launch rule,
                                   escoin_plan_stat "small_launch_rule")              */

/* Geometry of one ConvolutionLayer: what LayerSetUp/Reshape derive from
 * ConvolutionParameter + the bottom shape (base_conv_layer.cpp:276-530). */
typedef struct escoin_conv_desc {
  int N;        /* num_: largest batch a forward call may carry                 */
  int C, H, W;  /* channels_ (all groups), conv_input_shape_[1..2]              */
  int M;        /* num_output_ (all groups)                                     */
  int KH, KW;   /* kernel_shape_                                                */
  int pad_h, pad_w;
  int stride_h, stride_w;
  int dil_h, dil_w;
  int group;    /* group_                                                       */
  int has_bias; /* bias_term_: forward adds bias[oc] once (conv_layer.cu:21-24) */
  int fuse_relu;/* ConvolutionReLU: max(x + bias, 0) (conv_relu_layer.cu:8-30)  */
} escoin_conv_desc;

typedef struct escoin_plan escoin_plan;

/* Message of this thread's last failed call ("" if none). */
ESCOIN_API const char *escoin_last_error(void);

/* Number of visible HIP devices (0 without a GPU; never fails). */
ESCOIN_API int escoin_device_count(void);

/* ConvolutionLayer::compute_output_shape, conv_layer.cpp:8-22. */
ESCOIN_API int escoin_out_shape(const escoin_conv_desc *desc, int *out_h, int *out_w);

/* Length in elements of the reference's shared-halo padded image C(H+ph)(W+pw) + ph(W+2pw),
 * base_conv_layer.cpp:71,596 -- plus pad_w elements when pad_h == 0 < pad_w: the last row's right padding is read
 * out of the elements that follow the row, and without a bottom padding row nothing follows the last channel's last
 * row (the reference's buffer is pad_w short there and its kernels read past it; none of its models has such a
 * layer).  A zeroed buffer of this length is safe for escoin_gpu_sconv / escoin_cpu_sconv / copy_input_data. */
ESCOIN_API long escoin_padded_len(const escoin_conv_desc *desc);

/* LayerSetUp + Reshape: validates the geometry and creates an empty plan.
 * No device work happens here, so it also succeeds on a machine without a GPU. */
ESCOIN_API int escoin_plan_create(const escoin_conv_desc *desc, escoin_plan **plan);
ESCOIN_API int escoin_plan_destroy(escoin_plan *plan);

/* Options (all but "conv_mode", "cpu_channel_block" and "cpu_images_per_job" must precede weight_align / set_csr):
 *   "kernel"     = ESCOIN_KERNEL_*;
 *   "conv_mode"  = Caffe::ConvMode (common.hpp:112; tools/caffe.cpp:292-301 -conv_mode N):
 *                  SCONV / SCONV_PAR = the direct sparse path (the two differ only in the
 *                  reference's batching), LOWERED_SPARSE = the comparator (im2col + CSR x dense per
 *                  image, base_conv_layer.cpp:724-736), LOWERED_GEMM = every group on the dense
 *                  fp32-MFMA implicit-GEMM kernel (forward_gpu_gemm, base_conv_layer.cpp:713-746).
 *                  May be switched on an aligned plan;
 *   "dense_gate" = 0/1.  0 (default): KERNEL_AUTO sends EACH conv group whose own density exceeds
 *                  the measured sparse/dense crossover to the MFMA kernel and the others to the
 *                  sparse kernel.  1: reproduce the reference's gate -- density(group 0) > 0.2
 *                  sends the whole layer to the dense path (base_conv_layer.cpp:750-755,805-811);
 *   "dense_threshold_pct" = density in percent above which AUTO picks the dense kernel
 *                  (-1 = the built-in measured crossover; 100 = never);
 *   "tiling_batch" = choose the tiled kernel's tiling as for a batch of this many images
 *                  (0 = desc.N).  Results do not depend on it; tests use it to run small inputs
 *                  through the weight stream / tile shapes of the full-size configurations;
 *   "stream_stores" = 1: pointwise (1x1, stride 1, pad 0) layers write the top blob with non-temporal
 *                  stores -- for a blob nothing on the device reads next (3-4 % on a sequence of
 *                  independent layers); 0 / -1 (default): ordinary stores, which leave the blob in
 *                  L2 / Infinity Cache for the layer that consumes it (as the reference's kernels
 *                  do, math_functions.cu:524-587).  Results are the same either way;
 *   "max_launch_bytes" = bottom-blob bytes one launch of the LDS-tiled kernels may cover (0 = the 4 GiB range of
 *                  a buffer descriptor; larger batches run as consecutive sub-batch launches).  Results do not
 *                  depend on it; tests use it to exercise the sub-batch loop on small inputs.
 *   "cpu_channel_block" = input channels per block of escoin_forward_cpu's channel blocking (0 = chosen from the
 *                  geometry so that a tile's input window stays in L1; stride-1 layers only).  Results do not depend on
 *                  it (a row's sum continues across blocks in CSR order); tests use it to block small inputs.  May be set
 *                  on an aligned plan; stat "cpu_channel_block" = what the last escoin_forward_cpu used.
 *   "cpu_images_per_job" = escoin_forward_cpu runs small images (a few vectors each: 7 x 7, 4 x 4, LeNet's 8 x 8) two or three
 *                  to a job, one broadcast weight feeding every image's accumulators; 0 = chosen from the geometry,
 *                  n = at most n.  Results do not depend on it; tests use it.  May be set on an aligned plan.
 *   "code_loader" = how WeightAlign / import_aligned put generated code on the device.  0 (default): executable
 *                  device memory from the ROCm runtime's allocator, filled by a copy kernel (~0.1 ms per megabyte),
 *                  and the code object loader where that is not to be had; 1: always the code object loader
 *                  (hipModuleLoadData on the code wrapped in a code object, 0.6-1 ms per megabyte).  The code is the
 *                  same words either way; tests use 1 to exercise the fallback.  stat "code_direct" says which.
 * Environment: the product build reads ESCOIN_VERBOSE (diagnostics on stderr) and TMPDIR (temporary file of the
 * code object manager's fallback path) and nothing else -- no environment variable can change a result
 * ("Environment"; csrc/knobs.h for the experiment flavours built by tools/mkabl.sh). */
ESCOIN_API int escoin_plan_set_option(escoin_plan *plan, const char *key, int value);

/* WeightAlign(): dense blobs_[0] (M x C/g x KH x KW, zeros = pruned) -> per-group CSR
 * (caffe_{cpu,gpu}_sparse_dense2csr, math_functions.cpp:77-107 / .cu:103-152), index
 * stretch (base_conv_layer.cpp:96-107 / stretch_kernel math_functions.cu:706-727) and
 * the MI355X-specific blocked weight streams; uploads everything to the device.
 * `dense_w` is a host pointer, or a device pointer when w_on_device != 0. One-time. */
ESCOIN_API int escoin_weight_align(escoin_plan *plan, const float *dense_w, int w_on_device,
                        void *stream);

/* CSR hand-off without the dense blob (what a broadcast receiver calls, the
 * counterpart of NCCL<Dtype>::Broadcast, parallel.cpp:189-200).  Host arrays:
 * rowptr[group*(M/group+1)] group-local and 0-based, colidx/values concatenated over
 * groups, colidx UNSTRETCHED (ic*KH*KW + kr*KW + kc), nnz_per_group[group]. */
ESCOIN_API int escoin_plan_set_csr(escoin_plan *plan, const int *rowptr, const int *colidx,
                        const float *values, const int *nnz_per_group, void *stream);

/* nnz of one group (nz_num_[g], base_conv_layer.cpp:247), or of all groups for
 * group < 0.  Negative on error. */
ESCOIN_API long escoin_plan_nnz(const escoin_plan *plan, int group);

/* Copies the plan's CSR to host arrays sized as in escoin_plan_set_csr.  With
 * stretched != 0 colidx is the reference's stretched index
 * (ic*(H+ph)+kr)*(W+pw)+kc, i.e. exactly nz_weight_indices_ after WeightAlign. */
ESCOIN_API int escoin_plan_get_csr(const escoin_plan *plan, int *rowptr, int *colidx, float *values,
                        int stretched);

/* Device bytes owned by the plan (CSR + weight streams / generated code + scratch). */
ESCOIN_API size_t escoin_plan_workspace_bytes(const escoin_plan *plan);

/* The aligned form as one relocatable byte blob: the CSR and -- for a generated-code plan -- the
 * channel deal, the unit table and the machine code WeightAlign generated.  The reference recomputes
 * its aligned form on every weight load (Net::CopyTrainedLayersFrom -> WeightAlign, net.cpp:819);
 * here WeightAlign also compiles code (2-25 ms per layer since round 4, profiles/r04_weight_align.md), and a
 * deployment may persist the result once.
 *   export: buf == NULL queries the size (*bytes); otherwise writes *bytes <= capacity bytes.
 *   import: restores the CSR (always) and, when the blob's code section was written by this library
 *           build for this geometry / batch / tiling_batch, for the same split of the conv groups between the
 *           sparse and the dense kernel (options dense_threshold_pct, dense_gate, conv_mode decide it) and on a
 *           device of the same ISA and CU count, puts the persisted code on the device as it is -- no channel deal, no generator pass, no assembler, no code object loader
 *           (escoin_plan_stat(plan, "import_fast") == 1); otherwise it aligns from the CSR like
 *           escoin_plan_set_csr -- also when the persisted code cannot be placed on this device.  Plan options must be
 *           set before the import, as before weight_align.  Every field of the blob is range-checked before it
 *           sizes or indexes anything; a malformed blob gives ESCOIN_EINVAL, never a crash.  The header carries 64-bit
 *           content tags of the CSR section and of the code section and one binding the two: a blob whose code does not
 *           belong to its CSR (a torn broadcast, a spliced cache file) is refused with ESCOIN_EINVAL -- it can never
 *           run stale code on new weights.  (An integrity check against accidents, not a security boundary.) */
ESCOIN_API int escoin_plan_export_aligned(const escoin_plan *plan, void *buf, size_t capacity, size_t *bytes);
ESCOIN_API int escoin_plan_import_aligned(escoin_plan *plan, const void *buf, size_t bytes, void *stream);
/* The same from a DEVICE buffer -- where an RCCL broadcast leaves the blob (NCCL<Dtype>::Broadcast, parallel.cpp:189-200):
 * one copy into a pinned staging area the calling thread keeps between calls, then the import above. */
ESCOIN_API int escoin_plan_import_aligned_dev(escoin_plan *plan, const void *dev_buf, size_t bytes, void *stream);

/* Integer facts about an aligned plan (negative = error): "align_us" wall time of the last
 * weight_align / set_csr / import_aligned, "code_bytes" generated machine code on the device,
 * "device_bytes", "import_fast", "code_direct" (1: the plan's generated code sits in executable device memory the library
 * filled itself, 0: in a module the HIP loader loaded -- option "code_loader"), "cpu_channel_block" / "cpu_images_per_job"
 * (what the last escoin_forward_cpu used), "jit_rows", "jit_records", "lds_bytes", "workgroup_columns",
 * "kernel_choice" (the ESCOIN_KERNEL_* id AUTO resolved to for the sparse groups), "small_launch_rule" (KERNEL_AUTO's
 * rule for pointwise launches under 64 MFLOP that fit one round of workgroups -- the reference's SCONV mode runs
 * image by image, conv_layer.cu:16-26 --: 0 not considered, 1 kept generated code, 2 took the generic kernel (decided
 * from the tiling before any code is generated or loaded); a function of the options, the weights, the batch and the
 * device MODEL -- its CU count and whether generated code is available -- never of a timing: the same in every process
 * on the same kind of device),
 * "deal_slowest_over_mean_x1000" / "deal_worst_block_x1000" (generated-code plans aligned from weights: how evenly
 * WeightAlign dealt the output channels over the waves that meet at a block's barrier -- slowest wave / mean wave,
 * weighted over all blocks, and the worst single block, x 1000; 0 for a plan restored from a persisted code object),
 * "streamk" / "streamk_gave_up"
 * (dense kernel; a give-up also makes the next escoin_forward on the plan fail with ESCOIN_EHIP). */
ESCOIN_API long escoin_plan_stat(const escoin_plan *plan, const char *key);

/* Name of the device kernel the plan launches (the symbol rocprofv3 reports). */
ESCOIN_API const char *escoin_plan_kernel_name(const escoin_plan *plan);

/* One line describing how the plan's fast kernel tiles the layer (channels per wave, workgroup columns,
 * images per tile, quads per lane, input-channel blocks, plane buffers, generated code or weight
 * stream); "" for plans that run the generic / dense / lowered kernel.  Diagnostics only: the
 * reference has no counterpart (its kernels are fixed-shape, math_functions.cu:524-587).  The
 * string lives as long as the plan and changes with weight_align / set_csr / set_option. */
ESCOIN_API const char *escoin_plan_tiling_info(const escoin_plan *plan);

/* Forward_gpu body for one bottom/top pair, whole batch, asynchronous on `stream`:
 *   top[n][oc] = sconv(bottom[n], CSR)[oc] (+ bias[oc]) (then ReLU if fuse_relu)
 * bottom: n_images x C x H x W, top: n_images x M x OH x OW, bias: M floats or NULL.
 * bias may be NULL even when has_bias was set (the reference dereferences blobs_[1]
 * unconditionally, base_conv_layer.cpp:648 -- not copied).  n_images <= desc.N. */
ESCOIN_API int escoin_forward(escoin_plan *plan, const float *bottom_dev, const float *bias_dev,
                   float *top_dev, int n_images, void *stream);

/* ---- math_functions-level entry points (drop-ins for the reference's GPU helpers,
 * include/caffe/util/math_functions.hpp:194-230).  They operate on the reference's own
 * data layout (shared-halo padded input, stretched CSR) so a Caffe-HIP tree can call
 * them from an unmodified base_conv_layer.cpp. ------------------------------------ */

/* caffe_gpu_sconv<float>, math_functions.cu:590-704 (bias applied iff FUSE_RELU, as
 * there: :215,421 vs :282).  `input` is the padded image(s), per-image stride
 * ifmap_size*num_groups floats (math_functions.cu:566).  Like the reference's kernel it reads
 * the last row's right padding out of the floats that FOLLOW the row; the reference's buffer
 * (base_conv_layer.cpp:71) has those behind the last channel only when pad_h >= 1 -- with
 * pad_h == 0 < pad_w the caller must provide pad_w zero floats more behind the last image
 * (the layer-level path, escoin_forward, needs no padded buffer and has no such case). */
ESCOIN_API int escoin_gpu_sconv(int fuse_relu, int num, const float *input, int ifmap_size,
                     const int *rowptr, const int *colidx, const float *values,
                     const float *bias, int height, int width, int pad_h, int pad_w,
                     int stride_h, int stride_w, int dilation_h, int dilation_w,
                     int kernel_h, int kernel_w, float *output, int num_oc, int num_groups,
                     void *stream);

/* caffe_gpu_stretch, math_functions.cu:706-727 (in place on device colidx). */
ESCOIN_API int escoin_gpu_stretch(const int *rowptr, int *colidx, int M, int height, int width,
                       int pad_h, int pad_w, int kernel_h, int kernel_w, void *stream);

/* copy_input_data<float>, math_functions.cu:729-766: dense image -> padded layout. */
ESCOIN_API int escoin_copy_input_data(float *dst, const float *src, int num_channels, int height,
                           int width, int pad_h, int pad_w, void *stream);

/* caffe_gpu_sparse_csrmm<float>, math_functions.cu:48-62 (cusparseScsrmm2 + cublasSgeam there):
 * C[M x N] = alpha * A_csr[M x K] * B[K x N] + beta * C, all row-major on the device, 0-based
 * CSR with ascending columns.  No transpose scratch: C comes out row-major directly.  The
 * LOWERED_SPARSE comparator (conv_mode 1: im2col + this) is built on it; it is a baseline to
 * measure the direct path against, not the product path. */
ESCOIN_API int escoin_gpu_sparse_csrmm(int M, int N, int K, int nnz, float alpha, const float *values,
                            const int *rowptr, const int *colidx, const float *B, float beta,
                            float *C, void *stream);

/* caffe_gpu_sparse_dense2csr<float>, math_functions.cu:103-152: device dense M x N ->
 * device CSR (0-based, ascending columns); *nnz_total written on the host. */
ESCOIN_API int escoin_gpu_sparse_dense2csr(int M, int N, const float *A, int *nnz_per_row,
                                float *A_nonzero_buf, int *A_idx_pointer_buf,
                                int *A_nonzero_idx_buf, int *nnz_total, void *stream);

/* ---- Dtype = double on the device (conv_layer.cu:75; math_functions.cu:696-704,765-766) ----------------------
 * Same contracts as the float entry points above.  escoin_forward_f64 runs the order-preserving generic kernel
 * (one lane per output pixel, CSR order, fp64 fused multiply-add) in every conv_mode: its results are bit-identical to
 * caffe_cpu_sconv<double>'s.  The aligned-form export / import is defined for float plans only. */
ESCOIN_API int escoin_weight_align_f64(escoin_plan *plan, const double *dense_w, int w_on_device, void *stream);
ESCOIN_API int escoin_plan_set_csr_f64(escoin_plan *plan, const int *rowptr, const int *colidx, const double *values,
                            const int *nnz_per_group, void *stream);
ESCOIN_API int escoin_plan_get_csr_f64(const escoin_plan *plan, int *rowptr, int *colidx, double *values, int stretched);
ESCOIN_API int escoin_forward_f64(escoin_plan *plan, const double *bottom_dev, const double *bias_dev, double *top_dev,
                       int n_images, void *stream);
/* caffe_gpu_sconv<double>, copy_input_data<double>, caffe_gpu_sparse_csrmm<double>, caffe_gpu_sparse_dense2csr<double> */
ESCOIN_API int escoin_gpu_sconv_f64(int fuse_relu, int num, const double *input, int ifmap_size, const int *rowptr,
                         const int *colidx, const double *values, const double *bias, int height, int width,
                         int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w,
                         int kernel_h, int kernel_w, double *output, int num_oc, int num_groups, void *stream);
ESCOIN_API int escoin_copy_input_data_f64(double *dst, const double *src, int num_channels, int height, int width,
                               int pad_h, int pad_w, void *stream);
ESCOIN_API int escoin_gpu_sparse_csrmm_f64(int M, int N, int K, int nnz, double alpha, const double *values,
                                const int *rowptr, const int *colidx, const double *B, double beta, double *C,
                                void *stream);
ESCOIN_API int escoin_gpu_sparse_dense2csr_f64(int M, int N, const double *A, int *nnz_per_row, double *A_nonzero_buf,
                                    int *A_idx_pointer_buf, int *A_nonzero_idx_buf, int *nnz_total, void *stream);

/* ---- Caffe::CPU mode (Caffe::set_mode(Caffe::CPU)): host entry points, no HIP device needed ---------------------
 *   ConvolutionLayer<Dtype>::Forward_cpu          conv_layer.cpp:25-63
 *     -> BaseConvolutionLayer::forward_cpu_sconv  base_conv_layer.cpp:569-661 (pad copy :601-620, groups :626-658)
 *        -> caffe_cpu_sconv<Dtype>                math_functions.cpp:128-176
 *     -> forward_cpu_bias                         base_conv_layer.cpp:663-669
 *   WeightAlign, CPU branch                       base_conv_layer.cpp:46-107
 * Implemented in this library (csrc/sconv_cpu.cpp, sconv_cpu_kernel.cpp: AVX2 / AVX-512 register tiles over the
 * shared-halo layout, a team of host threads over images and output channels).  Every output is
 * fma(value_j, input_j, sum) over its row's nonzeros in CSR order from zero, + bias once afterwards, then ReLU --
 * the arithmetic of the reference's loop nest, so results are BIT-IDENTICAL to caffe_cpu_sconv + forward_cpu_bias for
 * float and for double, whatever the thread count.  Differences kept on purpose: the density gate that sends a layer
 * to the dense GEMM above 50 % density (base_conv_layer.cpp:572-577) is not reproduced (the sparse kernel computes the
 * same sums), every Caffe::ConvMode takes this path (the reference's Forward_cpu uses it for SCONV only), and bias
 * may be NULL. */

/* Which flavour of the host kernel this machine runs ("escoin_cpu_sconv_avx512" / "..._avx2"). */
ESCOIN_API const char *escoin_cpu_kernel_name(void);
/* Pins the flavour for the calls that follow, process-wide: "avx2", "avx512" (ESCOIN_ENODEVICE when this CPU lacks it)
 * or "auto" (what the CPU reports; the default).  Results are bit-identical in every flavour; the tests use this to run
 * both on one machine. */
ESCOIN_API int escoin_cpu_kernel_select(const char *which);

/* WeightAlign() in CPU mode: dense blobs_[0] on the host -> the plan's host CSR.  No device work; a plan aligned this
 * way serves escoin_forward_cpu only (escoin_forward needs escoin_weight_align).  Conversely escoin_forward_cpu also
 * works on a plan aligned by escoin_weight_align / set_csr / import_aligned: the CSR is on the host either way. */
ESCOIN_API int escoin_weight_align_cpu(escoin_plan *plan, const float *dense_w);
ESCOIN_API int escoin_weight_align_cpu_f64(escoin_plan *plan, const double *dense_w);

/* Forward_cpu body for one bottom/top pair, whole batch, host pointers; returns when the batch is done.
 * n_threads: host threads to use (<= 0: all the process may run on).  n_images is not bounded by desc.N.
 * The threads are a process-wide pool of parked std::threads, grown on demand; a new one inherits the CPU affinity of
 * the thread whose call started it (a caller pinned to one core -- e.g. by an OpenMP runtime's OMP_PROC_BIND -- gets
 * a pool pinned to that core: widen the mask around the first call if that is not wanted).  Inside that mask a worker
 * moves itself to a core of its own when it starts and takes the whole mask back at once: the team is spread over the
 * cores from the first call, and nothing stays bound (the scheduler may still move it). */
ESCOIN_API int escoin_forward_cpu(escoin_plan *plan, const float *bottom, const float *bias, float *top, int n_images,
                       int n_threads);
ESCOIN_API int escoin_forward_cpu_f64(escoin_plan *plan, const double *bottom, const double *bias, double *top,
                           int n_images, int n_threads);

/* caffe_cpu_sconv<Dtype>, math_functions.cpp:128-176 (declared include/caffe/util/math_functions.hpp:40-47): one conv
 * group of one image on the reference's padded layout with the stretched CSR; `bias` is accepted and ignored, as there.
 * Reads nothing past the last element an output needs; fails with ESCOIN_EINVAL if that lies at or beyond
 * input_padded_len (the reference asserts it, :168; input_padded_len <= 0 skips the check). */
ESCOIN_API int escoin_cpu_sconv(const float *input_padded, int in_channels, int height, int width, int pad_h, int pad_w,
                     int stride_h, int stride_w, int dilation_h, int dilation_w, const int *rowptr,
                     const int *colidx, const float *values, int kernel_h, int kernel_w, const float *bias,
                     float *output, int out_channels, int input_padded_len);
ESCOIN_API int escoin_cpu_sconv_f64(const double *input_padded, int in_channels, int height, int width, int pad_h,
                         int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w, const int *rowptr,
                         const int *colidx, const double *values, int kernel_h, int kernel_w, const double *bias,
                         double *output, int out_channels, int input_padded_len);

/* caffe_cpu_sparse_dense2csr<Dtype>, the hand loop of math_functions.cpp:92-105 (0-based, ascending columns;
 * argument order as there: values, column indices, row pointers). */
ESCOIN_API int escoin_cpu_sparse_dense2csr(int M, int N, const float *A, float *A_nonzero_buf, int *A_nonzero_idx_buf,
                                int *A_idx_pointer_buf);
ESCOIN_API int escoin_cpu_sparse_dense2csr_f64(int M, int N, const double *A, double *A_nonzero_buf,
                                    int *A_nonzero_idx_buf, int *A_idx_pointer_buf);

#ifdef __cplusplus
}
#endif
#endif /* ESCOIN_H_ */
