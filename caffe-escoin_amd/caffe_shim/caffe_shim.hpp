// caffe_shim.hpp -- header-only mirror of the slice of Caffe's C++ API that the sparse
// convolution path sits behind, so that ConvolutionLayer<Dtype> below could be pasted into a
// Caffe-HIP tree: same class names, method names, argument meaning and (fatal) error behaviour
// as the reference.  It is NOT Caffe: only what the hot path touches exists.
//
//   caffe::Caffe                 include/caffe/common.hpp:102-205   (mode, conv_mode, SetDevice)
//   caffe::SyncedMemory          include/caffe/syncedmem.hpp:56-91, src/caffe/syncedmem.cpp:40-140
//   caffe::Blob<Dtype>           include/caffe/blob.hpp
//   caffe::LayerParameter / ConvolutionParameter   caffe.proto:573-624 as PODs (no protobuf here)
//   caffe::Layer<Dtype>          include/caffe/layer.hpp:33-475 (SetUp, Forward wrapper, WeightAlign)
//   caffe::BaseConvolutionLayer  include/caffe/layers/base_conv_layer.hpp:20-202
//   caffe::ConvolutionLayer      include/caffe/layers/conv_layer.hpp:30-80, conv_layer.cu:8-40
//   caffe::ConvolutionReLULayer  include/caffe/layers/conv_relu_layer.hpp, conv_relu_layer.cu:8-30
//
// Forward_gpu and Forward_cpu both call the C ABI (include/escoin.h): Caffe::GPU mode runs the HIP kernels,
// Caffe::CPU mode the library's host kernel (escoin_forward_cpu; no device is touched, so a CPU-mode net runs on a
// machine without a GPU).  Every class is a template over Dtype = float | double like the reference's
// (INSTANTIATE_CLASS, conv_layer.cpp:102): EscApi<Dtype> below maps the type to the C entry points.
#ifndef ESCOIN_CAFFE_SHIM_HPP_
#define ESCOIN_CAFFE_SHIM_HPP_

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "escoin.h"

// glog-style fatal checks (the reference aborts on every error: device_alternate.hpp:51-78)
#define ESC_CHECK(cond)                                                                     \
  do {                                                                                      \
    if (!(cond)) {                                                                          \
      fprintf(stderr, "Check failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__);             \
      abort();                                                                              \
    }                                                                                       \
  } while (0)
#define ESC_HIP_CHECK(expr)                                                                 \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "HIP_CHECK failed: %s: %s (%s:%d)\n", #expr, hipGetErrorString(e__),  \
              __FILE__, __LINE__);                                                          \
      abort();                                                                              \
    }                                                                                       \
  } while (0)
#define ESCOIN_CHECK(expr)                                                                  \
  do {                                                                                      \
    int rc__ = (expr);                                                                      \
    if (rc__ != 0) {                                                                        \
      fprintf(stderr, "ESCOIN_CHECK failed: %s -> %d: %s (%s:%d)\n", #expr, rc__,           \
              escoin_last_error(), __FILE__, __LINE__);                                     \
      abort();                                                                              \
    }                                                                                       \
  } while (0)
#define NOT_IMPLEMENTED                                                                     \
  do {                                                                                      \
    fprintf(stderr, "Not Implemented Yet (%s:%d)\n", __FILE__, __LINE__);                   \
    abort();                                                                                \
  } while (0)

namespace caffe {

using std::shared_ptr;
using std::string;
using std::vector;

// common.hpp:102-205.  Thread-local singleton like the reference (common.cpp:13-19).
class Caffe {
 public:
  enum Brew { CPU, GPU };
  enum ConvMode { LOWERED_GEMM, LOWERED_SPARSE, SCONV, SCONV_PAR };   // common.hpp:112
  static Caffe &Get() {
    static thread_local Caffe instance;
    return instance;
  }
  static Brew mode() { return Get().mode_; }
  static void set_mode(Brew m) { Get().mode_ = m; }
  static ConvMode conv_mode() { return Get().conv_mode_; }
  static void set_conv_mode(ConvMode m) { Get().conv_mode_ = m; }    // common.hpp:161
  static void SetDevice(int device_id) { ESC_HIP_CHECK(hipSetDevice(device_id)); }
  static hipStream_t stream() { return nullptr; }   // every reference launch uses stream 0
  // host threads of CPU mode (0 = all the process may run on): the reference takes OMP_NUM_THREADS for its ICC build's
  // batch loop (conv_layer.cpp:41-43); there is no OpenMP runtime here, so the number is a setting
  static int cpu_threads() { return Get().cpu_threads_; }
  static void set_cpu_threads(int n) { Get().cpu_threads_ = n; }

 private:
  // the reference leaves conv_mode_ uninitialised; here it defaults to SCONV_PAR
  Caffe() : mode_(CPU), conv_mode_(SCONV_PAR), cpu_threads_(0) {}
  Brew mode_;
  ConvMode conv_mode_;
  int cpu_threads_;
};

// Dtype -> C ABI.  The float entry points are the unsuffixed ones, double the _f64 twins (include/escoin.h).
template <typename Dtype> struct EscApi;
template <> struct EscApi<float> {
  static int weight_align(escoin_plan *p, const float *w, int on_dev, void *s) { return escoin_weight_align(p, w, on_dev, s); }
  static int weight_align_cpu(escoin_plan *p, const float *w) { return escoin_weight_align_cpu(p, w); }
  static int forward(escoin_plan *p, const float *b, const float *bias, float *t, int n, void *s) { return escoin_forward(p, b, bias, t, n, s); }
  static int forward_cpu(escoin_plan *p, const float *b, const float *bias, float *t, int n, int threads) { return escoin_forward_cpu(p, b, bias, t, n, threads); }
};
template <> struct EscApi<double> {
  static int weight_align(escoin_plan *p, const double *w, int on_dev, void *s) { return escoin_weight_align_f64(p, w, on_dev, s); }
  static int weight_align_cpu(escoin_plan *p, const double *w) { return escoin_weight_align_cpu_f64(p, w); }
  static int forward(escoin_plan *p, const double *b, const double *bias, double *t, int n, void *s) { return escoin_forward_f64(p, b, bias, t, n, s); }
  static int forward_cpu(escoin_plan *p, const double *b, const double *bias, double *t, int n, int threads) { return escoin_forward_cpu_f64(p, b, bias, t, n, threads); }
};

// syncedmem.hpp:56-91: lazily mirrored host/device buffer with a head state.
class SyncedMemory {
 public:
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };
  explicit SyncedMemory(size_t size) : cpu_ptr_(nullptr), gpu_ptr_(nullptr), size_(size), head_(UNINITIALIZED) {}
  ~SyncedMemory() {
    if (cpu_ptr_) free(cpu_ptr_);
    if (gpu_ptr_) (void)hipFree(gpu_ptr_);
  }
  const void *cpu_data() { to_cpu(); return cpu_ptr_; }
  const void *gpu_data() { to_gpu(); return gpu_ptr_; }
  void *mutable_cpu_data() { to_cpu(); head_ = HEAD_AT_CPU; return cpu_ptr_; }
  void *mutable_gpu_data() { to_gpu(); head_ = HEAD_AT_GPU; return gpu_ptr_; }
  SyncedHead head() const { return head_; }
  size_t size() const { return size_; }

 private:
  void to_cpu() {   // syncedmem.cpp:40-64
    switch (head_) {
      case UNINITIALIZED:
        cpu_ptr_ = calloc(1, size_ ? size_ : 1);
        ESC_CHECK(cpu_ptr_);
        head_ = HEAD_AT_CPU;
        break;
      case HEAD_AT_GPU:
        if (!cpu_ptr_) { cpu_ptr_ = malloc(size_ ? size_ : 1); ESC_CHECK(cpu_ptr_); }
        ESC_HIP_CHECK(hipMemcpy(cpu_ptr_, gpu_ptr_, size_, hipMemcpyDeviceToHost));
        head_ = SYNCED;
        break;
      default: break;
    }
  }
  void to_gpu() {   // syncedmem.cpp:66-92
    switch (head_) {
      case UNINITIALIZED:
        ESC_HIP_CHECK(hipMalloc(&gpu_ptr_, size_ ? size_ : 1));
        ESC_HIP_CHECK(hipMemset(gpu_ptr_, 0, size_));
        head_ = HEAD_AT_GPU;
        break;
      case HEAD_AT_CPU:
        if (!gpu_ptr_) ESC_HIP_CHECK(hipMalloc(&gpu_ptr_, size_ ? size_ : 1));
        ESC_HIP_CHECK(hipMemcpy(gpu_ptr_, cpu_ptr_, size_, hipMemcpyHostToDevice));
        head_ = SYNCED;
        break;
      default: break;
    }
  }
  void *cpu_ptr_, *gpu_ptr_;
  size_t size_;
  SyncedHead head_;
};

template <typename Dtype>
class Blob {   // blob.hpp (data only: the forward path never touches diff_)
 public:
  Blob() : count_(0), capacity_(0) {}
  explicit Blob(const vector<int> &shape) : count_(0), capacity_(0) { Reshape(shape); }
  Blob(int num, int channels, int height, int width) : count_(0), capacity_(0) {
    Reshape(num, channels, height, width);
  }
  void Reshape(int num, int channels, int height, int width) {
    vector<int> s(4);
    s[0] = num; s[1] = channels; s[2] = height; s[3] = width;
    Reshape(s);
  }
  void Reshape(const vector<int> &shape) {   // blob.cpp: grows, never shrinks
    count_ = 1;
    shape_ = shape;
    for (size_t i = 0; i < shape.size(); ++i) { ESC_CHECK(shape[i] >= 0); count_ *= shape[i]; }
    if (count_ > capacity_) {
      capacity_ = count_;
      data_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
    }
  }
  const vector<int> &shape() const { return shape_; }
  int shape(int i) const { return shape_[i < 0 ? i + (int)shape_.size() : i]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  int count(int start, int end) const {
    int c = 1;
    for (int i = start; i < end; ++i) c *= shape(i);
    return c;
  }
  int num() const { return shape(0); }
  int channels() const { return shape(1); }
  int height() const { return shape(2); }
  int width() const { return shape(3); }
  const Dtype *cpu_data() const { ESC_CHECK(data_); return (const Dtype *)data_->cpu_data(); }
  const Dtype *gpu_data() const { ESC_CHECK(data_); return (const Dtype *)data_->gpu_data(); }
  Dtype *mutable_cpu_data() { ESC_CHECK(data_); return (Dtype *)data_->mutable_cpu_data(); }
  Dtype *mutable_gpu_data() { ESC_CHECK(data_); return (Dtype *)data_->mutable_gpu_data(); }

 private:
  shared_ptr<SyncedMemory> data_;
  vector<int> shape_;
  int count_, capacity_;
};

// caffe.proto:573-624 as a POD (no protobuf runtime for C++ in this environment)
struct ConvolutionParameter {
  int num_output = 0;
  bool bias_term = true;
  int pad_h = 0, pad_w = 0;
  int kernel_h = 0, kernel_w = 0;
  int stride_h = 1, stride_w = 1;
  int dilation = 1;
  int group = 1;
};
struct LayerParameter {
  string name, type;
  ConvolutionParameter convolution_param;
};

template <typename Dtype>
class Layer {   // layer.hpp:33-475
 public:
  explicit Layer(const LayerParameter &param) : layer_param_(param), test_time_(0) {}
  virtual ~Layer() {}
  void SetUp(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {   // :69-77
    ESC_CHECK((int)bottom.size() >= MinBottomBlobs());
    ESC_CHECK((int)top.size() >= MinTopBlobs());
    if (EqualNumBottomTopBlobs()) ESC_CHECK(bottom.size() == top.size());
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
  }
  virtual void LayerSetUp(const vector<Blob<Dtype> *> &, const vector<Blob<Dtype> *> &) {}
  virtual void Reshape(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) = 0;
  virtual void WeightAlign() {}   // layer.hpp:97-98, called by Net::CopyTrainedLayersFrom (net.cpp:819)
  // layer.hpp:435-475: Reshape every call, mode switch, per-layer forward time in microseconds
  inline Dtype Forward(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {
    Reshape(bottom, top);
    hipEvent_t e0, e1;
    const bool gpu = Caffe::mode() == Caffe::GPU;
    if (gpu) {
      ESC_HIP_CHECK(hipEventCreate(&e0));
      ESC_HIP_CHECK(hipEventCreate(&e1));
      ESC_HIP_CHECK(hipEventRecord(e0, Caffe::stream()));
    }
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    switch (Caffe::mode()) {
      case Caffe::CPU: Forward_cpu(bottom, top); break;
      case Caffe::GPU: Forward_gpu(bottom, top); break;
    }
    if (!gpu) test_time_ = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (gpu) {
      ESC_HIP_CHECK(hipEventRecord(e1, Caffe::stream()));
      ESC_HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      ESC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      test_time_ = ms * 1000.f;
      ESC_HIP_CHECK(hipEventDestroy(e0));
      ESC_HIP_CHECK(hipEventDestroy(e1));
    }
    return 0;
  }
  vector<shared_ptr<Blob<Dtype> > > &blobs() { return blobs_; }
  const LayerParameter &layer_param() const { return layer_param_; }
  virtual inline const char *type() const { return ""; }
  virtual inline int MinBottomBlobs() const { return -1; }
  virtual inline int MinTopBlobs() const { return -1; }
  virtual inline bool EqualNumBottomTopBlobs() const { return false; }
  float get_time() const { return test_time_; }   // layer.hpp:99

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) = 0;
  virtual void Forward_gpu(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {
    Forward_cpu(bottom, top);
  }
  LayerParameter layer_param_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  float test_time_;
};

template <typename Dtype>
class BaseConvolutionLayer : public Layer<Dtype> {   // base_conv_layer.hpp:20-202
 public:
  explicit BaseConvolutionLayer(const LayerParameter &param)
      : Layer<Dtype>(param), plan_(nullptr), num_(0), channels_(0), group_(1), num_output_(0),
        bias_term_(true), fuse_relu_(false), planned_num_(0) {}
  virtual ~BaseConvolutionLayer() { if (plan_) escoin_plan_destroy(plan_); }   // base_conv_layer.cpp:16-42

  // base_conv_layer.cpp:276-446: parse the conv params, allocate and shape blobs_
  virtual void LayerSetUp(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &) {
    const ConvolutionParameter &cp = this->layer_param_.convolution_param;
    ESC_CHECK(bottom[0]->num_axes() == 4);
    ESC_CHECK(cp.kernel_h > 0 && cp.kernel_w > 0 && cp.num_output > 0);
    channels_ = bottom[0]->shape(1);
    num_output_ = cp.num_output;
    group_ = cp.group;
    bias_term_ = cp.bias_term;
    ESC_CHECK(channels_ % group_ == 0);     // :393
    ESC_CHECK(num_output_ % group_ == 0);   // :395
    this->blobs_.resize(bias_term_ ? 2 : 1);   // :423-427
    vector<int> wshape(4);
    wshape[0] = num_output_; wshape[1] = channels_ / group_; wshape[2] = cp.kernel_h; wshape[3] = cp.kernel_w;
    this->blobs_[0].reset(new Blob<Dtype>(wshape));
    if (bias_term_) this->blobs_[1].reset(new Blob<Dtype>(vector<int>(1, num_output_)));
  }

  // base_conv_layer.cpp:449-530: output shape, all bottoms identical, top reshape
  virtual void Reshape(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {
    const ConvolutionParameter &cp = this->layer_param_.convolution_param;
    num_ = bottom[0]->shape(0);
    ESC_CHECK(bottom[0]->shape(1) == channels_);
    for (size_t i = 1; i < bottom.size(); ++i) ESC_CHECK(bottom[0]->shape() == bottom[i]->shape());   // :458-461
    escoin_conv_desc d = desc(bottom[0], cp);
    int oh = 0, ow = 0;
    ESCOIN_CHECK(escoin_out_shape(&d, &oh, &ow));   // compute_output_shape, conv_layer.cpp:8-22
    for (size_t i = 0; i < top.size(); ++i) top[i]->Reshape(num_, num_output_, oh, ow);
    bottom_dim_ = bottom[0]->count(1, 4);
    top_dim_ = top[0]->count(1, 4);
    escoin_conv_desc same_but_n = d;
    same_but_n.N = plan_ ? desc_.N : d.N;
    const bool reusable = plan_ && d.N <= desc_.N && memcmp(&same_but_n, &desc_, sizeof(d)) == 0;
    if (!reusable) {
      // geometry changed, batch grew past the planned one, or first call: a fresh plan.  The
      // reference's Reshape leaves the CSR blobs alone (they depend on the weights only) but its
      // stretched indices and padded buffer go stale until the next WeightAlign; here a layer
      // that was aligned is re-aligned from blobs_[0] on the spot, so Forward keeps working.
      escoin_plan *fresh = nullptr;
      ESCOIN_CHECK(escoin_plan_create(&d, &fresh));
      if (plan_) escoin_plan_destroy(plan_);
      plan_ = fresh;
      desc_ = d;
      const bool was_aligned = aligned_;
      aligned_ = false;
      if (was_aligned) WeightAlign();
    }
  }

  // base_conv_layer.cpp:46-273: dense blobs_[0] -> CSR (+ the device weight streams), once
  virtual void WeightAlign() {
    ESC_CHECK(plan_ != nullptr);   // SetUp must have run
    // Caffe::ConvMode and ESCOIN_CONV_MODE_* share their values (common.hpp:112).  The reference's
    // WeightAlign only builds the CSR for the sparse modes and leaves LOWERED_GEMM alone
    // (base_conv_layer.cpp:49-53); here every mode is served from the aligned plan.
    ESCOIN_CHECK(escoin_plan_set_option(plan_, "conv_mode", (int)Caffe::conv_mode()));
    aligned_mode_ = Caffe::conv_mode();
    // GPU mode: CSR + the device weight streams / generated code (CSR branch of base_conv_layer.cpp:109-273).
    // CPU mode: the host CSR only (:46-107) -- no device is touched; a later Forward in GPU mode aligns the device
    // side on the spot (forward_gpu_sconv_par below).
    if (Caffe::mode() == Caffe::GPU) {
      ESCOIN_CHECK(EscApi<Dtype>::weight_align(plan_, this->blobs_[0]->gpu_data(), 1, Caffe::stream()));
      aligned_on_device_ = true;
    } else {
      ESCOIN_CHECK(EscApi<Dtype>::weight_align_cpu(plan_, this->blobs_[0]->cpu_data()));
      aligned_on_device_ = false;
    }
    aligned_ = true;
  }

  // The aligned form as one byte blob (CSR + channel deal + unit table + code object), and WeightAlign from such
  // a blob instead of from blobs_[0]: what a Net::CopyTrainedLayersFrom that finds "<model>.escoin/<layer>.bin"
  // next to the .caffemodel would call (the reference recomputes the aligned form at every load, net.cpp:819; here
  // it contains compiled code, so a deployment persists it once).  Returns true when the persisted code object was
  // loaded as it was; a blob written for other options / batch / device falls back to aligning from its CSR.
  vector<unsigned char> ExportAligned() const {
    ESC_CHECK(plan_ != nullptr && aligned_);
    size_t n = 0;
    ESCOIN_CHECK(escoin_plan_export_aligned(plan_, nullptr, 0, &n));
    vector<unsigned char> blob(n);
    ESCOIN_CHECK(escoin_plan_export_aligned(plan_, blob.data(), blob.size(), &n));
    blob.resize(n);
    return blob;
  }
  bool WeightAlignFrom(const vector<unsigned char> &blob) {
    ESC_CHECK(plan_ != nullptr);
    ESCOIN_CHECK(escoin_plan_set_option(plan_, "conv_mode", (int)Caffe::conv_mode()));
    aligned_mode_ = Caffe::conv_mode();
    ESCOIN_CHECK(escoin_plan_import_aligned(plan_, blob.data(), blob.size(), Caffe::stream()));
    aligned_ = true;
    aligned_on_device_ = true;
    return escoin_plan_stat(plan_, "import_fast") == 1;
  }

  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MinTopBlobs() const { return 1; }
  virtual inline bool EqualNumBottomTopBlobs() const { return true; }
  long nnz() const { return plan_ ? escoin_plan_nnz(plan_, -1) : 0; }
  const char *kernel_name() const {
    if (Caffe::mode() == Caffe::CPU) return escoin_cpu_kernel_name();
    return plan_ ? escoin_plan_kernel_name(plan_) : "";
  }

 protected:
  escoin_conv_desc desc(const Blob<Dtype> *b, const ConvolutionParameter &cp) const {
    escoin_conv_desc d;
    d.N = b->shape(0); d.C = b->shape(1); d.H = b->shape(2); d.W = b->shape(3);
    d.M = cp.num_output; d.KH = cp.kernel_h; d.KW = cp.kernel_w;
    d.pad_h = cp.pad_h; d.pad_w = cp.pad_w; d.stride_h = cp.stride_h; d.stride_w = cp.stride_w;
    d.dil_h = cp.dilation; d.dil_w = cp.dilation; d.group = cp.group;
    d.has_bias = cp.bias_term ? 1 : 0; d.fuse_relu = fuse_relu_ ? 1 : 0;
    return d;
  }
  // forward_gpu_sconv_par + forward_gpu_bias (base_conv_layer.cpp:800-856) for the whole batch
  void forward_gpu_sconv_par(const Dtype *input, const Dtype * /*weights*/, Dtype *output) {
    ESC_CHECK(aligned_);   // the reference silently computes zeros here
    if (!aligned_on_device_) {   // WeightAlign ran in CPU mode, the net was switched to GPU mode afterwards
      ESCOIN_CHECK(EscApi<Dtype>::weight_align(plan_, this->blobs_[0]->gpu_data(), 1, Caffe::stream()));
      aligned_on_device_ = true;
    }
    if (Caffe::conv_mode() != aligned_mode_) {   // `caffe test -conv_mode N` flipped after the load
      ESCOIN_CHECK(escoin_plan_set_option(plan_, "conv_mode", (int)Caffe::conv_mode()));
      aligned_mode_ = Caffe::conv_mode();
    }
    const Dtype *bias = bias_term_ ? this->blobs_[1]->gpu_data() : nullptr;
    ESCOIN_CHECK(EscApi<Dtype>::forward(plan_, input, bias, output, num_, Caffe::stream()));
  }
  // Forward_cpu's body for one bottom/top pair: the batch loop of conv_layer.cpp:44-61 (forward_cpu_sconv per image,
  // base_conv_layer.cpp:569-661, then forward_cpu_bias, :663-669) inside the library, on Caffe::cpu_threads() threads
  void forward_cpu_sconv_batch(const Dtype *input, Dtype *output) {
    ESC_CHECK(aligned_);
    const Dtype *bias = bias_term_ ? this->blobs_[1]->cpu_data() : nullptr;
    ESCOIN_CHECK(EscApi<Dtype>::forward_cpu(plan_, input, bias, output, num_, Caffe::cpu_threads()));
  }
  escoin_plan *plan_;
  escoin_conv_desc desc_;
  bool aligned_ = false, aligned_on_device_ = false;
  Caffe::ConvMode aligned_mode_ = Caffe::SCONV_PAR;
  int num_, channels_, group_, num_output_;
  bool bias_term_, fuse_relu_;
  int planned_num_;
  int bottom_dim_ = 0, top_dim_ = 0;
};

template <typename Dtype>
class ConvolutionLayer : public BaseConvolutionLayer<Dtype> {   // conv_layer.hpp:30-80
 public:
  explicit ConvolutionLayer(const LayerParameter &param) : BaseConvolutionLayer<Dtype>(param) {}
  virtual inline const char *type() const { return "Convolution"; }

 protected:
  // conv_layer.cpp:25-63.  Every Caffe::ConvMode takes the direct sparse host kernel (the reference's Forward_cpu
  // uses it for SCONV and the dense GEMM otherwise: same sums, escoin.h "Caffe::CPU mode").
  virtual void Forward_cpu(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {
    for (size_t i = 0; i < bottom.size(); ++i) {
      const Dtype *bottom_data = bottom[i]->cpu_data();
      Dtype *top_data = top[i]->mutable_cpu_data();
      this->forward_cpu_sconv_batch(bottom_data, top_data);
    }
  }
  // conv_layer.cu:8-40.  SCONV and SCONV_PAR produce the same numbers; both go through one
  // batched launch per bottom (the per-image launches of SCONV are a reference artefact).
  virtual void Forward_gpu(const vector<Blob<Dtype> *> &bottom, const vector<Blob<Dtype> *> &top) {
    const Dtype *weight = this->blobs_[0]->gpu_data();
    for (size_t i = 0; i < bottom.size(); ++i) {
      const Dtype *bottom_data = bottom[i]->gpu_data();
      Dtype *top_data = top[i]->mutable_gpu_data();
      this->forward_gpu_sconv_par(bottom_data, weight, top_data);
    }
  }
};

template <typename Dtype>
class ConvolutionReLULayer : public ConvolutionLayer<Dtype> {   // conv_relu_layer.hpp / .cu:8-30
 public:
  explicit ConvolutionReLULayer(const LayerParameter &param) : ConvolutionLayer<Dtype>(param) {
    this->fuse_relu_ = true;
  }
  virtual inline const char *type() const { return "ConvolutionReLU"; }
};

// layer_factory.hpp:53-110 / layer_factory.cpp:74: the creator registry `Net::Init` looks layer
// types up in (LayerRegistry<Dtype>::CreateLayer(param)), with the two creators this path owns.
template <typename Dtype>
class LayerRegistry {
 public:
  typedef shared_ptr<Layer<Dtype> > (*Creator)(const LayerParameter &);
  static shared_ptr<Layer<Dtype> > CreateLayer(const LayerParameter &param) {
    if (param.type == "Convolution") return shared_ptr<Layer<Dtype> >(new ConvolutionLayer<Dtype>(param));
    if (param.type == "ConvolutionReLU") return shared_ptr<Layer<Dtype> >(new ConvolutionReLULayer<Dtype>(param));
    fprintf(stderr, "Unknown layer type: %s (known types: Convolution, ConvolutionReLU)\n", param.type.c_str());
    abort();   // LOG(FATAL), layer_factory.hpp:79-80
  }
  static vector<string> LayerTypeList() {
    vector<string> v;
    v.push_back("Convolution");
    v.push_back("ConvolutionReLU");
    return v;
  }
};

}  // namespace caffe
#endif  // ESCOIN_CAFFE_SHIM_HPP_
