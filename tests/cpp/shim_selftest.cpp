// tests/cpp/shim_selftest.cpp -- drives the C++ Caffe-compatible shim the way the reference's own typed conv tests
// drive ConvolutionLayer (src/caffe/test/test_convolution_layer.cpp: build a LayerParameter, SetUp, fill blobs_,
// Forward, compare with an explicit loop-nest convolution), in the reference's four TestDtypesAndDevices combinations
// {float, double} x {Caffe::CPU, Caffe::GPU} (test_caffe_main.hpp), plus the step the reference's tests never take:
// WeightAlign() and the sparse conv modes.
//
//   shim_selftest            all four combinations (needs a GPU); run by tests/test_shim_gpu.py
//   shim_selftest --cpu-only the two Caffe::CPU combinations; touches no device; run by tests/test_shim_cpu.py
// Prints one line per case, exit code = number of failures.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "caffe_shim.hpp"

using namespace caffe;

static unsigned rng = 2024;
static float frand() {
  rng = rng * 1664525u + 1013904223u;
  return ((rng >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
// GaussianFiller (filler.hpp) stand-in: Box-Muller on the generator above
static double grand() {
  double u1 = (frand() + 1.0) * 0.5, u2 = (frand() + 1.0) * 0.5;
  if (u1 < 1e-6) u1 = 1e-6;
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

template <typename Dtype> struct Tol;
template <> struct Tol<float> { static double abs_near() { return 1e-4; } static double rel() { return 1e-4; } static const char *name() { return "float"; } };
template <> struct Tol<double> { static double abs_near() { return 1e-11; } static double rel() { return 1e-12; } static const char *name() { return "double"; } };
static const char *brew_name() { return Caffe::mode() == Caffe::GPU ? "GPU" : "CPU"; }

// explicit reference convolution in the style of caffe_conv() (test_convolution_layer.cpp:19-140)
template <typename Dtype>
static void naive_conv(const Blob<Dtype> &in, const ConvolutionParameter &cp, const Dtype *w, const Dtype *bias,
                       bool relu, std::vector<double> *out, int oh, int ow) {
  const int N = in.num(), C = in.channels(), H = in.height(), W = in.width();
  const int M = cp.num_output, G = cp.group, Cg = C / G, Mg = M / G;
  const Dtype *x = in.cpu_data();
  out->assign((size_t)N * M * oh * ow, 0.0);
  for (int n = 0; n < N; ++n)
    for (int m = 0; m < M; ++m) {
      const int g = m / Mg;
      for (int y = 0; y < oh; ++y)
        for (int xo = 0; xo < ow; ++xo) {
          long double s = 0.0L;
          for (int c = 0; c < Cg; ++c)
            for (int kr = 0; kr < cp.kernel_h; ++kr)
              for (int kc = 0; kc < cp.kernel_w; ++kc) {
                const int iy = y * cp.stride_h - cp.pad_h + kr * cp.dilation;
                const int ix = xo * cp.stride_w - cp.pad_w + kc * cp.dilation;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                s += (long double)w[(((size_t)m * Cg + c) * cp.kernel_h + kr) * cp.kernel_w + kc] *
                     x[(((size_t)n * C + g * Cg + c) * H + iy) * W + ix];
              }
          if (cp.bias_term) s += bias[m];
          if (relu && s < 0) s = 0;
          (*out)[(((size_t)n * M + m) * oh + y) * ow + xo] = (double)s;
        }
    }
}

template <typename Dtype>
static double compare(const Blob<Dtype> &top, const std::vector<double> &want, double *max_abs) {
  const Dtype *got = top.cpu_data();
  double maxerr = 0, maxref = 0;
  for (size_t k = 0; k < want.size(); ++k) {
    maxerr = std::fmax(maxerr, std::fabs((double)got[k] - want[k]));
    maxref = std::fmax(maxref, std::fabs(want[k]));
  }
  if (max_abs) *max_abs = maxerr;
  return maxerr / std::fmax(1e-6, maxref);
}

// One layer the way the reference's typed tests make it.  sparsity < 0: the reference's own fill (GaussianFiller
// weights, every one nonzero; constant 0.1 bias, test_convolution_layer.cpp:241-243); otherwise pruned weights.
template <typename Dtype, class LayerT>
static int run_case(const char *name, int N, int C, int H, int W, ConvolutionParameter cp, float sparsity, bool relu,
                    int nbottom = 1) {
  LayerParameter lp;
  lp.name = name;
  lp.type = relu ? "ConvolutionReLU" : "Convolution";
  lp.convolution_param = cp;
  std::vector<Blob<Dtype> *> bottom, top;
  std::vector<shared_ptr<Blob<Dtype> > > keep;
  for (int i = 0; i < nbottom; ++i) {
    keep.push_back(shared_ptr<Blob<Dtype> >(new Blob<Dtype>(N, C, H, W)));
    bottom.push_back(keep.back().get());
    keep.push_back(shared_ptr<Blob<Dtype> >(new Blob<Dtype>()));
    top.push_back(keep.back().get());
    Dtype *x = bottom[i]->mutable_cpu_data();
    for (int k = 0; k < bottom[i]->count(); ++k) x[k] = sparsity < 0 ? (Dtype)grand() : (Dtype)frand();
  }
  LayerT layer(lp);
  layer.SetUp(bottom, top);
  // "CopyTrainedLayersFrom": weights into blobs_[0] (exact zeros = pruned), bias into blobs_[1]
  std::vector<Dtype> w(layer.blobs()[0]->count()), bias(cp.num_output, 0);
  for (size_t k = 0; k < w.size(); ++k) {
    if (sparsity < 0) {
      w[k] = (Dtype)grand();
    } else {
      const float v = frand();
      w[k] = (std::fabs(frand()) < sparsity) ? 0.f : (v == 0.f ? 0.25f : v);
    }
  }
  memcpy(layer.blobs()[0]->mutable_cpu_data(), w.data(), sizeof(Dtype) * w.size());
  if (cp.bias_term) {
    for (int k = 0; k < cp.num_output; ++k) bias[k] = sparsity < 0 ? (Dtype)0.1 : (Dtype)(0.1f * frand());
    memcpy(layer.blobs()[1]->mutable_cpu_data(), bias.data(), sizeof(Dtype) * bias.size());
  }
  layer.WeightAlign();                     // net.cpp:819
  layer.Forward(bottom, top);              // net.cpp:568 -> layer.hpp:435
  double worst = 0, worst_abs = 0;
  for (int i = 0; i < nbottom; ++i) {
    std::vector<double> want;
    naive_conv<Dtype>(*bottom[i], cp, w.data(), bias.data(), relu, &want, top[i]->height(), top[i]->width());
    double a = 0;
    worst = std::fmax(worst, compare<Dtype>(*top[i], want, &a));
    worst_abs = std::fmax(worst_abs, a);
  }
  // the reference's own criterion is EXPECT_NEAR(top, ref, 1e-4) on O(1) data (:255-257); the relative bound is
  // north_star's
  const bool ok = worst <= Tol<Dtype>::rel() && (sparsity >= 0 || worst_abs <= Tol<Dtype>::abs_near());
  printf("%-6s %-3s %-26s %-15s top %dx%dx%dx%d nnz=%ld %-38s %.1f us rel_err=%.2e %s\n", Tol<Dtype>::name(),
         brew_name(), name, layer.type(), top[0]->num(), top[0]->channels(), top[0]->height(), top[0]->width(),
         layer.nnz(), layer.kernel_name(), layer.get_time(), worst, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// TestSobelConvolution (test_convolution_layer.cpp:498-589): the Sobel G_x operator as one 3 x 3 stride-2 filter
// against the same operator as a 3 x 1 column filter followed by a 1 x 3 row filter.  Weights with exact zeros, no
// bias, non-square kernels and strides; the blobs are replaced after construction like there (:520-521).
template <typename Dtype>
static int sobel_case() {
  Blob<Dtype> b1(2, 3, 6, 4), b2(2, 3, 6, 4), t1, t2, sep;
  {
    Dtype *x = b1.mutable_cpu_data();
    for (int k = 0; k < b1.count(); ++k) x[k] = (Dtype)grand();
    memcpy(b2.mutable_cpu_data(), b1.cpu_data(), sizeof(Dtype) * b1.count());
  }
  std::vector<Blob<Dtype> *> bv(1, &b1), tv(1, &t1);
  LayerParameter lp;
  lp.type = "Convolution";
  ConvolutionParameter &cp = lp.convolution_param;
  cp.kernel_h = cp.kernel_w = 3; cp.stride_h = cp.stride_w = 2; cp.num_output = 1; cp.bias_term = false;
  {
    ConvolutionLayer<Dtype> layer(lp);
    layer.SetUp(bv, tv);
    Dtype *w = layer.blobs()[0]->mutable_cpu_data();
    const Dtype gx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};
    for (int c = 0; c < 3; ++c) memcpy(w + c * 9, gx, sizeof(gx));
    layer.WeightAlign();
    layer.Forward(bv, tv);
  }
  std::vector<Blob<Dtype> *> sb(1, &b2), st(1, &t2);
  cp.kernel_h = 3; cp.kernel_w = 1; cp.stride_h = 2; cp.stride_w = 1;     // (1) the [1 2 1] column filter
  {
    ConvolutionLayer<Dtype> layer(lp);
    layer.SetUp(sb, st);
    Dtype *w = layer.blobs()[0]->mutable_cpu_data();
    for (int c = 0; c < 3; ++c) { w[c * 3 + 0] = 1; w[c * 3 + 1] = 2; w[c * 3 + 2] = 1; }
    layer.WeightAlign();
    layer.Forward(sb, st);
  }
  sep.Reshape(t2.shape());                                                 // (2) the [-1 0 1] row filter
  memcpy(sep.mutable_cpu_data(), t2.cpu_data(), sizeof(Dtype) * t2.count());
  sb[0] = &sep;
  cp.kernel_h = 1; cp.kernel_w = 3; cp.stride_h = 1; cp.stride_w = 2;
  {
    ConvolutionLayer<Dtype> layer(lp);
    layer.SetUp(sb, st);
    Dtype *w = layer.blobs()[0]->mutable_cpu_data();
    w[0] = -1; w[1] = 0; w[2] = 1;
    layer.WeightAlign();
    layer.Forward(sb, st);
  }
  double maxerr = 0;
  const bool shape = t1.shape() == t2.shape();
  if (shape)
    for (int k = 0; k < t1.count(); ++k) maxerr = std::fmax(maxerr, std::fabs((double)t1.cpu_data()[k] - (double)t2.cpu_data()[k]));
  const bool ok = shape && maxerr <= Tol<Dtype>::abs_near() * 10;           // (:586-588 EXPECT_NEAR 1e-4; two layers deep)
  printf("%-6s %-3s %-26s top %dx%dx%dx%d full vs separable max|diff|=%.2e %s\n", Tol<Dtype>::name(), brew_name(),
         "TestSobelConvolution", t1.num(), t1.channels(), t1.height(), t1.width(), maxerr, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// The Forward wrapper reshapes on every call (layer.hpp:436): a net that was WeightAlign'ed and is
// then fed a larger batch, or another H x W, must keep producing the right numbers; so must a
// `-conv_mode` flip after the weights were loaded, and a layer made through the registry.
template <typename Dtype>
static int reshape_case() {
  LayerParameter lp;
  lp.name = "reshape";
  lp.type = "Convolution";
  lp.convolution_param.num_output = 12; lp.convolution_param.kernel_h = lp.convolution_param.kernel_w = 3;
  lp.convolution_param.pad_h = lp.convolution_param.pad_w = 1;
  const ConvolutionParameter cp = lp.convolution_param;
  shared_ptr<Layer<Dtype> > layer = LayerRegistry<Dtype>::CreateLayer(lp);   // layer_factory.cpp:74
  Blob<Dtype> b0(2, 8, 10, 10), t0;
  std::vector<Blob<Dtype> *> bottom(1, &b0), top(1, &t0);
  layer->SetUp(bottom, top);
  std::vector<Dtype> w(layer->blobs()[0]->count()), bias(cp.num_output);
  for (size_t k = 0; k < w.size(); ++k) { const float v = frand(); w[k] = (std::fabs(frand()) < 0.7f) ? 0.f : (v == 0.f ? 0.25f : v); }
  for (auto &v : bias) v = 0.1f * frand();
  memcpy(layer->blobs()[0]->mutable_cpu_data(), w.data(), sizeof(Dtype) * w.size());
  memcpy(layer->blobs()[1]->mutable_cpu_data(), bias.data(), sizeof(Dtype) * bias.size());
  layer->WeightAlign();
  int bad = 0;
  struct Step { int n, h, wd; Caffe::ConvMode mode; const char *what; };
  const Step steps[] = {{2, 10, 10, Caffe::SCONV_PAR, "as aligned"},      {5, 10, 10, Caffe::SCONV_PAR, "batch grew"},
                        {3, 14, 9, Caffe::SCONV_PAR, "H x W changed"},    {3, 14, 9, Caffe::LOWERED_GEMM, "-conv_mode 0"},
                        {3, 14, 9, Caffe::LOWERED_SPARSE, "-conv_mode 1"}, {1, 14, 9, Caffe::SCONV, "-conv_mode 2, batch shrank"}};
  for (const Step &st : steps) {
    Caffe::set_conv_mode(st.mode);
    b0.Reshape(st.n, 8, st.h, st.wd);
    Dtype *x = b0.mutable_cpu_data();
    for (int k = 0; k < b0.count(); ++k) x[k] = frand();
    layer->Forward(bottom, top);
    std::vector<double> want;
    naive_conv<Dtype>(b0, cp, w.data(), bias.data(), false, &want, t0.height(), t0.width());
    const double rel = compare<Dtype>(t0, want, nullptr);
    const bool ok = rel <= Tol<Dtype>::rel() && t0.num() == st.n && t0.height() == st.h && t0.width() == st.wd;
    printf("%-6s %-3s %-26s %-28s top %dx%dx%dx%d rel_err=%.2e %s\n", Tol<Dtype>::name(), brew_name(), "reshape_after_align",
           st.what, t0.num(), t0.channels(), t0.height(), t0.width(), rel, ok ? "OK" : "FAIL");
    bad += ok ? 0 : 1;
  }
  Caffe::set_conv_mode(Caffe::SCONV_PAR);
  return bad;
}

// WeightAlign in CPU mode, Forward in CPU mode, then the net is switched to the GPU (`caffe test -gpu 0` after a CPU
// load): the device side is aligned on the spot; both modes must give the same layer output.
template <typename Dtype>
static int mode_switch_case() {
  LayerParameter lp;
  lp.type = "Convolution";
  lp.convolution_param.num_output = 16; lp.convolution_param.kernel_h = lp.convolution_param.kernel_w = 3;
  lp.convolution_param.pad_h = lp.convolution_param.pad_w = 1;
  const ConvolutionParameter cp = lp.convolution_param;
  Blob<Dtype> b0(3, 8, 9, 11), t0;
  std::vector<Blob<Dtype> *> bottom(1, &b0), top(1, &t0);
  Dtype *x = b0.mutable_cpu_data();
  for (int k = 0; k < b0.count(); ++k) x[k] = frand();
  Caffe::set_mode(Caffe::CPU);
  ConvolutionLayer<Dtype> layer(lp);
  layer.SetUp(bottom, top);
  std::vector<Dtype> w(layer.blobs()[0]->count()), bias(cp.num_output);
  for (size_t k = 0; k < w.size(); ++k) { const float v = frand(); w[k] = (std::fabs(frand()) < 0.8f) ? 0.f : (v == 0.f ? 0.25f : v); }
  for (auto &v : bias) v = 0.1f * frand();
  memcpy(layer.blobs()[0]->mutable_cpu_data(), w.data(), sizeof(Dtype) * w.size());
  memcpy(layer.blobs()[1]->mutable_cpu_data(), bias.data(), sizeof(Dtype) * bias.size());
  layer.WeightAlign();
  layer.Forward(bottom, top);
  std::vector<Dtype> cpu_out(t0.cpu_data(), t0.cpu_data() + t0.count());
  Caffe::set_mode(Caffe::GPU);
  layer.Forward(bottom, top);
  std::vector<double> want;
  naive_conv<Dtype>(b0, cp, w.data(), bias.data(), false, &want, t0.height(), t0.width());
  const double rel = compare<Dtype>(t0, want, nullptr);
  double diff = 0;
  for (int k = 0; k < t0.count(); ++k) diff = std::fmax(diff, std::fabs((double)cpu_out[k] - (double)t0.cpu_data()[k]));
  const bool ok = rel <= Tol<Dtype>::rel() && diff <= Tol<Dtype>::abs_near();
  printf("%-6s %-26s CPU-aligned layer switched to GPU: rel_err=%.2e, max|cpu - gpu|=%.2e %s\n", Tol<Dtype>::name(),
         "mode_switch", rel, diff, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// A layer restored from the aligned form another layer exported (no dense blob, no WeightAlign): same numbers,
// bit for bit, and the persisted code object loaded as it was.  (GPU, float: the aligned form carries fp32 code.)
static int aligned_form_case() {
  LayerParameter lp;
  lp.name = "aligned";
  lp.type = "Convolution";
  lp.convolution_param.num_output = 64; lp.convolution_param.kernel_h = lp.convolution_param.kernel_w = 3;
  lp.convolution_param.pad_h = lp.convolution_param.pad_w = 1;
  lp.convolution_param.bias_term = false;
  Blob<float> b0(4, 32, 14, 14), t0, t1;
  std::vector<Blob<float> *> bottom(1, &b0), top0(1, &t0), top1(1, &t1);
  float *x = b0.mutable_cpu_data();
  for (int k = 0; k < b0.count(); ++k) x[k] = frand();
  ConvolutionLayer<float> a(lp), b(lp);
  a.SetUp(bottom, top0);
  b.SetUp(bottom, top1);
  float *w = a.blobs()[0]->mutable_cpu_data();
  for (int k = 0; k < a.blobs()[0]->count(); ++k) { const float v = frand(); w[k] = (std::fabs(frand()) < 0.9f) ? 0.f : (v == 0.f ? 0.25f : v); }
  a.WeightAlign();
  a.Forward(bottom, top0);
  const std::vector<unsigned char> blob = a.ExportAligned();
  const bool fast = b.WeightAlignFrom(blob);        // b's blobs_[0] was never filled
  b.Forward(bottom, top1);
  const bool same = memcmp(t0.cpu_data(), t1.cpu_data(), sizeof(float) * t0.count()) == 0;
  const bool code = strstr(a.kernel_name(), "jit") != nullptr;
  const bool ok = same && b.nnz() == a.nnz() && (fast || !code);
  printf("%-22s %-28s blob %zu bytes, code object loaded as persisted: %s, outputs identical: %s %s\n", "aligned_form", a.kernel_name(),
         blob.size(), fast ? "yes" : "no", same ? "yes" : "no", ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

static ConvolutionParameter P(int m, int k, int pad = 0, int stride = 1, int group = 1, bool bias = true,
                              int dil = 1) {
  ConvolutionParameter cp;
  cp.num_output = m; cp.kernel_h = cp.kernel_w = k; cp.pad_h = cp.pad_w = pad;
  cp.stride_h = cp.stride_w = stride; cp.group = group; cp.bias_term = bias; cp.dilation = dil;
  return cp;
}

// One {Dtype, Brew} combination of the reference's TestDtypesAndDevices list.
template <typename Dtype>
static int run_combination(Caffe::Brew brew) {
  Caffe::set_mode(brew);
  Caffe::set_conv_mode(Caffe::SCONV_PAR);      // tools/caffe.cpp:292-301 (-conv_mode 3)
  typedef ConvolutionLayer<Dtype> Conv;
  int bad = 0;
  // the forward cases of the reference's typed conv tests, filled as there (sparsity -1) ...
  bad += run_case<Dtype, Conv>("TestSimpleConvolution", 2, 3, 6, 4, P(4, 3, 0, 2), -1.f, false, 2);         // :231-265
  bad += run_case<Dtype, Conv>("TestDilatedConvolution", 2, 3, 8, 7, P(4, 3, 0, 1, 1, true, 2), -1.f, false, 2);   // :267-309
  bad += run_case<Dtype, Conv>("Test1x1Convolution", 2, 3, 6, 4, P(4, 1), -1.f, false);                     // :443-468
  bad += run_case<Dtype, Conv>("TestSimpleConvolutionGroup", 2, 3, 6, 4, P(3, 3, 0, 2, 3), -1.f, false);    // :470-496
  bad += sobel_case<Dtype>();                                                                               // :498-589
  // ... and again with pruned weights, which is what the path is for
  bad += run_case<Dtype, Conv>("TestSimpleConvolution/pruned", 2, 3, 6, 4, P(4, 3, 0, 2), 0.3f, false, 2);
  bad += run_case<Dtype, Conv>("TestDilatedConvolution/pruned", 2, 3, 8, 7, P(4, 3, 0, 1, 1, true, 2), 0.3f, false, 2);
  bad += run_case<Dtype, Conv>("Test1x1Convolution/pruned", 2, 3, 6, 4, P(4, 1), 0.25f, false);
  bad += run_case<Dtype, Conv>("TestConvolutionGroup/pruned", 2, 3, 6, 4, P(3, 3, 0, 2, 3), 0.3f, false);
  // BASELINE.json config shapes
  bad += run_case<Dtype, Conv>("lenet_conv2", 4, 20, 12, 12, P(50, 5), 0.5f, false);
  bad += run_case<Dtype, Conv>("alex_conv2", 2, 96, 27, 27, P(256, 5, 2, 1, 2), 0.8f, false);
  bad += run_case<Dtype, Conv>("res4_branch2b", 3, 256, 14, 14, P(256, 3, 1, 1, 1, false), 0.9f, false);
  bad += run_case<Dtype, Conv>("res2_branch2b", 2, 64, 56, 56, P(64, 3, 1, 1, 1, false), 0.9f, false);
  Caffe::set_conv_mode(Caffe::SCONV);          // -conv_mode 2: same numbers
  bad += run_case<Dtype, Conv>("res5_branch2b_sconv", 2, 512, 7, 7, P(512, 3, 1, 1, 1, false), 0.9f, false);
  bad += run_case<Dtype, ConvolutionReLULayer<Dtype> >("conv_relu_k3p1", 2, 16, 13, 13, P(24, 3, 1), 0.8f, true);
  Caffe::set_conv_mode(Caffe::SCONV_PAR);
  bad += reshape_case<Dtype>();
  return bad;
}

int main(int argc, char **argv) {
  const bool cpu_only = argc > 1 && !strcmp(argv[1], "--cpu-only");
  int bad = 0;
  bad += run_combination<float>(Caffe::CPU);
  bad += run_combination<double>(Caffe::CPU);
  if (!cpu_only) {
    Caffe::SetDevice(0);
    bad += run_combination<float>(Caffe::GPU);
    bad += run_combination<double>(Caffe::GPU);
    bad += mode_switch_case<float>();
    bad += mode_switch_case<double>();
    Caffe::set_mode(Caffe::GPU);
    bad += aligned_form_case();
  }
  printf(bad ? "shim self-test: %d FAILED\n" : "shim self-test: all OK\n", bad);
  return bad;
}
