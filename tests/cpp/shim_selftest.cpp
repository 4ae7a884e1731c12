// tests/cpp/shim_selftest.cpp -- drives the C++ Caffe-compatible shim the way the reference's own
// conv tests drive ConvolutionLayer (src/caffe/test/test_convolution_layer.cpp: build a
// LayerParameter, SetUp, fill blobs_, Forward, compare with an explicit loop-nest convolution),
// plus the step the reference's tests never take: WeightAlign() in SCONV mode.
// Needs a GPU; run by tests/test_shim_gpu.py.  Prints one line per case, exit code = #failures.
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

// explicit reference convolution in the style of caffe_conv() (test_convolution_layer.cpp:19-140)
static void naive_conv(const Blob<float> &in, const ConvolutionParameter &cp, const std::vector<float> &w,
                       const std::vector<float> &bias, bool relu, std::vector<double> *out, int oh, int ow) {
  const int N = in.num(), C = in.channels(), H = in.height(), W = in.width();
  const int M = cp.num_output, G = cp.group, Cg = C / G, Mg = M / G;
  const float *x = in.cpu_data();
  out->assign((size_t)N * M * oh * ow, 0.0);
  for (int n = 0; n < N; ++n)
    for (int m = 0; m < M; ++m) {
      const int g = m / Mg;
      for (int y = 0; y < oh; ++y)
        for (int xo = 0; xo < ow; ++xo) {
          double s = cp.bias_term ? bias[m] : 0.0;
          for (int c = 0; c < Cg; ++c)
            for (int kr = 0; kr < cp.kernel_h; ++kr)
              for (int kc = 0; kc < cp.kernel_w; ++kc) {
                const int iy = y * cp.stride_h - cp.pad_h + kr * cp.dilation;
                const int ix = xo * cp.stride_w - cp.pad_w + kc * cp.dilation;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                s += (double)w[(((size_t)m * Cg + c) * cp.kernel_h + kr) * cp.kernel_w + kc] *
                     x[(((size_t)n * C + g * Cg + c) * H + iy) * W + ix];
              }
          if (relu && s < 0) s = 0;
          (*out)[(((size_t)n * M + m) * oh + y) * ow + xo] = s;
        }
    }
}

template <class LayerT>
static int run_case(const char *name, int N, int C, int H, int W, ConvolutionParameter cp, float sparsity,
                    bool relu, int nbottom = 1) {
  LayerParameter lp;
  lp.name = name;
  lp.type = relu ? "ConvolutionReLU" : "Convolution";
  lp.convolution_param = cp;
  std::vector<Blob<float> *> bottom, top;
  std::vector<shared_ptr<Blob<float> > > keep;
  for (int i = 0; i < nbottom; ++i) {
    keep.push_back(shared_ptr<Blob<float> >(new Blob<float>(N, C, H, W)));
    bottom.push_back(keep.back().get());
    keep.push_back(shared_ptr<Blob<float> >(new Blob<float>()));
    top.push_back(keep.back().get());
    float *x = bottom[i]->mutable_cpu_data();
    for (int k = 0; k < bottom[i]->count(); ++k) x[k] = frand();
  }
  LayerT layer(lp);
  layer.SetUp(bottom, top);
  // "CopyTrainedLayersFrom": pruned weights into blobs_[0] (exact zeros = pruned), bias into blobs_[1]
  std::vector<float> w(layer.blobs()[0]->count()), bias(cp.num_output, 0.f);
  for (size_t k = 0; k < w.size(); ++k) {
    const float v = frand();
    w[k] = (std::fabs(frand()) < sparsity) ? 0.f : (v == 0.f ? 0.25f : v);
  }
  memcpy(layer.blobs()[0]->mutable_cpu_data(), w.data(), sizeof(float) * w.size());
  if (cp.bias_term) {
    for (int k = 0; k < cp.num_output; ++k) bias[k] = 0.1f * frand();
    memcpy(layer.blobs()[1]->mutable_cpu_data(), bias.data(), sizeof(float) * bias.size());
  }
  layer.WeightAlign();                     // net.cpp:819
  layer.Forward(bottom, top);              // net.cpp:568 -> layer.hpp:435
  double worst = 0;
  for (int i = 0; i < nbottom; ++i) {
    std::vector<double> want;
    naive_conv(*bottom[i], cp, w, bias, relu, &want, top[i]->height(), top[i]->width());
    const float *got = top[i]->cpu_data();
    double maxerr = 0, maxref = 0;
    for (size_t k = 0; k < want.size(); ++k) {
      maxerr = std::fmax(maxerr, std::fabs(got[k] - want[k]));
      maxref = std::fmax(maxref, std::fabs(want[k]));
    }
    worst = std::fmax(worst, maxerr / std::fmax(1e-6, maxref));
  }
  const bool ok = worst <= 1e-4;
  printf("%-22s %-8s top %dx%dx%dx%d nnz=%ld %-36s %.1f us rel_err=%.2e %s\n", name, layer.type(),
         top[0]->num(), top[0]->channels(), top[0]->height(), top[0]->width(), layer.nnz(),
         layer.kernel_name(), layer.get_time(), worst, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// The Forward wrapper reshapes on every call (layer.hpp:436): a net that was WeightAlign'ed and is
// then fed a larger batch, or another H x W, must keep producing the right numbers; so must a
// `-conv_mode` flip after the weights were loaded, and a layer made through the registry.
static int reshape_case() {
  LayerParameter lp;
  lp.name = "reshape";
  lp.type = "Convolution";
  lp.convolution_param.num_output = 12; lp.convolution_param.kernel_h = lp.convolution_param.kernel_w = 3;
  lp.convolution_param.pad_h = lp.convolution_param.pad_w = 1;
  const ConvolutionParameter cp = lp.convolution_param;
  shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);   // layer_factory.cpp:74
  Blob<float> b0(2, 8, 10, 10), t0;
  std::vector<Blob<float> *> bottom(1, &b0), top(1, &t0);
  layer->SetUp(bottom, top);
  std::vector<float> w(layer->blobs()[0]->count()), bias(cp.num_output);
  for (size_t k = 0; k < w.size(); ++k) { const float v = frand(); w[k] = (std::fabs(frand()) < 0.7f) ? 0.f : (v == 0.f ? 0.25f : v); }
  for (auto &v : bias) v = 0.1f * frand();
  memcpy(layer->blobs()[0]->mutable_cpu_data(), w.data(), sizeof(float) * w.size());
  memcpy(layer->blobs()[1]->mutable_cpu_data(), bias.data(), sizeof(float) * bias.size());
  layer->WeightAlign();
  int bad = 0;
  struct Step { int n, h, wd; Caffe::ConvMode mode; const char *what; };
  const Step steps[] = {{2, 10, 10, Caffe::SCONV_PAR, "as aligned"},      {5, 10, 10, Caffe::SCONV_PAR, "batch grew"},
                        {3, 14, 9, Caffe::SCONV_PAR, "H x W changed"},    {3, 14, 9, Caffe::LOWERED_GEMM, "-conv_mode 0"},
                        {3, 14, 9, Caffe::LOWERED_SPARSE, "-conv_mode 1"}, {1, 14, 9, Caffe::SCONV, "-conv_mode 2, batch shrank"}};
  for (const Step &st : steps) {
    Caffe::set_conv_mode(st.mode);
    b0.Reshape(st.n, 8, st.h, st.wd);
    float *x = b0.mutable_cpu_data();
    for (int k = 0; k < b0.count(); ++k) x[k] = frand();
    layer->Forward(bottom, top);
    std::vector<double> want;
    naive_conv(b0, cp, w, bias, false, &want, t0.height(), t0.width());
    const float *got = t0.cpu_data();
    double maxerr = 0, maxref = 0;
    for (size_t k = 0; k < want.size(); ++k) {
      maxerr = std::fmax(maxerr, std::fabs(got[k] - want[k]));
      maxref = std::fmax(maxref, std::fabs(want[k]));
    }
    const double rel = maxerr / std::fmax(1e-6, maxref);
    const bool ok = rel <= 1e-4 && t0.num() == st.n && t0.height() == st.h && t0.width() == st.wd;
    printf("%-22s %-28s top %dx%dx%dx%d rel_err=%.2e %s\n", "reshape_after_align", st.what, t0.num(), t0.channels(),
           t0.height(), t0.width(), rel, ok ? "OK" : "FAIL");
    bad += ok ? 0 : 1;
  }
  Caffe::set_conv_mode(Caffe::SCONV_PAR);
  return bad;
}

// A layer restored from the aligned form another layer exported (no dense blob, no WeightAlign): same numbers,
// bit for bit, and the persisted code object loaded as it was.
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

int main() {
  Caffe::SetDevice(0);
  Caffe::set_mode(Caffe::GPU);
  Caffe::set_conv_mode(Caffe::SCONV_PAR);      // tools/caffe.cpp:292-301 (-conv_mode 3)
  int bad = 0;
  // the reference's own conv test shapes (test_convolution_layer.cpp)
  bad += run_case<ConvolutionLayer<float> >("TestSimpleConvolution", 2, 3, 6, 4, P(4, 3, 0, 2), 0.3f, false, 2);
  bad += run_case<ConvolutionLayer<float> >("TestDilatedConvolution", 2, 3, 8, 7, P(4, 3, 0, 1, 1, true, 2), 0.3f, false);
  bad += run_case<ConvolutionLayer<float> >("Test1x1Convolution", 2, 3, 6, 4, P(4, 1), 0.25f, false);
  bad += run_case<ConvolutionLayer<float> >("TestConvolutionGroup", 2, 3, 6, 4, P(3, 3, 0, 1, 3), 0.3f, false);
  // BASELINE.json config shapes
  bad += run_case<ConvolutionLayer<float> >("lenet_conv2", 4, 20, 12, 12, P(50, 5), 0.5f, false);
  bad += run_case<ConvolutionLayer<float> >("alex_conv2", 2, 96, 27, 27, P(256, 5, 2, 1, 2), 0.8f, false);
  bad += run_case<ConvolutionLayer<float> >("res4_branch2b", 3, 256, 14, 14, P(256, 3, 1, 1, 1, false), 0.9f, false);
  bad += run_case<ConvolutionLayer<float> >("res2_branch2b", 2, 64, 56, 56, P(64, 3, 1, 1, 1, false), 0.9f, false);
  Caffe::set_conv_mode(Caffe::SCONV);          // -conv_mode 2: same numbers
  bad += run_case<ConvolutionLayer<float> >("res5_branch2b_sconv", 2, 512, 7, 7, P(512, 3, 1, 1, 1, false), 0.9f, false);
  bad += run_case<ConvolutionReLULayer<float> >("conv_relu_k3p1", 2, 16, 13, 13, P(24, 3, 1), 0.8f, true);
  Caffe::set_conv_mode(Caffe::SCONV_PAR);
  bad += reshape_case();
  bad += aligned_form_case();
  printf(bad ? "shim self-test: %d FAILED\n" : "shim self-test: all OK\n", bad);
  return bad;
}
