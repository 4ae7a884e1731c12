// sconv_cpu.cpp -- Caffe::CPU mode of the library: the host entry points of include/escoin.h
// (escoin_weight_align_cpu, escoin_forward_cpu, escoin_cpu_sconv, escoin_cpu_sparse_dense2csr and their _f64 twins).
//
//   ConvolutionLayer<Dtype>::Forward_cpu            conv_layer.cpp:25-63       batch loop, bias after
//     BaseConvolutionLayer::forward_cpu_sconv       base_conv_layer.cpp:569-661 pad copy, per-group kernel call
//       caffe_cpu_sconv<Dtype>                      math_functions.cpp:128-176
//     forward_cpu_bias                              base_conv_layer.cpp:663-669
//   WeightAlign (CPU branch)                        base_conv_layer.cpp:46-107  dense -> CSR -> stretched indices
//
// A product-side implementation written for this library (kernel: sconv_cpu_kernel.cpp); it works on a machine with
// no HIP device and neither includes nor loads anything under oracle/ (tests/ compare the two from outside).
//
// Threads: the reference's g++ build runs the batch loop serially and its ICC build parallelises it over images with
// per-thread padded buffers (conv_layer.cpp:41-44, base_conv_layer.cpp:72-75,605-608).  Here a team of n_threads
// threads (the caller + a process-wide pool of parked std::threads) takes (image, slice of output channels) items from
// a shared counter, every thread with its own padded buffer; when the batch has fewer images than threads the output
// channels of an image are split instead.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "escoin_plan.h"
#include "sconv_cpu.h"
#include "thread_place.h"

namespace escoin {
namespace cpu {

enum Isa { kNone = 0, kAvx2 = 2, kAvx512 = 512 };

static Isa detect_isa() {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq"))
    return kAvx512;
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return kAvx2;
  return kNone;
}

static std::atomic<int> g_isa_override{-1};      // escoin_cpu_kernel_select: -1 = what the CPU reports

static Isa isa() {
  static const Isa v = detect_isa();
  const int o = g_isa_override.load(std::memory_order_relaxed);
  return o < 0 ? v : (Isa)o;
}

template <typename T>
static void run_group(const GroupJob<T> &job) {
  if (isa() == kAvx512)
    run_group_avx512<T>(job);
  else
    run_group_avx2<T>(job);
}

// The host threads of the CPU mode: one pool per process, grown on demand, its threads parked on a condition variable
// between calls and placed on cores of their own when they start (place_on_own_core below).  (Starting fresh
// std::threads per call was measured first: a new thread starts on its parent's core and the scheduler spread the team
// one thread per 4 ms tick -- a 40 ms forward then ran on little more than one core.)  One team job runs at a time; a
// second host thread calling in waits for the pool (plans are thread-compatible, the pool is thread-safe).
class Pool {
 public:
  static Pool &get() {
    static Pool pool;
    return pool;
  }
  // Runs fn(tid, item) for item = 0 .. n_items-1 on n_threads threads (the caller is thread 0).  The first exception
  // any thread throws stops the hand-out and is rethrown on the caller after the job has drained: nothing crosses the
  // C ABI (the entry points run inside guarded()), nothing calls std::terminate.
  template <class F>
  void run(int n_threads, int n_items, F &&fn) {
    if (n_items < 1) return;
    if (n_threads > n_items) n_threads = n_items;
    if (n_threads < 1) n_threads = 1;
    if (n_threads == 1) {            // the caller alone: no pool, no lock -- host threads with one-thread calls run side by side
      for (int item = 0; item < n_items; ++item) fn(0, item);
      return;
    }
    std::lock_guard<std::mutex> one_job(job_mu_);
    Job job;
    job.n_items = n_items;
    job.call = [&fn](int tid, int item) { fn(tid, item); };
    if (n_threads > 1) {
      std::unique_lock<std::mutex> lk(mu_);
      try {
        while ((int)workers_.size() < n_threads - 1) {
          const int id = (int)workers_.size();
          workers_.emplace_back([this, id] { worker_main(id); });
        }
      } catch (...) {   // thread creation failed: the threads that exist share the items with the caller
        if (getenv("ESCOIN_VERBOSE"))
          fprintf(stderr, "[escoin] cpu pool: only %zu of %d threads could be started\n", workers_.size() + 1, n_threads);
      }
      job.helpers = std::min((int)workers_.size(), n_threads - 1);
      job.pending = job.helpers;
      job_ = &job;
      ++generation_;
      lk.unlock();
      wake_.notify_all();
    }
    work(job, 0);
    if (job.helpers > 0) {
      std::unique_lock<std::mutex> lk(mu_);
      done_.wait(lk, [&] { return job.pending == 0; });
      job_ = nullptr;
    }
    if (job.first) std::rethrow_exception(job.first);
  }

 private:
  struct Job {
    std::function<void(int, int)> call;
    std::atomic<int> next{0};
    std::atomic<bool> stop{false};
    int n_items = 0, helpers = 0, pending = 0;
    std::exception_ptr first;
    std::mutex first_mu;
  };
  static void work(Job &job, int tid) {
    try {
      for (;;) {
        if (job.stop.load(std::memory_order_relaxed)) return;
        const int item = job.next.fetch_add(1);
        if (item >= job.n_items) return;
        job.call(tid, item);
      }
    } catch (...) {
      job.stop.store(true);
      std::lock_guard<std::mutex> lk(job.first_mu);
      if (!job.first) job.first = std::current_exception();
    }
  }
  // (a worker is placed on a core of its own when it starts: thread_place.h; the caller keeps the mask's first core)
  void worker_main(int id) {
    place_on_own_core(id + 1);
    unsigned long seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      wake_.wait(lk, [&] { return quit_ || (generation_ != seen && job_ != nullptr); });
      if (quit_) return;
      seen = generation_;
      Job *job = job_;
      if (id >= job->helpers) continue;      // this job runs on fewer threads than the pool has
      lk.unlock();
      work(*job, id + 1);
      lk.lock();
      if (--job->pending == 0) done_.notify_all();
    }
  }
  Pool() {}
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    wake_.notify_all();
    for (auto &t : workers_)
      if (t.joinable()) t.join();
  }
  std::mutex job_mu_, mu_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> workers_;
  Job *job_ = nullptr;
  unsigned long generation_ = 0;
  bool quit_ = false;
};

template <class F>
static void team(int n_threads, int n_items, F &&body) {
  Pool::get().run(n_threads, n_items, body);
}

static int resolve_threads(int n_threads) {
  if (n_threads > 0) return n_threads;
  const unsigned hc = std::thread::hardware_concurrency();
  return hc ? (int)hc : 1;
}

// base_conv_layer.cpp:71 (+ the pad_w floats the reference forgets when pad_h == 0 < pad_w, see escoin_padded_len)
static size_t padded_len(const Geometry &g) {
  const escoin_conv_desc &d = g.d;
  return (size_t)d.C * (d.H + d.pad_h) * (d.W + d.pad_w) + (size_t)d.pad_h * (d.W + 2 * d.pad_w) +
         (size_t)(d.pad_h == 0 ? d.pad_w : 0);
}

template <typename T> static const std::vector<std::vector<T>> &plan_values(const escoin_plan *p);
template <> const std::vector<std::vector<float>> &plan_values<float>(const escoin_plan *p) { return p->values; }
template <> const std::vector<std::vector<double>> &plan_values<double>(const escoin_plan *p) { return p->values64; }

static void build_offsets(escoin_plan *p) {
  const Geometry &g = p->g;
  const int PH = g.d.H + g.d.pad_h, PW = g.d.W + g.d.pad_w;
  p->cpu_off.assign(g.d.group, std::vector<int>());
  for (int grp = 0; grp < g.d.group; ++grp) {
    const std::vector<int> &ci = p->colidx[grp];
    std::vector<int> &off = p->cpu_off[grp];
    off.resize(ci.size());
    for (size_t j = 0; j < ci.size(); ++j) {
      // the stretch of base_conv_layer.cpp:96-107 with the dilation folded in (the reference decodes it again per
      // multiply-add in its dilated branch, math_functions.cpp:142-160)
      const int col = ci[j];
      const int kc = col % g.d.KW, kr = (col / g.d.KW) % g.d.KH, ic = col / (g.d.KW * g.d.KH);
      off[j] = (ic * PH + kr * g.d.dil_h) * PW + kc * g.d.dil_w;
    }
  }
  p->cpu_off_valid = true;
  p->cpu_blk_cb = -1;
}

// The channel blocks of a stride-1 plan for the active flavour and element type (sconv_cpu.h): where every row's nonzeros
// cross from one block of input channels to the next.  Built once per plan (again if the flavour is switched).
template <typename T>
static void build_channel_blocks(escoin_plan *p) {
  const Geometry &g = p->g;
  const escoin_conv_desc &d = g.d;
  const int PW = d.W + d.pad_w;
  long nnz = 0;
  for (int grp = 0; grp < d.group; ++grp) nnz += (long)p->colidx[grp].size();
  const double avg_row = (double)nnz / std::max(1, d.group * g.Mg);
  int cb = 0;
  if (d.stride_h == 1 && d.stride_w == 1)
    cb = isa() == kAvx512 ? channel_block_avx512<T>(g.OH, g.OW, PW, (d.KH - 1) * d.dil_h, g.Cg, avg_row)
                          : channel_block_avx2<T>(g.OH, g.OW, PW, (d.KH - 1) * d.dil_h, g.Cg, avg_row);
  if (p->cpu_blk_force > 0 && d.stride_h == 1 && d.stride_w == 1) cb = p->cpu_blk_force >= g.Cg ? 0 : p->cpu_blk_force;
  p->cpu_blk_isa = (int)isa();
  p->cpu_blk_elem = (int)sizeof(T);
  p->cpu_blk_cb = cb;
  p->cpu_blk_n = cb > 0 ? (g.Cg + cb - 1) / cb : 0;
  p->cpu_blk.assign(d.group, std::vector<int>());
  if (cb <= 0) return;
  const int nb = p->cpu_blk_n, taps = d.KH * d.KW;
  for (int grp = 0; grp < d.group; ++grp) {
    const std::vector<int> &rp = p->rowptr[grp], &ci = p->colidx[grp];
    std::vector<int> &blk = p->cpu_blk[grp];
    blk.resize((size_t)g.Mg * (nb + 1));
    for (int m = 0; m < g.Mg; ++m) {
      int j = rp[m];
      for (int b = 0; b < nb; ++b) {
        blk[(size_t)m * (nb + 1) + b] = j;
        const int ic_end = (b + 1) * cb;                          // (ascending columns: ascending channels)
        while (j < rp[m + 1] && ci[j] / taps < ic_end) ++j;
      }
      blk[(size_t)m * (nb + 1) + nb] = rp[m + 1];
    }
  }
}

template <typename T>
static int forward_cpu(escoin_plan *p, const T *bottom, const T *bias, T *top, int n_images, int n_threads) {
  if (!p || !bottom || !top) return fail(ESCOIN_EINVAL, "null argument");
  if (!p->host_aligned) return fail(ESCOIN_ESTATE, "forward_cpu called before weight_align / set_csr");
  if (p->is_f64 != (sizeof(T) == 8))
    return fail(ESCOIN_ESTATE, p->is_f64 ? "forward_cpu: the plan holds double weights (use the _f64 entry point)"
                                         : "forward_cpu_f64: the plan holds float weights");
  if (n_images < 0) return fail(ESCOIN_EINVAL, "n_images must be >= 0");
  if (n_images == 0) return ESCOIN_OK;
  if (isa() == kNone) return fail(ESCOIN_ENODEVICE, "the CPU path needs AVX2 + FMA (the reference builds with -mavx2 -mfma too)");
  if (!p->cpu_off_valid) build_offsets(p);
  if (p->cpu_blk_cb < 0 || p->cpu_blk_isa != (int)isa() || p->cpu_blk_elem != (int)sizeof(T)) build_channel_blocks<T>(p);
  const Geometry &g = p->g;
  const escoin_conv_desc &d = g.d;
  const int PH = d.H + d.pad_h, PW = d.W + d.pad_w;
  const bool padded = d.pad_h != 0 || d.pad_w != 0;          // base_conv_layer.cpp:601
  const size_t plen = padded_len(g);
  const size_t bottom_dim = (size_t)d.C * d.H * d.W, top_dim = (size_t)d.M * g.OH * g.OW;
  const auto &values = plan_values<T>(p);
  n_threads = resolve_threads(n_threads);
  const unsigned long call_id = ++p->cpu_calls;   // the same blob pointer in a later call holds other data
  // small images travel two or three to a job (sconv_cpu.h GroupJob::n_img) -- as long as every thread still gets one
  int ni = isa() == kAvx512 ? images_per_job_avx512<T>(g.OH, g.OW, PW, d.stride_h, d.stride_w)
                            : images_per_job_avx2<T>(g.OH, g.OW, PW, d.stride_h, d.stride_w);
  {
    // ... and only where their windows stay close to the core: with channel blocks they do by construction; without
    // (the 95 %-sparse pointwise layers: few nonzeros per row, nothing to block for) three images' planes of every
    // channel are three times the L2 traffic (GoogLeNet's 7 x 7 layers measured -13..-27 % with three images a job)
    const bool will_block = p->cpu_blk_cb > 0 && p->cpu_blk_n > 1;
    const long one = isa() == kAvx512 ? window_bytes_avx512<T>(g.OH, g.OW, PW, (d.KH - 1) * d.dil_h)
                                      : window_bytes_avx2<T>(g.OH, g.OW, PW, (d.KH - 1) * d.dil_h);
    constexpr long kNearBytes = 192 * 1024;
    while (!will_block && ni > 1 && one * g.Cg * ni > kNearBytes) --ni;
  }
  if (p->cpu_img_force > 0) ni = std::min(ni, p->cpu_img_force);
  ni = std::max(1, std::min(ni, n_images / std::max(1, n_threads)));
  p->cpu_img_last = ni;
  const int n_jobs = (n_images + ni - 1) / ni;
  // fewer jobs than threads: split every job's output channels too
  int parts = 1;
  if (n_jobs < n_threads) parts = std::min(g.Mg, (n_threads + n_jobs - 1) / n_jobs);
  const int n_items = n_jobs * parts;
  const int team_size = std::min(n_threads, n_items);
  constexpr size_t kSlack = 16;   // one vector behind the image: the kernel loads whole vectors at the image's end
  // per-thread padded buffer + store scratch, kept in the plan between calls (a plan belongs to one host thread at a
  // time, like a Caffe layer instance); zeroed when (re)allocated -- base_conv_layer.cpp:78-80 -- and only the interior
  // is ever rewritten
  if ((int)p->cpu_ws.size() < team_size) p->cpu_ws.resize((size_t)team_size);
  const size_t pad_elems = plen + kSlack;                          // one image's slot in a thread's padded buffer
  const size_t pad_bytes = padded ? pad_elems * (size_t)ni * sizeof(T) : 0;
  const size_t scratch_bytes = scratch_elems(g.OH, PW) * sizeof(T);
  const bool blocked = p->cpu_blk_cb > 0 && p->cpu_blk_n > 1;
  const size_t partial_bytes = blocked ? (size_t)((g.Mg + parts - 1) / parts + 1) * kPartialElemsPerRow * sizeof(T) : 0;
  team(team_size, n_items, [&](int tid, int item) {
    CpuWorkspace &L = p->cpu_ws[(size_t)tid];
    if (L.pad.size() != pad_bytes) {
      L.pad.assign(pad_bytes, 0);
      L.src = nullptr;
    }
    if (L.scratch.size() < scratch_bytes) L.scratch.assign(scratch_bytes, 0);
    if (L.partial.size() < partial_bytes) L.partial.assign(partial_bytes, 0);
    const int job = item / parts, part = item - job * parts;
    const int n = job * ni, n_here = std::min(ni, n_images - n);     // images n .. n + n_here - 1
    const T *image = bottom + (size_t)n * bottom_dim;
    const T *in_p = image;
    if (padded) {
      T *pad = reinterpret_cast<T *>(L.pad.data());
      if (L.src != (const void *)image || L.call != call_id) {     // (a thread pads its images once for all their channel slices)
        for (int i = 0; i < n_here; ++i)
          for (int c = 0; c < d.C; ++c)                           // base_conv_layer.cpp:615-620
            for (int r = 0; r < d.H; ++r)
              memcpy(pad + (size_t)i * pad_elems + ((size_t)c * PH + r + d.pad_h) * PW + d.pad_w,
                     image + (size_t)i * bottom_dim + ((size_t)c * d.H + r) * d.W, sizeof(T) * (size_t)d.W);
        L.src = image;
        L.call = call_id;
      }
      in_p = pad;
    }
    const int m0 = (int)((long)g.Mg * part / parts), m1 = (int)((long)g.Mg * (part + 1) / parts);
    for (int grp = 0; grp < d.group; ++grp) {                     // base_conv_layer.cpp:626-658
      GroupJob<T> J;
      J.in = in_p + (size_t)g.Cg * grp * PH * PW;                 // :633
      J.rowptr = p->rowptr[grp].data();
      J.off = p->cpu_off[grp].data();
      J.val = values[grp].data();
      J.bias = bias ? bias + (size_t)grp * g.Mg : nullptr;
      J.out = top + (size_t)n * top_dim + (size_t)grp * g.Mg * g.OH * g.OW;
      J.m_begin = m0;
      J.m_end = m1;
      J.OH = g.OH; J.OW = g.OW; J.PW = PW;
      J.stride_h = d.stride_h; J.stride_w = d.stride_w;
      J.relu = d.fuse_relu;
      J.exact_reads = padded ? 0 : 1;                             // an unpadded layer reads the caller's blob itself
      J.scratch = reinterpret_cast<T *>(L.scratch.data());
      J.blk_ptr = blocked ? p->cpu_blk[grp].data() : nullptr;
      J.n_blk = blocked ? p->cpu_blk_n : 0;
      J.partial = blocked ? reinterpret_cast<T *>(L.partial.data()) : nullptr;
      J.n_img = n_here;
      J.in_stride = padded ? pad_elems : bottom_dim;
      J.out_stride = top_dim;
      run_group<T>(J);
    }
  });
  return ESCOIN_OK;
}

// caffe_cpu_sconv<Dtype>, math_functions.cpp:128-176, on the caller's buffers (exactly the reference's lengths).
template <typename T>
static int cpu_sconv(const T *input_padded, int in_channels, int height, int width, int pad_h, int pad_w, int stride_h,
                     int stride_w, int dilation_h, int dilation_w, const int *rowptr, const int *colidx, const T *values,
                     int kernel_h, int kernel_w, T *output, int out_channels, int input_padded_len, int n_threads) {
  if (!input_padded || !rowptr || !output || in_channels < 1 || height < 1 || width < 1 || pad_h < 0 || pad_w < 0 ||
      stride_h < 1 || stride_w < 1 || dilation_h < 1 || dilation_w < 1 || kernel_h < 1 || kernel_w < 1 || out_channels < 1)
    return fail(ESCOIN_EINVAL, "escoin_cpu_sconv: bad argument");
  if (isa() == kNone) return fail(ESCOIN_ENODEVICE, "the CPU path needs AVX2 + FMA");
  const int OH = (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;   // :136-137
  const int OW = (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  if (OH < 1 || OW < 1) return fail(ESCOIN_EINVAL, "escoin_cpu_sconv: empty output");
  const int nnz = rowptr[out_channels];
  if (rowptr[0] != 0 || nnz < 0 || (nnz > 0 && (!colidx || !values)))
    return fail(ESCOIN_EINVAL, "escoin_cpu_sconv: bad CSR");
  const int PH = height + pad_h, PW = width + pad_w;
  // colidx is the stretched index (ic * PH + kr) * PW + kc of base_conv_layer.cpp:96-107; the dilated branch decodes
  // it (math_functions.cpp:142-160) -- folded into the offsets once here.  The largest element any output reads must
  // lie inside the caller's buffer (the reference asserts it per multiply-add, :168).
  std::vector<int> off((size_t)nnz);
  long max_off = -1;
  for (int j = 0; j < nnz; ++j) {
    const int col = colidx[j];
    if (col < 0) return fail(ESCOIN_EINVAL, "escoin_cpu_sconv: negative column index");
    int o = col;
    if (dilation_h != 1 || dilation_w != 1) {
      const int kc = col % PW, kr = (col / PW) % PH, ic = col / (PW * PH);
      o = (ic * PH + kr * dilation_h) * PW + kc * dilation_w;
    }
    off[(size_t)j] = o;
    if (o > max_off) max_off = o;
  }
  const long last = (long)(OH - 1) * stride_h * PW + (long)(OW - 1) * stride_w;
  if (nnz > 0 && input_padded_len > 0 && max_off + last >= (long)input_padded_len)
    return fail(ESCOIN_EINVAL, "escoin_cpu_sconv: a nonzero reads past input_padded_len");
  n_threads = std::min(resolve_threads(n_threads), out_channels);
  std::vector<std::vector<T>> scratch((size_t)n_threads);
  team(n_threads, n_threads, [&](int /*tid*/, int part) {
    std::vector<T> &sc = scratch[(size_t)part];
    sc.assign(scratch_elems(OH, PW), T(0));
    GroupJob<T> J;
    J.in = input_padded; J.rowptr = rowptr; J.off = off.data(); J.val = values; J.bias = nullptr; J.out = output;
    J.m_begin = (int)((long)out_channels * part / n_threads);
    J.m_end = (int)((long)out_channels * (part + 1) / n_threads);
    J.OH = OH; J.OW = OW; J.PW = PW; J.stride_h = stride_h; J.stride_w = stride_w; J.relu = 0; J.exact_reads = 1;
    J.scratch = sc.data();
    J.blk_ptr = nullptr; J.n_blk = 0; J.partial = nullptr;      // (a one-off call on the caller's CSR: no block table to amortise)
    J.n_img = 1; J.in_stride = 0; J.out_stride = 0;
    run_group<T>(J);
  });
  return ESCOIN_OK;
}

// caffe_cpu_sparse_dense2csr<Dtype>, the hand loop of math_functions.cpp:92-105 (0-based, ascending columns).
template <typename T>
static int cpu_dense2csr(int M, int N, const T *A, T *A_nonzero_buf, int *A_nonzero_idx_buf, int *A_idx_pointer_buf) {
  if (M < 1 || N < 1 || !A || !A_nonzero_buf || !A_nonzero_idx_buf || !A_idx_pointer_buf)
    return fail(ESCOIN_EINVAL, "escoin_cpu_sparse_dense2csr: bad argument");
  int nnz = 0;
  A_idx_pointer_buf[0] = 0;
  for (int i = 0; i < M; ++i) {
    const T *row = A + (size_t)i * N;
    for (int j = 0; j < N; ++j)
      if (row[j] != 0) {
        A_nonzero_buf[nnz] = row[j];
        A_nonzero_idx_buf[nnz] = j;
        ++nnz;
      }
    A_idx_pointer_buf[i + 1] = nnz;
  }
  return ESCOIN_OK;
}

template <typename T>
static int weight_align_cpu(escoin_plan *p, const T *dense_w) {
  if (!p || !dense_w) return fail(ESCOIN_EINVAL, "null argument");
  csr_from_dense<T>(p, dense_w);     // leaves host_aligned = true, the device side (if any) released
  return ESCOIN_OK;
}

}  // namespace cpu
}  // namespace escoin

using namespace escoin;

extern "C" {

const char *escoin_cpu_kernel_name(void) {
  switch (cpu::isa()) {
    case cpu::kAvx512: return "escoin_cpu_sconv_avx512";
    case cpu::kAvx2: return "escoin_cpu_sconv_avx2";
    default: return "(no AVX2 + FMA: CPU path unavailable)";
  }
}

int escoin_cpu_kernel_select(const char *which) {
  if (!which) return fail(ESCOIN_EINVAL, "null argument");
  cpu::g_isa_override.store(-1);
  const cpu::Isa best = cpu::isa();
  if (!strcmp(which, "auto")) return ESCOIN_OK;
  if (!strcmp(which, "avx2")) {
    if (best == cpu::kNone) return fail(ESCOIN_ENODEVICE, "this CPU has no AVX2 + FMA");
    cpu::g_isa_override.store((int)cpu::kAvx2);
    return ESCOIN_OK;
  }
  if (!strcmp(which, "avx512")) {
    if (best != cpu::kAvx512) return fail(ESCOIN_ENODEVICE, "this CPU has no AVX-512 (F, VL, DQ)");
    cpu::g_isa_override.store((int)cpu::kAvx512);
    return ESCOIN_OK;
  }
  return fail(ESCOIN_EINVAL, "escoin_cpu_kernel_select: \"auto\", \"avx2\" or \"avx512\"");
}

int escoin_weight_align_cpu(escoin_plan *p, const float *dense_w) {
  return guarded([&]() -> int { return cpu::weight_align_cpu<float>(p, dense_w); });
}
int escoin_weight_align_cpu_f64(escoin_plan *p, const double *dense_w) {
  return guarded([&]() -> int { return cpu::weight_align_cpu<double>(p, dense_w); });
}

int escoin_forward_cpu(escoin_plan *p, const float *bottom, const float *bias, float *top, int n_images, int n_threads) {
  return guarded([&]() -> int { return cpu::forward_cpu<float>(p, bottom, bias, top, n_images, n_threads); });
}
int escoin_forward_cpu_f64(escoin_plan *p, const double *bottom, const double *bias, double *top, int n_images,
                           int n_threads) {
  return guarded([&]() -> int { return cpu::forward_cpu<double>(p, bottom, bias, top, n_images, n_threads); });
}

int escoin_cpu_sconv(const float *input_padded, int in_channels, int height, int width, int pad_h, int pad_w,
                     int stride_h, int stride_w, int dilation_h, int dilation_w, const int *rowptr, const int *colidx,
                     const float *values, int kernel_h, int kernel_w, const float * /*bias: unused, as in the reference*/,
                     float *output, int out_channels, int input_padded_len) {
  return guarded([&]() -> int {
    return cpu::cpu_sconv<float>(input_padded, in_channels, height, width, pad_h, pad_w, stride_h, stride_w, dilation_h,
                                 dilation_w, rowptr, colidx, values, kernel_h, kernel_w, output, out_channels,
                                 input_padded_len, 1);
  });
}
int escoin_cpu_sconv_f64(const double *input_padded, int in_channels, int height, int width, int pad_h, int pad_w,
                         int stride_h, int stride_w, int dilation_h, int dilation_w, const int *rowptr,
                         const int *colidx, const double *values, int kernel_h, int kernel_w, const double * /*bias*/,
                         double *output, int out_channels, int input_padded_len) {
  return guarded([&]() -> int {
    return cpu::cpu_sconv<double>(input_padded, in_channels, height, width, pad_h, pad_w, stride_h, stride_w,
                                  dilation_h, dilation_w, rowptr, colidx, values, kernel_h, kernel_w, output,
                                  out_channels, input_padded_len, 1);
  });
}

int escoin_cpu_sparse_dense2csr(int M, int N, const float *A, float *A_nonzero_buf, int *A_nonzero_idx_buf,
                                int *A_idx_pointer_buf) {
  return cpu::cpu_dense2csr<float>(M, N, A, A_nonzero_buf, A_nonzero_idx_buf, A_idx_pointer_buf);
}
int escoin_cpu_sparse_dense2csr_f64(int M, int N, const double *A, double *A_nonzero_buf, int *A_nonzero_idx_buf,
                                    int *A_idx_pointer_buf) {
  return cpu::cpu_dense2csr<double>(M, N, A, A_nonzero_buf, A_nonzero_idx_buf, A_idx_pointer_buf);
}

}  // extern "C"
