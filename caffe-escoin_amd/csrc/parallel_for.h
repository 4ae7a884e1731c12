// parallel_for.h -- WeightAlign's helper threads (channel deal, code generation): fn(i) for i = 0 .. n-1 on up to
// n_threads threads, the caller among them, items handed out from a shared counter.
//   * a helper moves itself to a core of its own when it starts (thread_place.h);
//   * a thread the system refuses to create is simply one worker fewer;
//   * an exception thrown by fn on ANY thread (std::bad_alloc from a growing code vector) stops the hand-out, is kept
//     (the first one wins) and rethrown on the calling thread after every helper has been joined -- through an RAII
//     guard, so the helpers are joined on every path.  Nothing reaches std::terminate, nothing crosses the C ABI:
//     the entry points catch what arrives here (escoin_plan.h guarded()).
#ifndef ESCOIN_PARALLEL_FOR_H_
#define ESCOIN_PARALLEL_FOR_H_

#include <atomic>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "thread_place.h"

namespace escoin {

template <class F>
void parallel_for(size_t n_items, size_t n_threads, F &&fn) {
  if (n_items == 0) return;
  if (n_threads > n_items) n_threads = n_items;
  if (n_threads <= 1) {
    for (size_t i = 0; i < n_items; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::atomic<bool> stop{false};
  std::exception_ptr first;
  std::mutex first_mu;
  auto worker = [&](int slot) {
    try {
      if (slot > 0) place_on_own_core(slot);   // a helper starts on the caller's core otherwise (thread_place.h)
      for (size_t i = next.fetch_add(1); i < n_items && !stop.load(std::memory_order_relaxed); i = next.fetch_add(1)) fn(i);
    } catch (...) {
      stop.store(true);
      std::lock_guard<std::mutex> lk(first_mu);
      if (!first) first = std::current_exception();
    }
  };
  struct Joiner {
    std::vector<std::thread> t;
    ~Joiner() {
      for (auto &th : t)
        if (th.joinable()) th.join();
    }
  } pool;
  try {
    pool.t.reserve(n_threads - 1);
    for (size_t th = 1; th < n_threads; ++th) pool.t.emplace_back(worker, (int)th);
  } catch (...) {   // std::system_error (no more threads) or std::bad_alloc: go on with the helpers that exist
  }
  worker(0);
  for (auto &th : pool.t) th.join();
  pool.t.clear();
  if (first) std::rethrow_exception(first);
}

}  // namespace escoin
#endif
