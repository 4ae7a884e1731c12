// thread_place.h -- where a helper thread of the library starts to run.
// A new thread starts on its creator's core, and on the hosts measured (VM guests: this build's container and the GPU
// box) the scheduler takes its time to move it -- a team of fresh threads spread one thread per 4 ms tick, a parked team
// woken by one thread ran on that thread's core for its first second.  place_on_own_core(slot) moves the CALLING thread
// to the slot-th core (modulo) of the affinity mask it inherited and gives it the whole mask back at once: it stays
// where it was put because that is now the core it last ran on, and the scheduler is still free to move it.  Nothing
// stays bound (the reference binds its OpenMP threads for good, cpu_info.cpp:483-605).
#ifndef ESCOIN_THREAD_PLACE_H_
#define ESCOIN_THREAD_PLACE_H_

#include <pthread.h>
#include <sched.h>

namespace escoin {

// Cores the calling thread may run on (its affinity mask; at least 1): how many helpers are worth starting.
inline int allowed_cores() {
  cpu_set_t mask;
  CPU_ZERO(&mask);
  if (pthread_getaffinity_np(pthread_self(), sizeof(mask), &mask) != 0) return 1;
  const int n = CPU_COUNT(&mask);
  return n > 0 ? n : 1;
}

inline void place_on_own_core(int slot) {
  cpu_set_t inherited;
  CPU_ZERO(&inherited);
  if (pthread_getaffinity_np(pthread_self(), sizeof(inherited), &inherited) != 0) return;
  const int n = CPU_COUNT(&inherited);
  if (n < 2) return;
  int want = slot % n, cpu = -1;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &inherited) && want-- == 0) {
      cpu = c;
      break;
    }
  if (cpu < 0) return;
  cpu_set_t one;
  CPU_ZERO(&one);
  CPU_SET(cpu, &one);
  if (pthread_setaffinity_np(pthread_self(), sizeof(one), &one) != 0) return;
  sched_yield();   // (runs on `cpu` from here)
  pthread_setaffinity_np(pthread_self(), sizeof(inherited), &inherited);
}

}  // namespace escoin
#endif
