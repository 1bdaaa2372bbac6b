// fa_lease.h -- the two host-side waiting primitives of a query call, free of HIP so that they can run under
// ThreadSanitizer on the CPU (scripts/host_sanitize.sh):
//   Lease        a call borrows one of the owner's workspaces for its duration and blocks while all are taken (queries are
//                re-entrant like the reference's, src/pyfastani/_fastani.pyx:1158-1161: several host threads, one mapper);
//   spin_for_seq the host side of the pass hand-over: polls a word that the pass's last kernel releases into pinned host
//                memory, for a bounded time.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>

namespace fa {

// Owner: `mtx`, `ws_free` (condition variable), `ws[NWS]` with a bool `in_use` each, `last_ws`, `static constexpr int NWS`.
// `prepare(W &)` runs outside the lock on the workspace just taken (the engine creates its stream there); if it throws the
// workspace is handed back before the exception leaves the constructor.
template <class Owner, class W>
struct Lease {
  Owner &m;
  W *w = nullptr;
  int index = -1;
  template <class Prepare>
  Lease(Owner &mm, Prepare &&prepare) : m(mm) {
    std::unique_lock<std::mutex> lock(m.mtx);
    for (;;) {
      for (int i = 0; i < Owner::NWS; i++) if (!m.ws[i].in_use) { index = i; break; }
      if (index >= 0) break;
      m.ws_free.wait(lock);
    }
    w = &m.ws[index];
    w->in_use = true;
    lock.unlock();
    try {
      prepare(*w);
    } catch (...) {
      { std::lock_guard<std::mutex> relock(m.mtx); w->in_use = false; }
      m.ws_free.notify_one();
      throw;
    }
  }
  Lease(const Lease &) = delete;
  Lease &operator=(const Lease &) = delete;
  ~Lease() {
    { std::lock_guard<std::mutex> lock(m.mtx); w->in_use = false; m.last_ws = index; }
    m.ws_free.notify_one();
  }
};

// true as soon as *word == seq (acquire), false after spin_us microseconds without it (the caller then sleeps on the stream)
inline bool spin_for_seq(const uint32_t *word, uint32_t seq, uint64_t spin_us) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t it = 0; spin_us; it++) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return true;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((it & 255) == 255 && (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us) break;
  }
  return __atomic_load_n(word, __ATOMIC_ACQUIRE) == seq;
}

}  // namespace fa
