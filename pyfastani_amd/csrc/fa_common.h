// fa_common.h -- shared host-side plumbing of libfastani_hip: error type, HIP checks, device buffers.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "fa_error.h"

namespace fa {

inline void hip_check(hipError_t e, const char *what, const char *file, int line) {
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    int code = (e == hipErrorOutOfMemory) ? FA_ERR_NOMEM : FA_ERR_NO_DEVICE;
    (void)hipGetLastError();
    throw Error(code, buf);
  }
}
#define FA_HIP(x) ::fa::hip_check((x), #x, __FILE__, __LINE__)

// Device memory that is handed back is kept, not freed (round 5).  A mapper life cycle allocates and frees tens of GB in a few
// dozen blocks (a 4 x 10^8-record index: 33 GB; its build: 10 GB of temporaries), and on this runtime memory that comes back from
// hipFree is scrubbed before it is handed out again: the SECOND index build of a process took 0.90 s where the first took 0.07
// (FA_TRACE=1: alloc 0.3 -> 255 ms, and every kernel that ran next to the scrubbing 5-10 x slower; profiles/r05_trace_index.txt) --
// that, not the sort, was the 0.58 s `index_build_s` of the config-3 leg.  Blocks are kept per device in size classes (four per
// octave: at most a quarter of a block is slack) and reused by the next request of their class.  `release` synchronises the
// device before the block can be handed out again -- what hipFree did implicitly, and what callers of `ensure` / `release`
// relied on.  The pool holds at most FA_POOL_MAX_GB (default 96) and gives everything back when an allocation fails.
class DevPool {
 public:
  static DevPool &get() { static DevPool *p = new DevPool(); return *p; }
  static size_t class_of(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int lg = 63 - __builtin_clzll((unsigned long long)(bytes - 1));   // 2^lg < bytes <= 2^(lg+1)
    const size_t q = (size_t)1 << (lg > 2 ? lg - 2 : 0);                 // a quarter of the octave
    return (bytes + q - 1) / q * q;
  }
  // `dev` receives the device the block lives on (the calling thread's current one): the owner hands it back to `free`
  void *alloc(size_t bytes, size_t &got, int &dev) {
    got = class_of(std::max<size_t>(bytes, 1));
    dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = free_.find({dev, got});
      if (it != free_.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        held_ -= got;
        return p;
      }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, got);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); trim(); e = hipMalloc(&p, got); }
    hip_check(e, "hipMalloc", __FILE__, __LINE__);
    return p;
  }
  // `dev` = the device `alloc` reported.  A handle may be released from a thread that never entered the library (Python's garbage
  // collector on a thread whose current device is 0): the barrier must be on the OWNING device and the block filed under it, so
  // the thread is switched there for the duration and back afterwards.
  void free(void *p, size_t got, int dev) {
    if (!p) return;
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();                                     // (hipFree's implicit barrier: nothing in flight reads the block)
    bool kept = false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (max_held_ == SIZE_MAX) size_cap();
      if (held_ + got <= max_held_) { free_[{dev, got}].push_back(p); held_ += got; kept = true; }
    }
    if (!kept) (void)hipFree(p);
    if (cur != dev) (void)hipSetDevice(cur);
  }
  // everything the pool holds back to the runtime
  void trim() {
    std::map<std::pair<int, size_t>, std::vector<void *>> all;
    { std::lock_guard<std::mutex> lk(mu_); all.swap(free_); held_ = 0; }
    for (auto &kv : all) for (void *p : kv.second) (void)hipFree(p);
  }
  size_t held() { std::lock_guard<std::mutex> lk(mu_); return held_; }

 private:
  DevPool() {}
  // FA_POOL_MAX_GB if set; else 30 % of the device's memory, 96 GB at most (decided at the first release: a device is current
  // then).  Other allocators of the process (torch tensors, RCCL buffers) never make the pool trim, so it must not sit on most of
  // a device; `fa_device_trim` gives everything back on request.
  void size_cap() {
    const char *e = getenv("FA_POOL_MAX_GB");
    double gb = 96.0;
    if (e) gb = atof(e);
    else {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) gb = std::min(96.0, 0.30 * (double)total_b / (1024.0 * 1024.0 * 1024.0));
      else (void)hipGetLastError();
    }
    max_held_ = gb <= 0 ? 0 : (size_t)(gb * 1024.0 * 1024.0 * 1024.0);
  }
  std::mutex mu_;
  std::map<std::pair<int, size_t>, std::vector<void *>> free_;
  size_t held_ = 0, max_held_ = SIZE_MAX;                            // (SIZE_MAX: not sized yet)
};

// Owning device array with geometric growth; contents are preserved on growth only when asked.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  size_t block = 0;      // bytes of the pool block behind p
  int dev = 0;           // the device the block lives on
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), block(o.block), dev(o.dev) { o.p = nullptr; o.cap = 0; o.block = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; block = o.block; dev = o.dev; o.p = nullptr; o.cap = 0; o.block = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) DevPool::get().free(p, block, dev);
    p = nullptr; cap = 0; block = 0;
  }
  void ensure(size_t n, bool keep = false, hipStream_t stream = nullptr, size_t used = 0) {
    if (n <= cap) return;
    size_t ncap = keep ? std::max(n, cap + cap / 2) : n;
    size_t got = 0;
    int ndev = 0;
    T *np = (T *)DevPool::get().alloc(std::max<size_t>(ncap, 1) * sizeof(T), got, ndev);
    if (keep && p && used) {
      hipError_t e = hipMemcpyAsync(np, p, used * sizeof(T), hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) { DevPool::get().free(np, got, ndev); hip_check(e, "copy on growth", __FILE__, __LINE__); }
    }
    if (p) DevPool::get().free(p, block, dev);
    p = np; cap = got / sizeof(T); block = got; dev = ndev;        // (the whole block is usable: fewer regrowths)
  }
  void upload(const T *src, size_t n, hipStream_t stream) {
    ensure(n);
    if (n) FA_HIP(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, stream));
  }
  void upload(const std::vector<T> &v, hipStream_t stream) { upload(v.data(), v.size(), stream); }
  void download(T *dst, size_t n, hipStream_t stream) const {
    if (n) FA_HIP(hipMemcpyAsync(dst, p, n * sizeof(T), hipMemcpyDeviceToHost, stream));
  }
};

}  // namespace fa
