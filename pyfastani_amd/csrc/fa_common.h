// fa_common.h -- shared host-side plumbing of libfastani_hip: error type, HIP checks, device buffers.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>

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

// Owning device array with geometric growth; contents are preserved on growth only when asked.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
  void ensure(size_t n, bool keep = false, hipStream_t stream = nullptr, size_t used = 0) {
    if (n <= cap) return;
    size_t ncap = keep ? std::max(n, cap + cap / 2) : n;
    T *np = nullptr;
    FA_HIP(hipMalloc((void **)&np, std::max<size_t>(ncap, 1) * sizeof(T)));
    if (keep && p && used) {
      FA_HIP(hipMemcpyAsync(np, p, used * sizeof(T), hipMemcpyDeviceToDevice, stream));
      FA_HIP(hipStreamSynchronize(stream));
    }
    if (p) (void)hipFree(p);
    p = np; cap = ncap;
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
