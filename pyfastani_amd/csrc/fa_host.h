// fa_host.h -- the host side of the sequence store: upper-casing, 2-bit packing (AVX2 + scalar exception path) on a
// persistent thread pool.  Plain C++ (no HIP): fa_sketch.hip.h includes it for the library, scripts/host_sanitize.sh builds
// it with AddressSanitizer / UBSan / ThreadSanitizer on the CPU.
// Replaces copy_upper / reverse_complement of the reference's _sequtils (src/pyfastani/_sequtils/sequtils.cpp:22-90) on
// the way into the store; layout of the store: fa_sketch.hip.h.
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "fa_error.h"

namespace fa {

// ----------------------------------------------------------------------------------------------------------
// host: upper-casing, complement, packing (replaces copy_upper / reverse_complement of _sequtils)
// ----------------------------------------------------------------------------------------------------------
inline uint8_t host_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

inline uint8_t host_read(const void *data, int width, int64_t i) {
  switch (width) {
    case 1: return ((const uint8_t *)data)[i];
    case 2: return (uint8_t)((const uint16_t *)data)[i];
    default: return (uint8_t)((const uint32_t *)data)[i];
  }
}

// ASCII -> 2-bit code (A/a=0 C/c=1 G/g=2 T/t=3), 4 = not a plain nucleotide (goes to the exception list)
static const uint8_t kCodeOf[256] = {
#define X4(v) v, v, v, v
#define X16(v) X4(v), X4(v), X4(v), X4(v)
    X16(4), X16(4), X16(4), X16(4),                               // 0x00-0x3f
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4,               // @ A B C D E F G H I J K L M N O
    4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,               // P Q R S T ...
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4,               // ` a b c d e f g ...
    4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,               // p q r s t ...
    X16(4), X16(4), X16(4), X16(4), X16(4), X16(4), X16(4), X16(4)  // 0x80-0xff
#undef X16
#undef X4
};

// host threads used for packing (FA_HOST_THREADS overrides; default = hardware concurrency, at most 128, divided by the ranks
// of this node when the process is one of several -- LOCAL_WORLD_SIZE, set by torch.distributed.run: eight ranks of a node
// with 256 hardware threads start 32 packer threads each, not 8 x 128)
inline int host_threads() {
  static const int v = [] {
    const char *e = getenv("FA_HOST_THREADS");
    int x = e ? atoi(e) : 0;
    if (x <= 0) {
      const char *lw = getenv("LOCAL_WORLD_SIZE");
      const unsigned ranks = (unsigned)std::max(1, lw ? atoi(lw) : 1);
      x = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency() / ranks), 128u);
    }
    return x;
  }();
  return v;
}

// Persistent host threads for the packer: a 5 Mb query is packed in ~0.1 ms when the work is cut finely, which a
// thread spawn per call (tens of microseconds each) would eat.  A call publishes a job and wakes only as many workers
// as it can use (waking all 127 for every contig of a draft assembly cost seconds); the calling thread works too, so a
// job completes even if no worker shows up, and callers on different threads simply publish their own jobs.  Workers
// are detached and live as long as the process.
class HostPool {
 public:
  static HostPool &get() { static HostPool *p = new HostPool(host_threads()); return *p; }
  // `want_helpers` > 0: every item is milliseconds of work (a FASTA file to read and pack) and the caller says how many workers
  // are worth waking
  void parallel_for(size_t total, const std::function<void(size_t)> &f, int want_helpers = 0) {
    // waking a thread costs about as much as a 64 Kbase chunk of packing: two chunks per helper at least
    // (measured on a 5 Mb query = 77 chunks: 16-32 helpers are fastest, 128 cost 15 % more); big jobs use every worker
    const size_t cap = total >= 4096 ? (size_t)nworkers_ : std::min<size_t>((size_t)nworkers_, 24);
    const int helpers = want_helpers > 0 ? (int)std::min<size_t>(std::min<size_t>((size_t)want_helpers, (size_t)nworkers_), total > 0 ? total - 1 : 0)
                                         : (int)std::min<size_t>(cap, total / 2);
    if (helpers == 0 || total < 3) { for (size_t i = 0; i < total; i++) f(i); return; }
    auto job = std::make_shared<Job>();
    job->fn = &f; job->total = total;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = job; epoch_++;
    }
    for (int i = 0; i < helpers; i++) cv_work_.notify_one();
    work(*job);
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return job->done.load() == total; });
    }
    // an item that threw (bad_alloc while collecting exceptions, say) was still counted, so every helper has let go of
    // the caller's function object by now; the first exception is re-raised on the calling thread
    if (job->failed.load()) std::rethrow_exception(job->error);
  }

 private:
  struct Job {
    const std::function<void(size_t)> *fn = nullptr;
    size_t total = 0;
    std::atomic<size_t> next{0}, done{0};
    std::atomic<bool> failed{false};
    std::exception_ptr error;         // first exception of any item (written once, under mu_)
  };
  explicit HostPool(int threads) : nworkers_(std::max(0, threads - 1)) {
    for (int id = 0; id < nworkers_; id++) std::thread([this] { worker(); }).detach();
  }
  // a worker that arrives late finds the items of its job taken and never touches the (by then dead) function object
  void work(Job &j) {
    for (size_t i; (i = j.next.fetch_add(1)) < j.total;) {
      try {
        (*j.fn)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!j.failed.exchange(true)) j.error = std::current_exception();
      }
      if (j.done.fetch_add(1) + 1 == j.total) { std::lock_guard<std::mutex> lk(mu_); cv_done_.notify_all(); }
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        j = job_;
      }
      if (j) work(*j);
    }
  }
  const int nworkers_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::shared_ptr<Job> job_;
  uint64_t epoch_ = 0;
};

// 32 plain nucleotides -> two packed words with AVX2 (an EPYC host packs ~10 GB/s per core this way, the table loop ~1):
// code = ((c >> 1) ^ (c >> 2)) & 3 maps A/a C/c G/g T/t to 0 1 2 3; a byte is a plain nucleotide iff looking the code up
// in "ACGT" gives the byte back (after clearing the case bit).  Returns false -- and writes nothing -- when the group
// holds anything else; the caller then takes the scalar path, which also records the exceptions.
__attribute__((target("avx2"))) inline bool pack32_avx2(const uint8_t *src, uint32_t *dst) {
  const __m256i v = _mm256_loadu_si256((const __m256i *)src);
  const __m256i three = _mm256_set1_epi8(3);
  const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i back = _mm256_shuffle_epi8(lut, code);
  const __m256i upper = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
  if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(upper, back)) != -1) return false;
  // codes (one per byte) -> 4 bits per pair -> 8 bits per quad, then the low byte of every dword is gathered
  const __m256i pairs = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));
  const __m256i sel = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i bytes = _mm256_shuffle_epi8(quads, sel);
  dst[0] = (uint32_t)_mm256_extract_epi32(bytes, 0);
  dst[1] = (uint32_t)_mm256_extract_epi32(bytes, 4);
  return true;
}
inline bool host_has_avx2() {
  static const bool v = __builtin_cpu_supports("avx2") && !getenv("FA_NO_AVX2");
  return v;
}

// std::vector whose resize(n) leaves new elements uninitialised: the packers write every word of what they append, and the
// zero fill of a 250 MB store by ONE thread (plus its page faults) took longer than reading and packing 200 genomes on all of
// them (scripts/ubench/ingest_host.cpp: 40 of 57 ms); the pages are now first touched by the pool's threads
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};

// Host image of a sequence store, appended to contig by contig and uploaded in one go.
struct HostStore {
  bool protein = false;
  std::vector<uint32_t, NoInitAlloc<uint32_t>> packed;   // nucleotide
  std::vector<uint8_t, NoInitAlloc<uint8_t>> bytes;      // protein
  std::vector<int64_t> exc_pos;
  std::vector<uint8_t> exc_val;
  std::vector<int64_t> seq_off;   // store offset of each sequence
  std::vector<int64_t> seq_len;
  int64_t total = 0;              // store length in bases (multiple of 64)

  void clear() {
    packed.clear(); bytes.clear(); exc_pos.clear(); exc_val.clear(); seq_off.clear(); seq_len.clear(); total = 0;
  }

  // appends one sequence; returns its index
  int64_t append(const void *data, int width, int64_t len) {
    const void *ptrs[1] = {data};
    int64_t lens[1] = {len};
    int64_t first = (int64_t)seq_off.size();
    append_many(ptrs, lens, 1, width);
    return first;
  }

  // bases (padded to 64 per sequence) a call of append_many / pack_many adds to the store
  static int64_t padded_bases(const int64_t *lens, int64_t n) {
    int64_t add = 0;
    for (int64_t q = 0; q < n; q++) add += (lens[q] + 63) / 64 * 64;
    return add;
  }

  // Appends n sequences, packing them with a pool of host threads (the packer is the host-side bottleneck of the
  // many-to-many workloads).  Work is cut into chunks of whole words; every chunk collects its own exceptions, which
  // are concatenated in store order afterwards.
  void append_many(const void *const *datas, const int64_t *lens, int64_t n, int width) {
    // one allocation for the whole call (a thousand genomes would otherwise regrow -- and copy -- the store many times;
    // geometric: an exact reserve on every call would copy the whole store once per added contig)
    const size_t add = (size_t)padded_bases(lens, n);
    auto grow = [](auto &v, size_t need) { if (need > v.capacity()) v.reserve(std::max(need, v.capacity() * 2)); };
    // all or nothing: a failure half-way (an allocation while collecting exceptions, say) leaves the store as it was
    const size_t o_packed = packed.size(), o_bytes = bytes.size(), o_seq = seq_off.size(), o_exc = exc_pos.size();
    const int64_t o_total = total;
    try {
      if (protein) {
        grow(bytes, o_bytes + add);
        bytes.resize(o_bytes + add);                                  // (uninitialised: pack_many writes every byte / word)
        pack_many(datas, lens, n, width, nullptr, bytes.data() + o_bytes);
      } else {
        grow(packed, o_packed + add / 16);
        packed.resize(o_packed + add / 16);
        pack_many(datas, lens, n, width, packed.data() + o_packed, nullptr);
      }
    } catch (...) {
      packed.resize(o_packed); bytes.resize(o_bytes); seq_off.resize(o_seq); seq_len.resize(o_seq);
      exc_pos.resize(o_exc); exc_val.resize(o_exc); total = o_total;
      throw;
    }
  }

  // The packer proper: n sequences into caller-provided memory (dst32: padded_bases / 16 words for nucleotides, dst8:
  // padded_bases bytes for protein), which may be pinned staging memory that is uploaded as it stands.  Every word or
  // byte of the destination is written (padding as zeros).  Sequence offsets, lengths and exceptions are appended to
  // this store; `total` advances.
  void pack_many(const void *const *datas, const int64_t *lens, int64_t n, int width, uint32_t *dst32, uint8_t *dst8) {
    struct Chunk { const void *data; int64_t src0, count, store_off; size_t word0; std::vector<int64_t> epos; std::vector<uint8_t> eval; };
    std::vector<Chunk> chunks;
    // bases per chunk (multiple of 64): fine enough to spread one 5 Mb genome over the helpers, coarse enough that waking
    // a helper (tens of microseconds) is worth it -- the AVX2 loop packs 64 Kbases in ~8 us, the table loop in ~60 us
    static const int64_t ch_env = [] { const char *e = getenv("FA_PACK_CHUNK"); long long x = e ? atoll(e) : 0; return (int64_t)(x > 0 ? (x + 63) / 64 * 64 : 0); }();
    const bool avx2 = host_has_avx2() && width == 1 && !protein;
    const int64_t CH = ch_env ? ch_env : (avx2 ? (1 << 18) : (1 << 16));
    seq_off.reserve(seq_off.size() + (size_t)n); seq_len.reserve(seq_len.size() + (size_t)n);
    size_t base = 0;                                                   // words / bytes written so far in the destination
    for (int64_t q = 0; q < n; q++) {
      const int64_t len = lens[q], off = total, padded = (len + 63) / 64 * 64;
      seq_off.push_back(off);
      seq_len.push_back(len);
      for (int64_t c0 = 0; c0 < len; c0 += CH) {
        Chunk c;
        c.data = datas[q]; c.src0 = c0; c.count = std::min(CH, len - c0); c.store_off = off + c0;
        c.word0 = base + (size_t)(protein ? c0 : c0 / 16);
        chunks.push_back(std::move(c));
      }
      // padding behind the last (possibly partial) word / byte of the sequence
      if (protein) { for (int64_t i = len; i < padded; i++) dst8[base + (size_t)i] = 0; }
      else { for (int64_t i = (len + 15) / 16; i < padded / 16; i++) dst32[base + (size_t)i] = 0u; }
      base += (size_t)(protein ? padded : padded / 16);
      total += padded;
    }
    auto work = [&](Chunk &c) {
      if (protein) {
        uint8_t *dst = dst8 + c.word0;
        for (int64_t i = 0; i < c.count; i++) dst[i] = host_upper(host_read(c.data, width, c.src0 + i));
        return;
      }
      uint32_t *dst = dst32 + c.word0;
      const uint8_t *src8 = width == 1 ? (const uint8_t *)c.data + c.src0 : nullptr;
      int64_t i = 0;
      auto slow16 = [&](int64_t at) {
        uint32_t wv = 0;
        for (int j = 0; j < 16; j++) {
          const uint8_t ch = src8[at + j];
          uint8_t code = kCodeOf[ch];
          if (code > 3) { c.epos.push_back(c.store_off + at + j); c.eval.push_back(host_upper(ch)); code = 0; }
          wv |= (uint32_t)code << (2 * j);
        }
        dst[at >> 4] = wv;
      };
      if (src8 && avx2) {
        for (; i + 32 <= c.count; i += 32)
          if (!pack32_avx2(src8 + i, dst + (i >> 4))) { slow16(i); slow16(i + 16); }
      }
      if (src8) {
        // 16 bases -> one word, branch-free; a group that holds anything but ACGT/acgt is redone by the general loop
        for (; i + 16 <= c.count; i += 16) {
          uint32_t wv = 0, bad = 0;
#pragma GCC unroll 16
          for (int j = 0; j < 16; j++) { const uint32_t code = kCodeOf[src8[i + j]]; bad |= code; wv |= (code & 3u) << (2 * j); }
          if (bad > 3u) slow16(i); else dst[i >> 4] = wv;
        }
      }
      for (; i < c.count; i += 16) {
        const int m = (int)std::min<int64_t>(16, c.count - i);
        uint32_t wv = 0;
        for (int j = 0; j < m; j++) {
          const uint8_t ch = src8 ? src8[i + j] : host_read(c.data, width, c.src0 + i + j);
          uint8_t code = kCodeOf[ch];
          if (code > 3) { c.epos.push_back(c.store_off + i + j); c.eval.push_back(host_upper(ch)); code = 0; }
          wv |= (uint32_t)code << (2 * j);
        }
        dst[i >> 4] = wv;
      }
    };
    HostPool::get().parallel_for(chunks.size(), [&](size_t i) { work(chunks[i]); });
    for (auto &c : chunks) {
      exc_pos.insert(exc_pos.end(), c.epos.begin(), c.epos.end());
      exc_val.insert(exc_val.end(), c.eval.begin(), c.eval.end());
    }
  }
};

}  // namespace fa
