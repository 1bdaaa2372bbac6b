// fa_engine.hip -- host side of libfastani_hip.so: owns the HIP stream and all HBM allocations, drives the
// kernels of fa_sketch.hip.h / fa_map.hip.h, and exports the C ABI declared in include/fastani_hip.h.
// There is no CPU fallback anywhere in this file: without a HIP device every compute entry point fails.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <utility>
#include <vector>
#include <cstddef>
#include <deque>

#include "fa_common.h"
#include "fa_fasta.h"
#include "fa_lease.h"
#include "fa_map.hip.h"
#include "fa_sketch.hip.h"
#include "fa_sketch_fast.hip.h"
#include "fa_stats.h"

using namespace fa;

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

template <typename F>
static int guarded(F &&fn) {
  try {
    fn();
    return FA_OK;
  } catch (const Error &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc &) {
    g_last_error = "host allocation failed";
    return FA_ERR_NOMEM;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return FA_ERR_INTERNAL;
  }
}

// The device chosen with fa_set_device (-1: whatever the calling thread has current).  HIP's current device is a
// per-thread setting, and sketches / mappers are used from any host thread: each object remembers its device and every
// entry point binds it again.
static std::atomic<int> g_device{-1};
static void bind_device(int device) {
  if (device >= 0) FA_HIP(hipSetDevice(device));
}

static void require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    throw Error(FA_ERR_NO_DEVICE, "no HIP device available: libfastani_hip has no CPU fallback");
  }
}

static void validate_params(const fa_params &p) {
  FA_REQUIRE(p.kmer_size >= 1 && p.kmer_size <= 2048, FA_ERR_INVALID, "kmer_size must be in [1, 2048]");
  FA_REQUIRE(p.fragment_length >= 1, FA_ERR_INVALID, "fragment_length must be strictly positive");
  FA_REQUIRE(p.window_size >= 1, FA_ERR_INVALID, "window_size must be strictly positive");
  FA_REQUIRE(p.alphabet_size == 4 || p.alphabet_size == 20, FA_ERR_INVALID, "alphabet_size must be 4 or 20");
  size_t lds = sketch_lds_bytes(p.kmer_size, p.window_size);
  FA_REQUIRE(lds <= 160 * 1024, FA_ERR_UNSUPPORTED,
             "window_size/kmer_size too large for the LDS-staged sketch kernel (tile + 2w + k must fit 160 KiB)");
}

static int floor_log2(int v) {
  int l = 0;
  while ((2 << l) <= v) l++;
  return l;
}

// ------------------------------------------------------------------------------------------------------------
// K1 launcher shared by the reference and the query side
// ------------------------------------------------------------------------------------------------------------
struct SketchWork {
  DevBuf<Tile> tiles;
  DevBuf<uint32_t> stage_hash;
  DevBuf<int32_t> stage_wpos;
  DevBuf<int32_t> tile_count;   // ntiles + 1
  DevBuf<int32_t> tile_off;     // ntiles + 1
  DevBuf<unsigned char> cub_temp;
};

// FA_TRACE=1: wall-clock of the host-visible stages of sketching and index construction on stderr
struct StageTrace {
  bool on, live = false;
  const char *what;
  std::chrono::steady_clock::time_point t0, t;
  std::vector<std::pair<std::string, double>> acc;
  explicit StageTrace(const char *w) : on(getenv("FA_TRACE") != nullptr), what(w) {
    live = on && atoi(getenv("FA_TRACE")) >= 2;
    t0 = t = std::chrono::steady_clock::now();
  }
  void mark(const char *stage, hipStream_t st) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    auto now = std::chrono::steady_clock::now();
    double ms = std::chrono::duration<double, std::milli>(now - t).count();
    t = now;
    if (live) { fprintf(stderr, "[fa trace] %s: %s done after %.1f ms\n", what, stage, ms); fflush(stderr); }   // FA_TRACE=2: as it happens
    for (auto &a : acc) if (a.first == stage) { a.second += ms; return; }
    acc.emplace_back(stage, ms);
  }
  ~StageTrace() {
    if (!on) return;
    fprintf(stderr, "[fa trace] %s: %.1f ms total;", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (auto &a : acc) fprintf(stderr, " %s %.1f", a.first.c_str(), a.second);
    fprintf(stderr, "\n");
  }
};

// `clear` (a query pass): ranges zeroed by extra workgroups of the first launch, beside the hashing
static uint64_t env_u64(const char *name, uint64_t dflt);
static uint64_t exp_u64(const char *name, uint64_t dflt);     // the same in builds with -DFA_EXPERIMENTS, else `dflt`

// `fuse` (a query pass, F fragments): K1 and the per-fragment sketch in one launch (k_query_fused) where the pass qualifies
// -- returns true then, and the caller skips k_query_sketch
static bool launch_sketch_tiles(const fa_params &P, const StoreView &store, const Tile *d_tiles, int ntiles, uint32_t *stage_hash,
                                int32_t *stage_wpos, int32_t *tile_count, hipStream_t st, const ClearArgs *clear = nullptr,
                                const QuerySketchArgs *fuse = nullptr, int64_t F = 0) {
  if (ntiles <= 0) {
    if (clear && clear->count) hipLaunchKernelGGL(k_clear, dim3(256), dim3(256), 0, st, *clear);
    return false;
  }
  SketchArgs a;
  a.ntiles = ntiles;
  a.clear.count = 0; a.clear.stamp = nullptr;
  int extra = 0;
  if (clear && clear->count) {
    a.clear = *clear;
    uint64_t most = 0;
    for (int i = 0; i < clear->count; i++) most = std::max(most, clear->n16[i]);
    extra = (int)std::min<uint64_t>(512, std::max<uint64_t>(1, (most + SK_THREADS * 8 - 1) / (SK_THREADS * 8)));
  }
  a.tiles = d_tiles;
  a.packed = store.packed; a.bytes = store.bytes; a.exc_pos = store.exc_pos; a.exc_val = store.exc_val;
  a.stage_hash = stage_hash; a.stage_wpos = stage_wpos; a.tile_count = tile_count;
  a.k = P.kmer_size; a.w = P.window_size; a.levels = floor_log2(P.window_size);
  a.protein = P.alphabet_size != 4;
  a.npos_cap = TILE + 2 * P.window_size - 2;
  // FA_K1_GENERAL=1: every tile through the 64-bit form of the window minimum (tests compare the two)
  static const bool k1_general = getenv("FA_K1_GENERAL") && atoi(getenv("FA_K1_GENERAL")) != 0;
  a.fast = (!k1_general && P.window_size >= 3 && P.window_size <= 1000) ? 1 : 0;   // (three padded arrays in the LDS of two key arrays)
  size_t lds = sketch_lds_bytes(P.kmer_size, P.window_size);
  size_t image = lds - ((size_t)a.npos_cap * 16 + ((size_t)a.npos_cap / 64 + 1) * 8 + (TILE / 64) * 8 + (TILE / 64 + 1) * 4 + 16 + 4 * 256 * 8);
  a.code_words = (int32_t)(image / 4);
  static const size_t lds_pad = (size_t)exp_u64("FA_K1_LDS_PAD", 0);     // experiment: unused LDS, i.e. fewer workgroups per CU
  auto launch = [&](auto kernel) {
    if (lds + lds_pad > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + lds_pad)));
    hipLaunchKernelGGL(kernel, dim3(ntiles + extra), dim3(SK_THREADS), lds + lds_pad, st, a);
    extra = 0; a.clear.count = 0; a.clear.stamp = nullptr;          // (only the first launch zeroes)
  };
  // plain-ACGT tiles from the 2-bit image; protein tiles and tiles with other bytes through the byte image
  if (!a.protein && !k1_general && P.window_size >= SKF_MIN_W && P.window_size <= SKF_MAX_W) {
    // the hot form (fa_sketch_fast.hip.h): 13 KB of LDS, eight workgroups per CU; k = 14 / 16 / 21 hash from the premix tables
    const size_t flds = skf_layout(P.kmer_size, P.window_size).total;
    // The (k, w) cells of the default identity cut-off over the usual fragment lengths -- BASELINE config 5's grid: k in
    // {14, 16, 21} x fragment in {1000, 3000, 5000}, windows from recommendedWindowSize -- are built with BOTH parameters at
    // compile time (the window loop unrolled, 42 registers instead of 61 in k_sketch_fast); any other window takes the
    // run-time form of its k.
#define FA_KW_CELLS(X) X(14, 12) X(14, 37) X(14, 50) X(16, 13) X(16, 24) X(16, 40) X(21, 15) X(21, 25)
    // fused with the per-fragment sketch (k_query_fused, one of those cells): fragments whose records fit QF_CAP with room to
    // spare (a denser one voids the pass).  Tiles that touch a byte outside ACGT are sketched by k_sketch_tiles<0, true> into the
    // staging arrays FIRST (a launch whose other workgroups exit at once, and which carries the zeroing workgroups of the pass);
    // the fused kernel takes their records from there.  The run-time-w forms of the fused kernel do not fit the registers of seven waves per
    // SIMD -- a few words would go to scratch memory, which the runtime then keeps per stream for good -- and are not built.
    static const bool fuse_on = !(getenv("FA_QUERY_FUSED") && atoi(getenv("FA_QUERY_FUSED")) == 0);
    const bool may_fuse = fuse && fuse_on && (int64_t)5 * P.fragment_length / (P.window_size + 1) <= QF_CAP;
    auto launch_fast = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(ntiles + extra), dim3(SK_THREADS), flds + lds_pad, st, a);
      extra = 0; a.clear.count = 0; a.clear.stamp = nullptr;
    };
    bool served = false;
#define FA_TRY_CELL(K, W)                                                                                                        \
    if (!served && P.kmer_size == K && P.window_size == W) {                                                                      \
      served = true;                                                                                                              \
      if (may_fuse) {                                                                                                             \
        QuerySketchArgs qa = *fuse;                                                                                               \
        qa.exc_tiles = store.n_exc > 0 ? 1 : 0;                                                                                   \
        if (store.n_exc > 0) launch(k_sketch_tiles<0, true>);                                                                     \
        const size_t qlds = flds + (size_t)QF_CAP * 4;                                                                            \
        hipLaunchKernelGGL((k_query_fused<K, W>), dim3((unsigned)F + extra), dim3(SK_THREADS), qlds + lds_pad, st, a, qa, (int)F);  \
        FA_HIP(hipGetLastError());                                                                                                \
        return true;                                                                                                              \
      }                                                                                                                           \
      launch_fast(k_sketch_fast<K, W>);                                                                                           \
    }
    FA_KW_CELLS(FA_TRY_CELL)
#undef FA_TRY_CELL
#undef FA_KW_CELLS
    if (served) {}
    else if (P.kmer_size == 16) launch_fast(k_sketch_fast<16, 0>);
    else if (P.kmer_size == 14) launch_fast(k_sketch_fast<14, 0>);
    else if (P.kmer_size == 21) launch_fast(k_sketch_fast<21, 0>);
    else launch_fast(k_sketch_fast<0, 0>);
  } else if (!a.protein) { if (P.kmer_size == 16) launch(k_sketch_tiles<16, false>); else launch(k_sketch_tiles<0, false>); }
  if (a.protein || store.n_exc > 0) launch(k_sketch_tiles<0, true>);
  FA_HIP(hipGetLastError());
  return false;
}

// rocPRIM directly (device-wide scan / radix sort / run-length encode; sizes are size_t)
static void exclusive_sum_i32(DevBuf<unsigned char> &temp, const int32_t *in, int32_t *out, size_t n, hipStream_t st) {
  size_t bytes = 0;
  FA_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), st));
  temp.ensure(bytes + 16);
  FA_HIP(rocprim::exclusive_scan(temp.p, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), st));
}

// ------------------------------------------------------------------------------------------------------------
// fa_sketch: reference genomes being collected (skch::Sketch + pyfastani's counters)
// ------------------------------------------------------------------------------------------------------------
struct fa_sketch {
  fa_params P;
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t up_stream = nullptr;        // uploads of a flush (the next chunk's sequence while this one is hashed)
  HostStore pending;                      // packed contigs not yet sketched
  std::vector<int32_t> pending_contig;    // contig id of each pending sequence
  int64_t counter = 0;                    // contigs seen (Sketch._counter)
  uint64_t cur_total = 0;
  std::vector<uint64_t> lengths;          // per genome, rounded to whole fragments
  std::vector<int32_t> seqs_by_file;      // sequencesByFileInfo
  DevBuf<uint32_t> rec_hash;
  DevBuf<int32_t> rec_seq, rec_wpos;
  int64_t nrec = 0;
  SketchWork work;
  std::mutex mtx;

  void reset_data() {
    pending.clear(); pending.protein = P.alphabet_size != 4;
    pending_contig.clear();
    counter = 0; cur_total = 0; lengths.clear(); seqs_by_file.clear(); nrec = 0;
  }

  // sketch every pending contig on the device and append the records
  void flush() {
    if (pending.seq_off.empty()) return;
    require_device();
    bind_device(device);
    if (!stream) FA_HIP(hipStreamCreate(&stream));
    StageTrace tr("sketch flush");
    // the packed sequence goes up chunk by chunk on a stream of its own: the copy of chunk c + 1 (a staged copy from pageable
    // memory: the host thread is inside it) runs while the device hashes chunk c -- 25-50 ms of a thousand genomes' 90-110
    if (!up_stream) FA_HIP(hipStreamCreate(&up_stream));
    DevStore store;
    store.begin(pending, stream);
    const int64_t nseq_all = (int64_t)pending.seq_off.size();
    // staging is 8 B per k-mer position; a chunk takes an eighth of the free HBM at most, between 96 M positions (768 MiB) and
    // 512 M (4 GiB: a thousand genomes are ten chunks, not fifty-two with two synchronisations each -- and not two chunks whose
    // 34 GB of staging a fresh process has to map first)
    size_t hbm_free = 0, hbm_total = 0;
    if (hipMemGetInfo(&hbm_free, &hbm_total) != hipSuccess) { (void)hipGetLastError(); hbm_free = 0; }
    const int64_t chunk_positions = std::max<int64_t>(96LL << 20, std::min<int64_t>((int64_t)(hbm_free / 8 / 8), 512LL << 20));
    // the records are appended chunk by chunk: reserve for all of them once (2 / (w + 1) of the positions are minimizers, a
    // quarter of headroom; `ensure(keep)` below still grows the arrays if a sequence is denser) instead of regrowing -- and
    // copying -- them a dozen times
    {
      int64_t positions_all = 0;
      for (int64_t q = 0; q < nseq_all; q++) positions_all += pending.seq_len[q];
      const double per_pos = P.alphabet_size == 4 ? 2.0 / (P.window_size + 1) : 1.0;
      const size_t expect = (size_t)nrec + (size_t)((double)positions_all * std::min(1.0, per_pos * 1.25)) + 1024;
      rec_hash.ensure(expect, true, stream, (size_t)nrec);
      rec_seq.ensure(expect, true, stream, (size_t)nrec);
      rec_wpos.ensure(expect, true, stream, (size_t)nrec);
    }
    // chunk boundaries first (the copy of the next chunk is issued while this one is hashed)
    std::vector<int64_t> cuts{0};
    for (int64_t q = 0, positions = 0; q < nseq_all; q++) {
      if (q > cuts.back() && positions + pending.seq_len[q] > chunk_positions) { cuts.push_back(q); positions = 0; }
      positions += pending.seq_len[q];
    }
    cuts.push_back(nseq_all);
    auto base_of = [&](int64_t q) { return q < nseq_all ? pending.seq_off[(size_t)q] : pending.total; };
    std::vector<hipEvent_t> sent(cuts.size() - 1, nullptr);
    struct EventsGuard { std::vector<hipEvent_t> &v; ~EventsGuard() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } events_guard{sent};
    auto send = [&](size_t c) {
      store.upload_bases(pending, base_of(cuts[c]), base_of(cuts[c + 1]), up_stream);
      FA_HIP(hipEventCreateWithFlags(&sent[c], hipEventDisableTiming));
      FA_HIP(hipEventRecord(sent[c], up_stream));
    };
    FA_HIP(hipStreamSynchronize(stream));           // (the buffers of `begin` exist before the other stream writes them)
    send(0);
    tr.mark("upload", up_stream);
    DevBuf<int32_t> d_seq_tile_lo, d_drop, d_drop_off, d_seq_ids;
    for (size_t chunk = 0; chunk + 1 < cuts.size(); chunk++) {
      const int64_t s0 = cuts[chunk], s1 = cuts[chunk + 1];
      std::vector<Tile> tiles;
      std::vector<int32_t> seq_tile_lo, seq_ids;
      // (tiles whose positions + halo are whole hashing trips: k1_tile_len)
      const int tile_len = k1_tile_len(P.window_size);
      // (five million tiles for a thousand genomes: counted first, then filled by the host pool -- 91 ms of one thread before)
      int64_t tile_total = 0;
      for (int64_t q = s0; q < s1; q++) {
        seq_tile_lo.push_back((int32_t)tile_total);
        tile_total += tile_count(pending.seq_len[q], P.kmer_size, tile_len);
        seq_ids.push_back(pending_contig[q]);
      }
      seq_tile_lo.push_back((int32_t)tile_total);
      FA_REQUIRE(tile_total < (1LL << 31) - 1, FA_ERR_UNSUPPORTED, "too many tiles in one sketch chunk");
      tiles.resize((size_t)tile_total);
      {
        const int64_t nq = s1 - s0, per = std::max<int64_t>(1, nq / 512);       // (a few hundred tasks at most)
        HostPool::get().parallel_for((size_t)((nq + per - 1) / per), [&](size_t t) {
          for (int64_t q = s0 + (int64_t)t * per; q < std::min(s1, s0 + ((int64_t)t + 1) * per); q++)
            make_tiles_at(tiles.data() + seq_tile_lo[(size_t)(q - s0)], pending, pending.seq_off[q], pending.seq_len[q], (int)(q - s0), P.kmer_size, P.window_size, tile_len);
        });
      }
      const int nseq = (int)(s1 - s0), ntiles = (int)tiles.size();
      tr.mark("host_tiles", stream);
      if (ntiles > 0) {
        work.tiles.upload(tiles, stream);
        work.stage_hash.ensure((size_t)ntiles * TILE);
        work.stage_wpos.ensure((size_t)ntiles * TILE);
        work.tile_count.ensure(ntiles + 1);
        work.tile_off.ensure(ntiles + 1);
        FA_HIP(hipMemsetAsync(work.tile_count.p + ntiles, 0, sizeof(int32_t), stream));
        tr.mark("alloc_tiles", stream);
        FA_HIP(hipStreamWaitEvent(stream, sent[chunk], 0));
        launch_sketch_tiles(P, store.view(), work.tiles.p, ntiles, work.stage_hash.p, work.stage_wpos.p, work.tile_count.p, stream);
        if (chunk + 2 < cuts.size()) send(chunk + 1);          // (behind the launch: the host copies while the device hashes)
        tr.mark("k_sketch", stream);
        d_seq_tile_lo.upload(seq_tile_lo, stream);
        d_seq_ids.upload(seq_ids, stream);
        d_drop.ensure(nseq + 1);
        d_drop_off.ensure(nseq + 1);
        FA_HIP(hipMemsetAsync(d_drop.p + nseq, 0, sizeof(int32_t), stream));
        hipLaunchKernelGGL(k_suppress_runs, dim3(ceil_div(nseq, 128)), dim3(128), 0, stream, d_seq_tile_lo.p, nseq,
                           work.tile_count.p, work.stage_hash.p, work.stage_wpos.p, d_drop.p);
        exclusive_sum_i32(work.cub_temp, work.tile_count.p, work.tile_off.p, ntiles + 1, stream);
        exclusive_sum_i32(work.cub_temp, d_drop.p, d_drop_off.p, nseq + 1, stream);
        int32_t total = 0, dropped = 0;
        FA_HIP(hipMemcpyAsync(&total, work.tile_off.p + ntiles, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipMemcpyAsync(&dropped, d_drop_off.p + nseq, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        const int64_t nout = (int64_t)total - dropped;
        tr.mark("scan", stream);
        rec_hash.ensure((size_t)(nrec + nout), true, stream, (size_t)nrec);
        rec_seq.ensure((size_t)(nrec + nout), true, stream, (size_t)nrec);
        rec_wpos.ensure((size_t)(nrec + nout), true, stream, (size_t)nrec);
        tr.mark("grow", stream);
        hipLaunchKernelGGL(k_compact_records, dim3(ntiles), dim3(256), 0, stream, work.tiles.p, work.tile_count.p,
                           work.tile_off.p, d_seq_tile_lo.p, d_drop.p, d_drop_off.p, work.stage_hash.p, work.stage_wpos.p,
                           d_seq_ids.p, nrec, rec_hash.p, rec_seq.p, rec_wpos.p);
        FA_HIP(hipGetLastError());
        FA_HIP(hipStreamSynchronize(stream));
        tr.mark("compact", stream);
        nrec += nout;
      } else if (chunk + 2 < cuts.size()) send(chunk + 1);
    }
    FA_HIP(hipStreamSynchronize(up_stream));
    pending.clear();
    pending.protein = P.alphabet_size != 4;
    pending_contig.clear();
  }
};

// ------------------------------------------------------------------------------------------------------------
// fa_genomes: packed query genomes resident in HBM, cut into fragments and tiles
// ------------------------------------------------------------------------------------------------------------
// Pinned host staging memory (grows, never shrinks; one per workspace / upload)
struct PinnedBuf {
  unsigned char *p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t ncap = std::max(n, cap * 2);
    FA_HIP(hipHostMalloc((void **)&p, ncap, hipHostMallocDefault));
    cap = ncap;
  }
};

// Everything the kernels read of a batch sits in ONE device allocation, filled by ONE host-to-device copy from a staging
// image of the same layout: [packed 2-bit words | residue bytes | tiles | frag_tile_lo | frag_query | frag_qseq |
// total_frag], every part 16-byte aligned.  Only the exception lists (rare: N runs, IUPAC codes) are separate.
struct fa_genomes {
  fa_params P;
  DevBuf<unsigned char> blob;
  DevBuf<int64_t> exc_pos;
  DevBuf<uint8_t> exc_val;
  StoreView store;                          // pointers into blob / the exception buffers
  const Tile *tiles = nullptr;
  const int32_t *d_frag_tile_lo = nullptr, *d_frag_query = nullptr, *d_frag_qseq = nullptr, *d_total_frag = nullptr;
  int32_t n_genomes = 0;
  std::vector<int64_t> genome_frag_lo;      // [n_genomes + 1] fragment range of each genome
  std::vector<int32_t> frag_tile_lo;        // [F + 1]
  std::vector<int64_t> contig_frag_lo;      // first fragment of every contig that holds fragments (ascending)
  std::vector<uint64_t> total_fragments, total_length;
  std::vector<int32_t> n_short;
  int64_t F = 0, ntiles = 0;
  uint64_t total_bases = 0;                 // bases inside fragments
  std::vector<unsigned char, NoInitAlloc<unsigned char>> host_image;   // staging image when no pinned buffer is supplied (every byte of it is written: no zero fill)
  PinnedBuf pin_image;                      // staging image of a batch that is refilled (fa_genomes_reload_fasta)
  hipStream_t up_stream = nullptr;          // FASTA uploads run on the batch's own stream
  ~fa_genomes() { if (up_stream) (void)hipStreamDestroy(up_stream); }
  fa_genomes() = default;
  uint64_t serial = 0;                      // a new number for every upload (caches keyed on a batch compare this, not its address)
};
static std::atomic<uint64_t> g_batch_serial{0};

// Every small counter / statistic of a pass in ONE device block, mirrored into pinned host memory by one copy.
static uint64_t env_u64(const char *name, uint64_t dflt) {
  const char *e = getenv(name);
  long long x = e ? atoll(e) : 0;
  return x > 0 ? (uint64_t)x : dflt;
}

struct PassStatus {
  int32_t stats[4];                 // [0] largest query sketch
  int32_t total_rows, pad0[3];
  uint64_t totals[4];               // seeds, largest fragment, scratch words, reference records in L2 ranges
  uint32_t counters[8];             // [0] fragments k_l1 merged instead of block-sorting, [1] fragments on the HBM road / cut by k_l1_big; [2] loci overflow, wide-state loci, finished row workgroups, k_l1 roads (2: FA_L1_STATS samples)
  unsigned long long pinfo[4];      // slide events reserved (fused L2 form), speculation flags
  unsigned long long ev_region[EV_REGIONS], rec_region[EV_REGIONS];   // k_l2_events: events reserved / records read per arena region
  uint32_t loci_region[LOCI_REGIONS];                                  // k_l1: loci reserved per region (LociRegions)
  unsigned long long dbg[16];       // FA_L1_STATS=1: shader-clock ticks of k_l1's phases, summed over the sampled workgroups (thread 0's view); [8..10] why block sorts gave up
  unsigned long long stamp[6];      // stage_stamp: pass start, lookup, L2, CGI, end (100 MHz ticks); not cleared with the rest
  uint32_t seq, pad1;               // host copy only: number of the pass whose status this is (k_publish_status)
};

static_assert(offsetof(PassStatus, stamp) % 16 == 0, "k_clear zeroes whole 16-byte words: the cleared prefix of the status block must end on one");

// The hand-over of a pass (publish_pass, fa_map.hip.h): copies the status block -- and, for a one-query call whose rows go to the host, the rows --
// into pinned host memory and then releases the pass number.  The host polls that word instead of waiting for a
// device-to-host copy and a stream synchronisation, which together return tens of microseconds after the GPU is done
// (more on a slow host: the step time of the bench varied by 0.09 ms between boxes on that account).
static PublishArgs publish_args(PassStatus *dev, PassStatus *host_mapped, uint32_t seq, const fa_cgi_row *rows_dev, fa_cgi_row *rows_host, int64_t cap) {
  PublishArgs p;
  p.status_dev = (uint32_t *)dev; p.status_host = (uint32_t *)host_mapped;
  p.words = (int32_t)(offsetof(PassStatus, seq) / 4); p.seq_word = p.words; p.seq = seq;
  p.stamp = dev->stamp; p.total_rows = &dev->total_rows;
  p.rows_dev = rows_dev; p.rows_host = rows_host; p.cap = cap;
  return p;
}
// regions of the event arena used by a part of F fragments (L2Args::n_regions): a power of two, one per sixteen fragments
static uint32_t ev_regions_for(int64_t F) {
  uint32_t n = 1;
  while (n < (uint32_t)EV_REGIONS && (int64_t)n * 16 <= F) n <<= 1;
  return n;
}
// host side of k_publish_status: polls for FA_SPIN_US microseconds (default 20 000), then sleeps on the stream
static void wait_published(const PassStatus *h, uint32_t seq, hipStream_t st) {
  static const uint64_t spin_us = env_u64("FA_SPIN_US", 20000);
  if (spin_for_seq(&h->seq, seq, spin_us)) return;
  FA_HIP(hipStreamSynchronize(st));
  FA_REQUIRE(__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == seq, FA_ERR_INTERNAL, "the status of the pass was not published");
}

// Everything one query call owns: its stream, every intermediate of the pipeline, its status block and timing events.
struct Workspace {
  bool in_use = false;
  hipStream_t stream = nullptr;
  SketchWork sk;
  DevBuf<uint32_t> q_hash, q_off, q_cnt, n_seeds, ovf_off, ovf_buf;
  DevBuf<PassStatus> status;
  PassStatus *h_status = nullptr;     // pinned, written by k_publish_status
  uint32_t seq = 0;                   // passes published on this workspace
  DevBuf<int32_t> q_size, l_frag, l_seq, l_start, l_end, l_group, l_shared, l_pos, row_count, row_flag, row_off;
  DevBuf<int32_t> l_beg, l_end0, l_last, l_ndrop, l_rfirst, l_rlast, l_rpart;
  DevBuf<uint32_t> l_nev, l_ioff, f_loci_lo, f_loci_n;
  DevBuf<unsigned char> items;
  DevBuf<uint8_t> l_redo, big_state;
  DevBuf<uint32_t> scan_hist, scan_order;   // k_l2_scan's loci by stream length: class counts + cursors, the order
  // workgroup order of k_l2_events for passes of several genomes (L2Args::frag_order), cached while the same part repeats
  DevBuf<int32_t> frag_order;
  PinnedBuf pin_order;
  uint64_t order_batch = 0;            // fa_genomes::serial of the batch the cached order was built for
  int64_t order_f0 = -1, order_f1 = -1;
  uint32_t order_len = 0;
  DevBuf<unsigned long long> group_best, bins;
  DevBuf<float> row_ident;
  DevBuf<fa_cgi_row> rows_dev;
  // LUT pointers captured for the call (the mapper may publish larger tables while this call is in flight)
  const int32_t *lut_min_hits = nullptr, *lut_pass = nullptr;
  const float *lut_ident = nullptr;
  // last-pass bookkeeping for the debug getters
  int64_t last_F = 0, last_f0 = 0;
  uint32_t last_loci = 0;             // loci of the last accepted part (all regions)
  uint32_t loci_n = 1, loci_shift = 0;                    // regions of the locus numbering of the part in flight / last accepted
  bool l1_pf = false, l1_small_class = false;             // the part in flight ran k_l1 with the pre-filter / with the 256-thread class next to others
  uint32_t last_region_count[LOCI_REGIONS] = {0};         // live loci per region of the last accepted part
  uint64_t last_items = 0;
  const fa_genomes *last_genomes = nullptr;
  float last_ms[24] = {0};
  hipEvent_t ev[6] = {nullptr};
  // the parts of a pass are pipelined over this workspace and two more *lanes* (run_query_pass): sub-workspaces with their
  // own stream and per-part buffers; `serial` tells the debug getters which lanes took part in the last call
  std::unique_ptr<Workspace> sub[2];
  hipEvent_t ev_bins = nullptr;
  uint64_t serial = 0;
  int64_t pass_f0 = 0, pass_F = 0;    // first fragment and number of fragments of the last pass
  // the one-query-at-a-time call (fa_mapper_query) recycles its batch object -- no device allocation per call -- and
  // builds the upload image in pinned memory; its rows come back through a pinned block too
  std::unique_ptr<fa_genomes> query_batch;
  PinnedBuf pin_image, pin_rows;
  ~Workspace() {
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_bins) (void)hipEventDestroy(ev_bins);
    if (stream) (void)hipStreamDestroy(stream);
    if (h_status) (void)hipHostFree(h_status);
  }
};

// ------------------------------------------------------------------------------------------------------------
// fa_mapper: the indexed reference + the workspace of the query pipeline
// ------------------------------------------------------------------------------------------------------------
struct fa_mapper {
  fa_params P;
  int device = -1;
  hipStream_t stream = nullptr;
  std::mutex mtx;
  // reference records and index (see fa_map.hip.h for the layout)
  DevBuf<uint32_t> rec_hash, uniq_hash, uniq_off, pos_ridx;
  DevBuf<uint4> table;
  DevBuf<int32_t> rec_seq, rec_wpos, rec_prev, rec_fwd, rec_bwd, contig_rec, contig_genome, contig_bin, genome_bin;
  DevBuf<uint8_t> rec_flags;
  DevBuf<uint2> rec_hg;           // (hash, packed window geometry + flags) for k_l2_events (empty when cmw >= 8191)
  DevBuf<uint16_t> rec_prev16;
  DevBuf<uint32_t> rec_gpos, wrap_rec;   // padded global coordinate of every record (low word) + its 2^32 boundaries, for k_l1
  int32_t n_wraps = 0, gpos_bits = 32;
#ifdef FA_EXPERIMENTS
  DevBuf<uint32_t> ev_bits;       // merged admit / drop order of the slide (k_event_bits), 2 bits per record
  DevBuf<uint2> rec_hf;           // hash + flags + distance to the previous record of the hash (k_pack_hf), for k_l2_fused
#endif
  bool packed_geo = false;
  int64_t N = 0, U = 0;
  int32_t C = 0, G = 0, table_bits = 4, freq_threshold = INT_MAX, total_bins = 0;
  std::vector<uint64_t> lengths;
  std::vector<int32_t> seqs_by_file;
  int32_t cmw = 0, qcap = 1;
  // LUTs
  StatTables stats;
  DevBuf<int32_t> d_min_hits, d_pass;
  DevBuf<float> d_ident;
  // data-dependent sizes speculated from earlier passes (see run_query_pass); shared by all workspaces, guarded by mtx
  struct Spec {
    bool init = false;
    int smax = 0;
    uint32_t seed_slots = 0;
    uint64_t scratch_words = 0, items_cap = 0;
    int64_t l_cap = 0;
    int64_t part_frags = 0;   // fragments per part of a pass (shrinks when a part overflows the 32-bit workspace)
    bool redo = false;        // launch the wide-state scan as well (set once a locus overflowed the one-byte state)
    // k_query_fused met a fragment with more records than its LDS holds: the void range runs again through the two kernels
    // (`fuse_off`, a property of that attempt only).  The mapper keeps a back-off, not a verdict: the first overflow costs
    // nothing afterwards, consecutive ones skip 1, 3, 7 ... 63 passes before the fused form is tried again, and a fused pass
    // that is accepted clears the record -- one dense fragment no longer decides every later query of the mapper.
    bool fuse_off = false;
    int fuse_skip = 0, fuse_penalty = 0;
    int smax_misses = 0;      // times the largest sketch outgrew the bound (the first growth is tight, later ones are not)
    // share of the fragments of the last accepted part in the two lower size classes of k_l1 (-1: not seen yet)
    float l1_small_share = -1.0f, l1_mid_share = -1.0f, l1_tiny_share = -1.0f;   // (tiny: up to half the small class's bound)
    bool l1_prefilter = false;  // an accepted part saw fragments fall off k_l1's block sort: later passes drop dead hits before the sort (sticky)
    int64_t l2_loci_last = 0;   // loci of the last accepted part: k_l2_scan sorts its loci by stream length when there are waves to balance
    bool l1_no_small = false;   // ... and they still did with the pre-filter on and the 256-thread class in use: its table is too small for this index (sticky)
  } spec;
  // Queries are re-entrant (_fastani.pyx:1158-1161): every call takes one of NWS workspaces -- its own stream and every
  // intermediate of the pipeline -- so calls from different host threads overlap on the device (their phases interleave,
  // which is worth ~27 % of throughput over strictly serial calls).  `mtx` guards the pool, the speculation record and
  // the LUTs; the index itself is read-only.
  static constexpr int NWS = 4;
  Workspace ws[NWS];
  std::condition_variable ws_free;
  int last_ws = 0;                    // workspace of the most recent call (stage getters, timings)
  bool stage_events = false;          // fa_mapper_set_stage_events
  std::vector<DevBuf<int32_t>> retired_i32;   // LUT generations still referenced by calls in flight
  std::vector<DevBuf<float>> retired_f32;

  IndexView view() const {
    IndexView v;
    v.rec_hash = rec_hash.p; v.rec_seq = rec_seq.p; v.rec_wpos = rec_wpos.p; v.rec_prev = rec_prev.p; v.rec_fwd = rec_fwd.p; v.rec_bwd = rec_bwd.p; v.rec_flags = rec_flags.p;
    v.rec_hg = packed_geo ? rec_hg.p : nullptr; v.rec_prev16 = packed_geo ? rec_prev16.p : nullptr; v.rec_gpos = rec_gpos.p; v.wrap_rec = wrap_rec.p; v.n_wraps = n_wraps; v.gpos_bits = gpos_bits;
#ifdef FA_EXPERIMENTS
    v.ev_bits = ev_bits.p; v.rec_hf = rec_hf.p;
#endif
    v.uniq_hash = uniq_hash.p; v.uniq_off = uniq_off.p; v.pos_ridx = pos_ridx.p; v.table = table.p;
    v.contig_rec = contig_rec.p; v.contig_genome = contig_genome.p; v.contig_bin = contig_bin.p; v.genome_bin = genome_bin.p;
    v.N = N; v.U = U; v.C = C; v.G = G; v.table_bits = table_bits; v.freq_threshold = freq_threshold; v.total_bins = total_bins;
    return v;
  }
};

__global__ void k_contig_bins(const int32_t *contig_rec, const int32_t *rec_wpos, int C, int bin_len, int32_t *nbins) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  int n = 0;
  if (c < C) {
    int lo = contig_rec[c], hi = contig_rec[c + 1];
    if (hi > lo) n = rec_wpos[hi - 1] / bin_len + 1;
  }
  nbins[c] = n;
}

// Knobs that only ever served A/B measurements (LDS padding, forced workgroup shapes, the fused L2 kernel) exist in builds
// with -DFA_EXPERIMENTS only (scripts/experiments/README.md); the product library ignores them.
#ifdef FA_EXPERIMENTS
static uint64_t exp_u64(const char *name, uint64_t dflt) { return env_u64(name, dflt); }
// FA_L2_FUSED=1: build the extra index arrays of the fused L2 kernel and use it (see run_query_pass)
static bool fused_l2_enabled() {
  static const bool on = getenv("FA_L2_FUSED") && atoi(getenv("FA_L2_FUSED")) != 0;
  return on;
}
#else
static uint64_t exp_u64(const char *, uint64_t dflt) { return dflt; }
#endif

// Sketch_t::index() + computeFreqHist() on the device
static void build_index(fa_mapper &m) {
  hipStream_t st = m.stream;
  const int64_t N = m.N;
  // record numbers are 32-bit throughout the mapping kernels (a seed hit IS a record number): 2^31 minimizers, about 5 000
  // genomes of 5 Mb, per index -- beyond that shard the references (sharding.build_ref_sharded_mapper)
  // (k_window_links adds up to a fragment's windows + 2 to a record number in 32-bit arithmetic: that much headroom below 2^31)
  FA_REQUIRE(N < (1LL << 31) - 1 - (int64_t)std::max(0, m.P.fragment_length) - 64, FA_ERR_UNSUPPORTED, "more than 2^31 minimizers in one index (shard the references)");
  m.C = m.seqs_by_file.empty() ? 0 : m.seqs_by_file.back();
  m.G = (int32_t)m.seqs_by_file.size();
  // minimizer windows per fragment.  When it is <= 0 no fragment holds a window, query sketches are empty and L2 never
  // runs (the degenerate (k=21, fragment 1000) cell); the window links are then built for 1 and never read
  m.cmw = std::max(1, m.P.fragment_length - (m.P.window_size - 1) - (m.P.kmer_size - 1));
  m.qcap = std::max(1, m.P.fragment_length - m.P.kmer_size + 1 - (m.P.window_size - 1));
  const int bin_len = m.P.fragment_length - 20;
  DevBuf<unsigned char> temp;
  StageTrace tr("build_index");
  // contig tables
  std::vector<int32_t> cg((size_t)m.C + 1, 0);
  {
    int g = 0;
    for (int c = 0; c < m.C; c++) {
      while (g < m.G && m.seqs_by_file[g] <= c) g++;
      cg[c] = g;
    }
  }
  m.contig_genome.upload(cg, st);
  m.contig_rec.ensure((size_t)m.C + 2);
  hipLaunchKernelGGL(k_contig_ranges, dim3(ceil_div(m.C + 1, 256)), dim3(256), 0, st, m.rec_seq.p, N, m.C, m.contig_rec.p);
  DevBuf<int32_t> nbins;
  nbins.ensure((size_t)m.C + 2);
  m.contig_bin.ensure((size_t)m.C + 2);
  hipLaunchKernelGGL(k_contig_bins, dim3(ceil_div(m.C + 1, 256)), dim3(256), 0, st, m.contig_rec.p, m.rec_wpos.p, m.C,
                     bin_len > 0 ? bin_len : 1, nbins.p);
  exclusive_sum_i32(temp, nbins.p, m.contig_bin.p, m.C + 1, st);
  std::vector<int32_t> cb((size_t)m.C + 1);
  m.contig_bin.download(cb.data(), (size_t)m.C + 1, st);
  FA_HIP(hipStreamSynchronize(st));
  m.total_bins = cb[m.C];
  std::vector<int32_t> gb((size_t)m.G + 1);
  for (int g = 0; g <= m.G; g++) {
    int c = g == 0 ? 0 : m.seqs_by_file[g - 1];
    gb[g] = cb[std::min(c, m.C)];
  }
  m.genome_bin.upload(gb, st);
  tr.mark("contigs", st);

  // hash-grouped order (stable radix sort keeps record order inside a hash group)
  m.pos_ridx.ensure((size_t)N + 4);
  m.rec_prev.ensure((size_t)N + 4);
  m.rec_fwd.ensure((size_t)N + 4);
  m.rec_bwd.ensure((size_t)N + 4);
  m.rec_flags.ensure(((size_t)N + 7) / 4 * 4);
  FA_HIP(hipMemsetAsync(m.rec_flags.p, 0, ((size_t)N + 7) / 4 * 4, st));
  m.U = 0;
  m.freq_threshold = INT_MAX;
  if (N > 0) {
    DevBuf<uint32_t> iota, sorted_hash, counts, counts_sorted, blk_lo;
    DevBuf<int32_t> num_runs;
    iota.ensure((size_t)N); sorted_hash.ensure((size_t)N); counts.ensure((size_t)N + 1); num_runs.ensure(1);
    tr.mark("alloc", st);
    hipLaunchKernelGGL(k_iota, dim3(ceil_div(N, 256)), dim3(256), 0, st, iota.p, N);
    size_t bytes = 0;
    FA_HIP(rocprim::radix_sort_pairs(nullptr, bytes, m.rec_hash.p, sorted_hash.p, iota.p, m.pos_ridx.p, (size_t)N, 0u, 32u, st));
    temp.ensure(bytes + 16);
    FA_HIP(rocprim::radix_sort_pairs(temp.p, bytes, m.rec_hash.p, sorted_hash.p, iota.p, m.pos_ridx.p, (size_t)N, 0u, 32u, st));
    tr.mark("sort", st);
    m.uniq_hash.ensure((size_t)N + 1);
    bytes = 0;
    FA_HIP(rocprim::run_length_encode(nullptr, bytes, sorted_hash.p, (size_t)N, m.uniq_hash.p, counts.p, num_runs.p, st));
    temp.ensure(bytes + 16);
    FA_HIP(rocprim::run_length_encode(temp.p, bytes, sorted_hash.p, (size_t)N, m.uniq_hash.p, counts.p, num_runs.p, st));
    int32_t U = 0;
    num_runs.download(&U, 1, st);
    FA_HIP(hipStreamSynchronize(st));
    m.U = U;
    FA_HIP(hipMemsetAsync(counts.p + U, 0, sizeof(uint32_t), st));
    m.uniq_off.ensure((size_t)U + 2);
    bytes = 0;
    FA_HIP(rocprim::exclusive_scan(nullptr, bytes, counts.p, m.uniq_off.p, 0u, (size_t)U + 1, rocprim::plus<uint32_t>(), st));
    temp.ensure(bytes + 16);
    FA_HIP(rocprim::exclusive_scan(temp.p, bytes, counts.p, m.uniq_off.p, 0u, (size_t)U + 1, rocprim::plus<uint32_t>(), st));
    tr.mark("rle_scan", st);
    // frequency threshold (computeFreqHist): walk the distinct list lengths from the most frequent down
    int64_t to_ignore = (int64_t)((float)(int64_t)U * 0.001f / 100);
    int64_t M = std::min<int64_t>(U, to_ignore + 1);
    // the M largest list lengths, in descending order: from a histogram of the short lists plus the long ones verbatim
    // (k_length_histogram); only an index with more than 65 536 lists of 4 096 positions and more sorts all lengths instead
    std::vector<uint32_t> top;
    top.reserve((size_t)M);
    {
      // (FA_FREQ_OVER_CAP: the tests shrink the room for long lists -- 1 = none fits -- so that the sort is exercised too)
      const uint32_t OVER_CAP = (uint32_t)std::min<uint64_t>(1u << 16, env_u64("FA_FREQ_OVER_CAP", 1u << 16));
      DevBuf<unsigned long long> d_hist;
      DevBuf<uint32_t> d_over;
      d_hist.ensure(FREQ_BINS + 1); d_over.ensure(OVER_CAP);
      FA_HIP(hipMemsetAsync(d_hist.p, 0, (FREQ_BINS + 1) * sizeof(unsigned long long), st));
      uint32_t *d_over_n = (uint32_t *)(d_hist.p + FREQ_BINS);
      hipLaunchKernelGGL(k_length_histogram, dim3((unsigned)std::min<int64_t>(ceil_div(U, 256), 4096)), dim3(256), 0, st, counts.p, (int64_t)U, d_hist.p,
                         d_over.p, OVER_CAP, d_over_n);
      std::vector<unsigned long long> h_hist(FREQ_BINS + 1);
      d_hist.download(h_hist.data(), FREQ_BINS + 1, st);
      FA_HIP(hipStreamSynchronize(st));
      const uint32_t n_over = (uint32_t)(h_hist[FREQ_BINS] & 0xFFFFFFFFu);
      if (n_over < OVER_CAP || n_over == 0) {
        std::vector<uint32_t> h_over(n_over);
        if (n_over) { d_over.download(h_over.data(), n_over, st); FA_HIP(hipStreamSynchronize(st)); }
        std::sort(h_over.begin(), h_over.end(), std::greater<uint32_t>());
        for (uint32_t i = 0; i < n_over && (int64_t)top.size() < M; i++) top.push_back(h_over[i]);
        for (int b = FREQ_BINS - 1; b >= 0 && (int64_t)top.size() < M; b--)
          for (unsigned long long c = h_hist[(size_t)b]; c > 0 && (int64_t)top.size() < M; c--) top.push_back((uint32_t)b);
      } else {
        counts_sorted.ensure((size_t)U);
        bytes = 0;
        FA_HIP(rocprim::radix_sort_keys_desc(nullptr, bytes, counts.p, counts_sorted.p, (size_t)U, 0u, 32u, st));
        temp.ensure(bytes + 16);
        FA_HIP(rocprim::radix_sort_keys_desc(temp.p, bytes, counts.p, counts_sorted.p, (size_t)U, 0u, 32u, st));
        top.resize((size_t)M);
        counts_sorted.download(top.data(), (size_t)M, st);
        FA_HIP(hipStreamSynchronize(st));
      }
    }
    FA_REQUIRE((int64_t)top.size() == M, FA_ERR_INTERNAL, "list-length histogram does not add up");
    for (int64_t i = 0; i < M;) {
      int64_t j = i;
      while (j < M && top[j] == top[i]) j++;
      if (j == M && M < U) break;   // the run continues past the prefix: sum would exceed to_ignore
      if (j < to_ignore) { m.freq_threshold = (int)top[i]; i = j; }
      else if (j == to_ignore) { m.freq_threshold = (int)top[i]; break; }
      else break;
    }
    tr.mark("threshold", st);
    // lookup table at load factor <= 1/2
    m.table_bits = std::max(4, floor_log2(std::max(U, 1)) + 2);
    FA_REQUIRE(m.table_bits <= 31, FA_ERR_UNSUPPORTED, "too many distinct minimizers for the lookup table");
    m.table.ensure((size_t)1 << m.table_bits);
    tr.mark("table_alloc", st);
    FA_HIP(hipMemsetAsync(m.table.p, 0, ((size_t)1 << m.table_bits) * sizeof(uint4), st));
    tr.mark("table_clear", st);
    hipLaunchKernelGGL(k_build_table, dim3(ceil_div(U, 256)), dim3(256), 0, st, m.uniq_hash.p, m.uniq_off.p, (int64_t)U, m.table_bits, m.table.p);
    tr.mark("table", st);
    {
      // window links first (they write the flag bytes whole), then the same-hash links OR their bits in
      const int halo = (int)std::min<int64_t>(((int64_t)m.cmw + 2 + 63) / 64 * 64, WL_HALO_MAX);
      if (halo >= m.cmw + 2)
        hipLaunchKernelGGL(k_window_links<true>, dim3(ceil_div(N, WL_TILE)), dim3(WL_THREADS), (size_t)(WL_TILE + 2 * halo) * sizeof(int32_t), st, m.rec_seq.p,
                           m.rec_wpos.p, m.contig_rec.p, N, m.cmw, halo, m.rec_fwd.p, m.rec_bwd.p, m.rec_flags.p);
      else
        hipLaunchKernelGGL(k_window_links<false>, dim3(ceil_div(N, WL_TILE)), dim3(WL_THREADS), (size_t)(WL_TILE + 2 * halo) * sizeof(int32_t), st, m.rec_seq.p,
                           m.rec_wpos.p, m.contig_rec.p, N, m.cmw, halo, m.rec_fwd.p, m.rec_bwd.p, m.rec_flags.p);
      // (FA_LINK_BLOCK_BITS: the tests shrink the blocks so that a small index has blocks inside a contig AND straddling ones)
      const int shift = (int)std::min<uint64_t>(20, std::max<uint64_t>(2, env_u64("FA_LINK_BLOCK_BITS", 10)));
      const int64_t blocks = (N + ((int64_t)1 << shift) - 1) >> shift;
      blk_lo.ensure((size_t)blocks + 1);
      hipLaunchKernelGGL(k_block_contig, dim3(ceil_div(blocks, 256)), dim3(256), 0, st, m.rec_seq.p, m.contig_rec.p, N, shift, blk_lo.p);
      FA_HIP(hipMemsetAsync(m.rec_prev.p, 0xFF, (size_t)N * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_link_duplicates, dim3(ceil_div(N, 256)), dim3(256), 0, st, sorted_hash.p, m.pos_ridx.p, N, m.rec_seq.p, m.rec_wpos.p,
                         m.contig_rec.p, blk_lo.p, shift, m.cmw, m.rec_prev.p, m.rec_flags.p);
    }
    tr.mark("links", st);
    {
      // padded global coordinate of every record (k_rec_gpos): spans of the contigs, their prefix sums, the low words
      DevBuf<unsigned long long> span, base;
      DevBuf<int32_t> d_wraps;
      span.ensure((size_t)m.C + 2); d_wraps.ensure(1);
      hipLaunchKernelGGL(k_contig_span, dim3(ceil_div(m.C + 1, 256)), dim3(256), 0, st, m.contig_rec.p, m.rec_wpos.p, m.C, m.P.fragment_length, span.p);
      // (the prefix sums on the host: a contig list is small next to the records, and a 64-bit device scan would be the one
      // kernel of the library that needs scratch memory, which the runtime keeps per stream for good)
      std::vector<unsigned long long> h_span((size_t)m.C + 1), h_base((size_t)m.C + 1);
      span.download(h_span.data(), (size_t)m.C + 1, st);
      FA_HIP(hipStreamSynchronize(st));
      unsigned long long run = 0;
      for (int c = 0; c <= m.C; c++) { h_base[(size_t)c] = run; run += h_span[(size_t)c]; }
      base.upload(h_base, st);
      m.rec_gpos.ensure((size_t)N + 4);
      m.wrap_rec.ensure(GPOS_MAX_WRAPS);
      // (FA_GPOS_BITS: the tests shrink the low word so that a small index crosses many of its boundaries)
      m.gpos_bits = (int)std::min<uint64_t>(32, std::max<uint64_t>(12, env_u64("FA_GPOS_BITS", 32)));
      FA_REQUIRE(m.gpos_bits == 32 || (1LL << m.gpos_bits) > 2 * (int64_t)m.P.fragment_length, FA_ERR_INVALID, "FA_GPOS_BITS too small for the fragment length");
      hipLaunchKernelGGL(k_rec_gpos, dim3(ceil_div(N, 256)), dim3(256), 0, st, m.rec_seq.p, m.rec_wpos.p, base.p, N, m.gpos_bits, m.rec_gpos.p, m.wrap_rec.p, d_wraps.p);
      d_wraps.download(&m.n_wraps, 1, st);
      FA_HIP(hipStreamSynchronize(st));
      FA_REQUIRE(m.n_wraps <= GPOS_MAX_WRAPS, FA_ERR_UNSUPPORTED, m.gpos_bits == 32 ? "the index spans more than 2^40 bases (shard the references)"
                                                                                       : "FA_GPOS_BITS: more than 256 boundaries in this index");
    }
#ifdef FA_EXPERIMENTS
    if (fused_l2_enabled()) {
      const size_t words = ((size_t)2 * (size_t)N + 31) / 32 + 4;
      m.ev_bits.ensure(words);
      FA_HIP(hipMemsetAsync(m.ev_bits.p, 0, words * sizeof(uint32_t), st));
      hipLaunchKernelGGL(k_event_bits, dim3(ceil_div(N, 256)), dim3(256), 0, st, m.rec_seq.p, m.rec_bwd.p, m.contig_rec.p, N, m.ev_bits.p);
      m.rec_hf.ensure((size_t)N + 4);
      hipLaunchKernelGGL(k_pack_hf, dim3(ceil_div(N, 256)), dim3(256), 0, st, m.rec_hash.p, m.rec_flags.p, m.rec_prev.p, N, m.rec_hf.p);
    }
#endif
    m.packed_geo = m.cmw + 1 < (1 << GEO_BITS);
    if (m.packed_geo) {
      m.rec_hg.ensure((size_t)N + 4); m.rec_prev16.ensure((size_t)N + 4);
      hipLaunchKernelGGL(k_pack_geometry, dim3(ceil_div(N, 256)), dim3(256), 0, st, m.rec_hash.p, m.rec_prev.p, m.rec_fwd.p, m.rec_bwd.p, m.rec_flags.p, N,
                         m.rec_hg.p, m.rec_prev16.p);
    }
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(st));
    tr.mark("geometry", st);
  } else {
    m.table_bits = 4;
    m.table.ensure(16);
    FA_HIP(hipMemsetAsync(m.table.p, 0, 16 * sizeof(uint4), st));
    m.uniq_hash.ensure(2); m.uniq_off.ensure(2);
    m.rec_gpos.ensure(4); m.wrap_rec.ensure(GPOS_MAX_WRAPS); m.n_wraps = 0;
    FA_HIP(hipMemsetAsync(m.uniq_off.p, 0, 2 * sizeof(uint32_t), st));
    FA_HIP(hipStreamSynchronize(st));
  }
  m.stats.k = m.P.kmer_size;
  m.stats.pid = m.P.percentage_identity;
  m.stats.smax = -1;
}

// ------------------------------------------------------------------------------------------------------------
// query pipeline over the fragment range [f0, f1) of a resident batch
// ------------------------------------------------------------------------------------------------------------
// (Re)builds the LUTs for sketches up to smax.  Called with m.mtx held; calls in flight keep the tables they captured
// (older generations are retired, not freed), so the tables can grow while other workspaces are busy.
static void ensure_luts(fa_mapper &m, int smax) {
  if (m.stats.extend(std::max(smax, 1))) {
    if (m.d_min_hits.p) { m.retired_i32.push_back(std::move(m.d_min_hits)); m.retired_i32.push_back(std::move(m.d_pass)); m.retired_f32.push_back(std::move(m.d_ident)); }
    m.d_min_hits.upload(m.stats.min_hits, m.stream);
    m.d_pass.upload(m.stats.pass_shared, m.stream);
    m.d_ident.upload(m.stats.ident, m.stream);
    FA_HIP(hipStreamSynchronize(m.stream));
  }
}

// seed hits of one fragment sorted in LDS by k_l1 (4 bytes each); more go through HBM scratch
static uint32_t lds_seed_cap_max(int smax) {
  // (the dynamic request of k_l1 -- l1_lds_bytes: seeds, list offsets and sources, six staged locus arrays -- plus its static
  // LDS, a few hundred bytes, must stay within the 160 KB of a CU)
  const int64_t room = 160 * 1024 - 2048 - (int64_t)L1_STAGE * 6 * 4 - std::max<int64_t>(((int64_t)smax + 2) * 8, 1024 * 10) - 64;
  return (uint32_t)std::max<int64_t>(256, room / 4 / 256 * 256);
}


// zero several device ranges with one launch (every DevBuf is at least 16-byte aligned; sizes are rounded up to 16 bytes:
// the buffers listed below are allocated with slack, and the cleared prefix of the status block is a multiple of 16 bytes
// -- asserted next to PassStatus -- so the rounding never reaches the stamps behind it)
struct ClearList {
  ClearArgs a;
  ClearList() { a.count = 0; a.stamp = nullptr; }
  void add(void *p, size_t bytes) {
    if (!bytes) return;
    a.ptr[a.count] = (uint4 *)p; a.n16[a.count] = (bytes + 15) / 16; a.count++;
  }
  void launch(hipStream_t st) {
    if (!a.count) return;
    uint64_t most = 0;
    for (int i = 0; i < a.count; i++) most = std::max(most, a.n16[i]);
    int blocks = (int)std::min<uint64_t>(2048, std::max<uint64_t>(1, (most + 255) / 256));
    hipLaunchKernelGGL(k_clear, dim3(blocks), dim3(256), 0, st, a);
    a.count = 0;
  }
};

// FA_DEBUG_SYNC=1: synchronise after every stage of a query pass and name the stage that failed
static void debug_sync(hipStream_t st, const char *stage) {
  static const bool on = getenv("FA_DEBUG_SYNC") != nullptr;
  if (!on) return;
  fprintf(stderr, "[fa] %s ...\n", stage);
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) throw Error(FA_ERR_NO_DEVICE, std::string("stage ") + stage + " failed: " + hipGetErrorString(e));
}

// fragments mapped per pass (bounds the workspace); FA_PASS_FRAGMENTS overrides it (tests force several passes)
static int64_t pass_fragments() {
  static const int64_t v = [] {
    const char *e = getenv("FA_PASS_FRAGMENTS");
    long long x = e ? atoll(e) : 0;
    return (int64_t)(x > 0 ? x : 48 * 1024);
  }();
  return v;
}

// Workgroup order of k_l2_events over the fragments [f0, f1) of a pass that holds several genomes: fragments sorted by their
// offset inside their genome (then by genome), every group of equal offset dealt to the XCD whose list is the shortest so far
// (equal-sized genomes: group p lands on XCD p mod 8), and the eight lists interleaved the way workgroups are dispatched
// (workgroup b runs on XCD b mod 8); -1 pads the shorter lists.  Returns 0 -- the caller keeps the identity order -- when
// the lists cannot be balanced: fewer groups than XCDs (a batch of plasmids or viral contigs of one or two fragments each
// would put every real workgroup on one XCD) or more than 15 % of padding.
static uint32_t build_frag_order(const fa_genomes &g, int32_t g0, int64_t f0, int64_t f1, std::vector<int32_t> &out) {
  const int64_t F = f1 - f0;
  out.clear();
  int32_t q = g0;
  while (g.genome_frag_lo[q + 1] <= f0) q++;
  if (g.genome_frag_lo[q + 1] >= f1) {
    // ONE genome (round 5): its fragments in eight contiguous runs, one per XCD.  The loci of neighbouring fragments overlap by
    // two thirds on every reference (a locus spans ~2.6 fragment lengths), so the workgroups of a run read the same index
    // stretches -- from their XCD's L2 instead of the memory side, where the identity order (fragment b on XCD b mod 8)
    // had put every neighbour on another XCD.
    if (F < 64) return 0;
    const int64_t run = (F + 7) / 8;
    out.assign((size_t)run * 8, -1);
    for (int64_t i = 0; i < F; i++) out[(size_t)((i % run) * 8 + i / run)] = (int32_t)i;
    return (uint32_t)out.size();
  }
  // offset of every fragment inside its genome, counting sort by it (stable: genomes stay in order inside a group)
  std::vector<int32_t> off((size_t)F);
  int32_t max_off = 0;
  for (int64_t i = 0, qq = q; i < F; i++) {
    while (g.genome_frag_lo[qq + 1] <= f0 + i) qq++;
    off[(size_t)i] = (int32_t)(f0 + i - g.genome_frag_lo[qq]);
    max_off = std::max(max_off, off[(size_t)i]);
  }
  if (max_off + 1 < 8) return 0;
  std::vector<int32_t> start((size_t)max_off + 2, 0);
  for (int64_t i = 0; i < F; i++) start[(size_t)off[(size_t)i] + 1]++;
  for (int32_t p = 0; p <= max_off; p++) start[(size_t)p + 1] += start[(size_t)p];
  std::vector<int32_t> sorted((size_t)F), fill(start.begin(), start.end() - 1);
  for (int64_t i = 0; i < F; i++) sorted[(size_t)fill[(size_t)off[(size_t)i]]++] = (int32_t)i;
  // groups -> XCD lists (the shortest list takes the next group; ties to the lowest XCD), then interleave
  size_t len[8] = {0};
  std::vector<uint8_t> xcd_of((size_t)max_off + 1);
  for (int32_t p = 0; p <= max_off; p++) {
    int x = 0;
    for (int i = 1; i < 8; i++) if (len[i] < len[x]) x = i;
    xcd_of[(size_t)p] = (uint8_t)x;
    len[x] += (size_t)(start[(size_t)p + 1] - start[(size_t)p]);
  }
  const size_t longest = *std::max_element(len, len + 8);
  if ((double)longest * 8.0 > 1.15 * (double)F) return 0;
  out.assign(longest * 8, -1);
  size_t at[8] = {0};
  for (int32_t p = 0; p <= max_off; p++) {
    const size_t x = xcd_of[(size_t)p];
    for (int32_t i = start[(size_t)p]; i < start[(size_t)p + 1]; i++) out[(at[x]++) * 8 + x] = sorted[(size_t)i];
  }
  return (uint32_t)out.size();
}

// One pass of the hot path over genomes [g0, g1) of a resident batch.  Everything between the first kernel and the
// final read-back is asynchronous on one stream: sizes that depend on the data (largest sketch, seed hits per
// fragment, loci, slide events) are *speculated* from earlier passes (fa_mapper::spec), checked on the device, and the
// pass is repeated with larger bounds if a check fails.  Returns the number of rows written at rows_dev[row_base ...].
//
// A pass normally covers all fragments of its genomes at once.  It is cut into *parts* (fragment ranges) when a genome
// alone holds more fragments than pass_fragments(), or when the loci / seeds / slide events of the range exceed what
// the 32-bit offsets of the workspace can address: the parts share the CGI bin table (step 2 of computeCGI is an
// atomicMax, so it simply accumulates) and the rows are formed after the last part.
//
// The pass as an object: what the stages share are its members, and the seams are its methods -- speculation (fetch_spec /
// publish_spec / scan_occupancy), the plan of the pass (plan: buffers, lanes, the bin table), the launches of one part
// (launch_part, launch_rows) and the verdict on a finished part (judge_part); run() queues the parts.
struct QueryPass {
  struct Range { int64_t f0, f1; bool unfused; };     // unfused: the repeat of a range that overflowed k_query_fused
  struct Run { int lane; int64_t f0, f1; fa_mapper::Spec sp; bool with_rows; bool fused = false, forced_unfused = false, ordered = false; };
  // ---- what the caller gave ----
  fa_mapper &m;
  Workspace &w;
  const fa_genomes &g;
  const int32_t g0, g1;
  fa_cgi_row *const rows_dev;
  const int64_t cap, row_base;
  fa_cgi_row *const host_rows;
  // ---- constants of the pass ----
  hipStream_t st;
  const int64_t range_f0, range_f1;
  const int NQ, qcap;
  const IndexView ix;
  const int64_t npairs;
  uint64_t items_max = 0;
  size_t qs_lds = 0;
  int64_t F_total = 0, auto_part = 0;
  Workspace *lanes[3];
  int n_lanes = 1;
  // ---- state of the run ----
  // the speculated bounds are shared by all workspaces: every attempt works on a copy taken under the lock and
  // publishes what it learnt (bounds only ever grow, except the LDS seed slots, which follow the latest pass)
  fa_mapper::Spec sp;
  bool bins_cleared = false;
  std::deque<Range> todo;
  std::deque<Run> flight;
  bool busy[3] = {false, false, false}, ran[3] = {false, false, false};
  bool rows_valid = false;
  int rows_lane = -1;
  unsigned long long t_begin = ~0ULL, t_end = 0;
  int attempts = 0;

  QueryPass(fa_mapper &m_, Workspace &w_, const fa_genomes &g_, int32_t g0_, int32_t g1_, fa_cgi_row *rows_dev_, int64_t cap_, int64_t row_base_, fa_cgi_row *host_rows_)
      : m(m_), w(w_), g(g_), g0(g0_), g1(g1_), rows_dev(rows_dev_), cap(cap_), row_base(row_base_), host_rows(host_rows_), st(w_.stream),
        range_f0(g_.genome_frag_lo[g0_]), range_f1(g_.genome_frag_lo[g1_]), NQ(g1_ - g0_), qcap(m_.qcap), ix(m_.view()), npairs((int64_t)(g1_ - g0_) * m_.G) {
    lanes[0] = &w; lanes[1] = lanes[2] = nullptr;
  }

  // ================================================ speculation ================================================
  void fetch_spec() {
    std::lock_guard<std::mutex> lock(m.mtx);
    fa_mapper::Spec &ms = m.spec;
    if (!ms.init) {
      ms.init = true;
      ms.smax = (int)env_u64("FA_SMAX_INIT", 256);                  // (development: a smaller first guess for small fragments)
      ms.seed_slots = 4096;
      ms.scratch_words = 0;
      ms.l_cap = (int64_t)env_u64("FA_LOCI_CAP_MIN", 1u << 18);   // the tests force the retry path with a tiny value
      ms.items_cap = env_u64("FA_EVENTS_CAP_MIN", 1u << 26);
      ms.part_frags = pass_fragments();
    }
    sp = ms;
    sp.fuse_off = ms.fuse_skip > 0;
    FA_REQUIRE(sp.smax < 32768, FA_ERR_UNSUPPORTED, "query sketch larger than 32767 minimizers");
    ensure_luts(m, sp.smax);
    w.lut_min_hits = m.d_min_hits.p; w.lut_pass = m.d_pass.p; w.lut_ident = m.d_ident.p;
  }
  void publish_spec(const fa_mapper::Spec &sp) {
    std::lock_guard<std::mutex> lock(m.mtx);
    fa_mapper::Spec &ms = m.spec;
    ms.smax = std::max(ms.smax, sp.smax);
    ms.seed_slots = sp.seed_slots;
    ms.scratch_words = std::max(ms.scratch_words, sp.scratch_words);
    ms.items_cap = std::max(ms.items_cap, sp.items_cap);
    ms.l_cap = std::max(ms.l_cap, sp.l_cap);
    ms.part_frags = std::min(ms.part_frags, sp.part_frags);
    ms.redo = ms.redo || sp.redo;
    ms.l1_small_share = sp.l1_small_share; ms.l1_mid_share = sp.l1_mid_share; ms.l1_tiny_share = sp.l1_tiny_share;
    ms.l1_prefilter = ms.l1_prefilter || sp.l1_prefilter; ms.l1_no_small = ms.l1_no_small || sp.l1_no_small;
    ms.l2_loci_last = sp.l2_loci_last;
    ms.smax_misses = std::max(ms.smax_misses, sp.smax_misses);
  }
  // workgroups per CU of the two L2 kernels at a sketch bound (their LDS grows with it), as one number; 0 = not the usual
  // instantiation (wide events, fewer than 64 loci per scan workgroup) or the runtime does not say
  int scan_occupancy(int smax) {
    const int slots = smax + 1;
    if (slots + 1 >= (1 << EvBits<uint16_t>::RANK) || !m.packed_geo) return 0;
    const size_t lds_scan = ((size_t)(slots + 1) * L2_THREADS + 15) / 16 * 16;
    if (lds_scan > 64 * 1024) return 0;
    const int per_window = std::max(1, 2 * m.P.fragment_length / (m.P.window_size + 1));
    const int ev_stage = std::min(2048, std::max(512, (per_window * 11 / 2 + 127) & ~127));
    const size_t lds_ev = ev_sketch_bytes(slots) + (size_t)ev_stage * 2 * (EV_THREADS / 64) + 16;
    if (lds_ev > 64 * 1024) return 0;
    int n_scan = 0, n_ev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n_scan, (const void *)k_l2_scan<uint16_t, uint8_t, 64>, L2_THREADS, lds_scan) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n_ev, (const void *)k_l2_events<uint16_t, true, 1>, EV_THREADS, lds_ev) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    return n_scan > 0 && n_ev > 0 ? n_scan * 64 + n_ev : 0;
  }

  // ================================================ the plan of the pass =======================================
  void plan() {
    // event offsets are 32-bit: at most this many slide events per part (FA_EVENTS_CAP_MAX: the tests force the split)
    items_max = env_u64("FA_EVENTS_CAP_MAX", (1ULL << 32) - 64);
    w.bins.ensure((size_t)NQ * std::max(m.total_bins, 1) + 2);
    w.row_count.ensure((size_t)npairs + 1); w.row_ident.ensure((size_t)npairs + 1);
    w.row_flag.ensure((size_t)npairs + 1); w.row_off.ensure((size_t)npairs + 1);
    qs_lds = (size_t)next_pow2((uint32_t)std::max(qcap, 2)) * 4;
    FA_REQUIRE(qs_lds <= 150 * 1024, FA_ERR_UNSUPPORTED, "fragment_length too large for the LDS fragment sort");

    // ---- the parts of the pass; optionally (FA_QUERY_LANES = 2 or 3) pipelined over *lanes*: sub-workspaces with their own
    // stream and buffers, so that part n + 1 is sketched and looked up while part n slides.  The parts share the CGI bin
    // table (atomicMax); the rows are formed on the lane of the last part, behind the bins of all lanes.  Every part keeps
    // its own speculation verdict: a void part is queued again (and the rows, if they were formed already, are formed
    // again after it).  Measured: no gain (bench step 0.623 / 0.622 / 0.641 ms with 1 / 2 / 3 lanes, 16 queries per
    // launch 237 k / 240 k pairs/s, config 3 2.59 M / 2.55 M) -- every kernel of the path runs its workgroups in one or a
    // few resident rounds, so a third of the fragments takes nearly as long as all of them, and what the lanes add in
    // overlap they lose in occupancy.  Hence one lane by default; profiles/EXPERIMENTS.md.
    static const int lanes_wanted = (int)std::min<uint64_t>(3, std::max<uint64_t>(1, env_u64("FA_QUERY_LANES", 1)));
    static const int64_t lane_min_frags = (int64_t)env_u64("FA_LANE_MIN_FRAGMENTS", 256);
    F_total = range_f1 - range_f0;
    auto_part = F_total;
    if (lanes_wanted > 1 && F_total >= 2 * lane_min_frags) {
      const int64_t parts = std::min<int64_t>(lanes_wanted, F_total / lane_min_frags);
      auto_part = (F_total + parts - 1) / parts;
    }
    if (lanes_wanted > 1 && std::min<int64_t>(auto_part, sp.part_frags) < F_total) {
      for (int i = 0; i + 1 < lanes_wanted; i++) {
        if (!w.sub[i]) w.sub[i].reset(new Workspace());
        Workspace &x = *w.sub[i];
        if (!x.stream) FA_HIP(hipStreamCreate(&x.stream));
        lanes[n_lanes++] = &x;
      }
    }
    w.serial++;
    w.pass_f0 = range_f0; w.pass_F = range_f1 - range_f0;
    for (int i = 0; i < n_lanes; i++) for (int e = 0; e < 6; e++) if (!lanes[i]->ev[e]) FA_HIP(hipEventCreate(&lanes[i]->ev[e]));
    if (!w.ev_bins) FA_HIP(hipEventCreate(&w.ev_bins));
    // the CGI bin table is cleared once per pass: with one lane by the k_clear of the first part launched (a void first
    // part cleared it all the same), with several lanes up front on the first lane's stream, which the others wait for
    bins_cleared = npairs == 0;
    if (n_lanes > 1) {
      if (npairs > 0) FA_HIP(hipMemsetAsync(w.bins.p, 0, (size_t)NQ * std::max(m.total_bins, 1) * sizeof(unsigned long long), st));
      bins_cleared = true;
      FA_HIP(hipEventRecord(w.ev_bins, st));
    }

  }

  // ================================================ launches ===================================================
  // the lane that forms the rows waits for the bins of the parts launched on the other lanes
  void join_lanes(int me) {
    for (int i = 0; i < n_lanes; i++) if (i != me && ran[i]) FA_HIP(hipStreamWaitEvent(lanes[me]->stream, lanes[i]->ev[4], 0));
  }
  // forms the rows; returns true if the kernel also hands the pass over to the host (small passes: its last workgroup does)
  bool launch_rows(Workspace &ln, const PublishArgs &pub) {
    hipStream_t st = ln.stream;
    uint32_t *const d_counters = ln.status.p->counters;
    int32_t *const d_total_rows = &ln.status.p->total_rows;
    RowsArgs ra;
    ra.bins = w.bins.p; ra.genome_bin = m.genome_bin.p; ra.total_bins = m.total_bins; ra.G = m.G; ra.NQ = NQ;
    ra.row_count = w.row_count.p; ra.row_ident = w.row_ident.p;
    // small passes: the last workgroup of k_cgi_rows also forms the rows and hands the pass over -- five launches less.  Every
    // workgroup pays for it with a device-scope fence, which writes its XCD's L2 back: ~1 us each, one after another per XCD.
    // Break-even at 400-800 pairs (profiles/r05_emit_threshold.txt); at the 12-14 000 pairs of a 24-genome chunk against 500
    // references the fences were 0.44 of the 0.52 ms of the kernel.  (FA_ROWS_EMIT_MAX: the measurement's knob, <= 16384)
    static const int64_t emit_max = (int64_t)env_u64("FA_ROWS_EMIT_MAX", 512);
    ra.emit = npairs <= std::min<int64_t>(emit_max, 16384);
    ra.pub = pub;
    if (!ra.emit) ra.pub.seq = 0;
    ra.done = d_counters + 4; ra.query_total_frag = g.d_total_frag + g0; ra.query_id_base = g0;
    ra.rows = rows_dev + row_base; ra.cap = cap - row_base; ra.total_rows = d_total_rows;
    hipLaunchKernelGGL(k_cgi_rows, dim3(ceil_div(npairs, 4)), dim3(256), 0, st, ra);
    if (ra.emit) {
    } else {
      hipLaunchKernelGGL(k_flag_nonzero, dim3(ceil_div(npairs, 256)), dim3(256), 0, st, w.row_count.p, npairs, w.row_flag.p);
      FA_HIP(hipMemsetAsync(w.row_flag.p + npairs, 0, sizeof(int32_t), st));
      exclusive_sum_i32(ln.sk.cub_temp, w.row_flag.p, w.row_off.p, (int)npairs + 1, st);
      FA_HIP(hipMemcpyAsync(d_total_rows, w.row_off.p + npairs, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_emit_rows, dim3(ceil_div(npairs, 256)), dim3(256), 0, st, w.row_count.p, w.row_ident.p, w.row_off.p, m.G,
                         npairs, g.d_total_frag + g0, g0, rows_dev + row_base, cap - row_base);
    }
    return ra.emit != 0;
  }
  // what the stage launches of one part share (sized by size_part)
  struct Part {
    Workspace &ln;
    hipStream_t st;
    const fa_mapper::Spec &sp;
    const int64_t f0, f1, F;
    const int t0, ntiles;
    const int smax;
    const int64_t l_cap;
    struct L1Class { int nt; uint32_t slots, n_lo, n_hi; };   // one launch of k_l1: the fragments with n_lo <= hits <= n_hi
    L1Class l1[3];
    int n_l1 = 0;
    bool l1_prefilter = false;                                // this part drops dead hits before k_l1's block sort
    bool scan_sorted = false;                                 // k_l2_scan takes its loci sorted by stream length (k_l2_order)
    uint32_t seed_slots = 0;                                  // slots of the last class
    bool wide = false;
    LociRegions loci{nullptr, 1, 0};          // regions of the locus numbering of this part
    const int32_t *frag_order = nullptr;      // workgroup order of the part (prepare_order), null = identity
    uint32_t order_len = 0;
    Part(QueryPass &q, Run &r)
        : ln(*q.lanes[r.lane]), st(ln.stream), sp(r.sp), f0(r.f0), f1(r.f1), F(r.f1 - r.f0), t0(q.g.frag_tile_lo[r.f0]),
          ntiles(q.g.frag_tile_lo[r.f1] - q.g.frag_tile_lo[r.f0]), smax(r.sp.smax),
          // (every region of the locus numbering holds at least one locus: a capacity below the number of regions -- only the
          //  FA_LOCI_CAP_MIN hook of the tests gets there -- would number loci beyond the arrays sized and cleared for l_cap)
          l_cap(std::max<int64_t>(r.sp.l_cap, (int64_t)std::min<uint32_t>(LOCI_REGIONS, ev_regions_for(r.f1 - r.f0)))) {}
  };

  // one part: its buffers, then the stages in order, then the hand-over -- all asynchronous on the lane's stream
  void launch_part(Run &r) {
    Part p(*this, r);
    Workspace &ln = p.ln;
    hipStream_t st = p.st;
    const fa_mapper::Spec &sp = p.sp;
    const int64_t f0 = p.f0, F = p.F;
    ln.serial = w.serial;
    if (&ln != &w) FA_HIP(hipStreamWaitEvent(st, w.ev_bins, 0));     // (the bin table is cleared on the first lane's stream)
    ln.last_F = 0; ln.last_f0 = f0; ln.last_loci = 0;                 // (filled in when the part is accepted)
    const int ntiles = p.ntiles;
    // buffers whose size depends only on the geometry of the part
    ln.sk.stage_hash.ensure((size_t)std::max(ntiles, 1) * TILE);
    ln.sk.stage_wpos.ensure((size_t)std::max(ntiles, 1) * TILE);
    ln.sk.tile_count.ensure((size_t)ntiles + 1);
    ln.q_hash.ensure((size_t)F * qcap); ln.q_off.ensure((size_t)F * qcap); ln.q_cnt.ensure((size_t)F * qcap);
    ln.q_size.ensure((size_t)F); ln.n_seeds.ensure((size_t)F); ln.ovf_off.ensure((size_t)F);
    ln.f_loci_lo.ensure((size_t)F); ln.f_loci_n.ensure((size_t)F);
    ln.status.ensure(1);
    if (!ln.h_status) {
      FA_HIP(hipHostMalloc((void **)&ln.h_status, sizeof(PassStatus), hipHostMallocMapped | hipHostMallocCoherent));
      memset(ln.h_status, 0, sizeof(PassStatus));
    }
    // ---- buffers and tables sized by the speculated bounds ----
    const int smax = p.smax;
    const int64_t l_cap = p.l_cap;
    // LDS also holds smax list offsets; the in-place merge keeps at most 32 seeds per thread in registers
    static const int l1_forced = (int)exp_u64("FA_L1_THREADS", 0);
    // k_l1 runs once per size class of fragments (L1Args::n_lo / n_hi): up to 4 096 hits the 256-thread form with 16 hits per
    // thread (4-wave workgroups, eight per CU: a 5 Mb query is one round of workgroups), up to 7 936 the 512-thread form with 16,
    // beyond the 512-thread form with 32 (above 4 096 hits 512 threads measured best: fewer hits per thread shorten every thread's
    // chain of dependent LDS round trips; 1 024 threads pay more for barriers than they gain).  A class the speculated bound
    // (sp.seed_slots: the largest fragment seen, plus a quarter) does not reach is not launched; the last class takes everything
    // above its lower bound, overflow into HBM scratch included.  FA_L1_THREADS = 256 / 512 / 1024 forces ONE launch of that form.
    {
      const uint32_t cap_max = lds_seed_cap_max(smax), need = std::min(sp.seed_slots, cap_max);
      p.n_l1 = 0;
      if (l1_forced) {
        const int nt = l1_forced >= 1024 ? 1024 : (l1_forced >= 512 ? 512 : 256);
        uint32_t slots = std::min(need, (uint32_t)(L1_INPLACE_MAX * nt));
        if (nt == 512 && slots <= 16u * 512u) slots = std::min<uint32_t>(16u * 512u - 256u, cap_max);
        p.l1[p.n_l1++] = Part::L1Class{nt, slots, 0u, 0xFFFFFFFFu};
      } else {
        const uint32_t s_slots = std::min<uint32_t>(need, L1_SMALL_HITS);
        p.l1[p.n_l1++] = Part::L1Class{256, s_slots, 0u, need <= L1_SMALL_HITS ? 0xFFFFFFFFu : s_slots};
        if (need > L1_SMALL_HITS) {
          // (a 512-thread workgroup is eight waves: four of them fill a CU whatever their LDS up to 39 KB, so the block table of
          // l1_block_sort -- a third as many entries as seed slots -- gets all the slots the 16-per-thread form can address)
          // (7 936, not 8 192: the kilobyte goes to the key buffer behind the slots, which then holds the (key, place) pairs of 1 024
          // blocks -- the 4 x 10^8-record index of config 3 adds ~500 chance hits, each a block of its own, to the ~250 blocks of a
          // fragment's relatives -- and four workgroups still fill a CU)
          const uint32_t m_slots = std::min<uint32_t>(L1_MID_HITS, cap_max);
          p.l1[p.n_l1++] = Part::L1Class{512, m_slots, s_slots + 1u, need <= m_slots ? 0xFFFFFFFFu : m_slots};
          if (need > m_slots)
            p.l1[p.n_l1++] = Part::L1Class{512, std::min<uint32_t>(need, (uint32_t)(L1_INPLACE_MAX * 512)), m_slots + 1u, 0xFFFFFFFFu};
        }
      }
      // The pre-filter of the block sort (l1_block_sort: hits that cannot belong to a candidate are dropped before the sort, one more
      // sweep over the position lists) pays where chance hits push the blocks of a fragment beyond what the register sort holds, and
      // costs where they do not (profiles/r06_l1_prefilter.txt: lookup + L1 70.4 -> 60.9 ms on config 3, 211 -> 156 on 2000 x 2000
      // genomes, 38.9 -> 20.1 in the (k = 14, fragment 1000) cell, whose 28-bit hashes collide everywhere; 52.8 -> 66.3 us on the
      // one-query step, 5.4 -> 6.2 ms in the (16, 3000) cell).  So it follows the evidence: on from 3 x 10^8 index records (~400
      // chance hits per fragment), and on any index once an accepted part of this mapper had one fragment in two hundred fall back
      // to the merge (Spec::l1_prefilter, sticky).  FA_L1_PREFILTER = 0 / 1: never / always.
      static const int l1_pf_env = getenv("FA_L1_PREFILTER") ? atoi(getenv("FA_L1_PREFILTER")) : -1;
      p.l1_prefilter = l1_pf_env < 0 ? (m.N >= 300000000LL || sp.l1_prefilter) : l1_pf_env != 0;
      // Which classes get a launch of their own is decided from the shares `seed_totals` counted in the last accepted part (a launch
      // walks every fragment: 1.7 million workgroups that return at once cost config 3 two milliseconds of 67).  Measured (round 6,
      // profiles/r06_l1_classes_ab.txt, r06_l1_prefilter.txt): fragments of ~1 500 hits -- a genome-like index of 200 genomes --
      // take 7.5 ms per step in the 256-thread form and 10.3 in the 512-thread one; config 3's fragments of 3 000-4 000 hits, nine in
      // ten below the 4 096 bound, take 69.2 in the 512-thread form against 70.4 WITHOUT the pre-filter (their ~500 chance hits are
      // blocks of their own, and the 256-thread form sorts 1 024 blocks at most: what overflows takes the merge) and 60.9 against
      // 70.2 WITH it.  So: with the pre-filter the small class is kept when it holds a third of the fragments; without, when half
      // the fragments hold at most HALF its bound.  The middle class is kept from a twentieth of the fragments on (the
      // 32-hits-per-thread form behind it runs two workgroups per CU and is three times slower per fragment), or to carry the
      // small ones.  Ranges stay contiguous from 0: a wrong guess costs time, not results.
      static const float thin_small = getenv("FA_L1_THIN_SMALL") ? (float)atof(getenv("FA_L1_THIN_SMALL")) : -1.0f;
      static const float thin_mid = getenv("FA_L1_THIN_MID") ? (float)atof(getenv("FA_L1_THIN_MID")) : 0.05f;
      if (p.n_l1 >= 2) {
        // S form for the small fragments, or do they ride in the middle form; the middle class stays if it has fragments of its
        // own worth a launch, or small ones to carry
        const bool keep_s = thin_small >= 0.0f ? (sp.l1_small_share < 0.0f || sp.l1_small_share >= thin_small)        // (forced: tests, A/B)
                            : sp.l1_no_small ? false
                            : p.l1_prefilter ? (sp.l1_small_share < 0.0f || sp.l1_small_share >= 0.35f)
                                             : (sp.l1_tiny_share < 0.0f || sp.l1_tiny_share >= 0.5f);
        const bool keep_m = !keep_s || p.n_l1 == 2 || sp.l1_mid_share < 0.0f || sp.l1_mid_share >= thin_mid;
        Part::L1Class c[3];
        int n = 0;
        uint32_t lo = 0u;                                            // lower bound of the next class kept
        if (keep_s) { c[n++] = p.l1[0]; lo = p.l1[0].n_hi + 1u; }
        if (keep_m) { c[n] = p.l1[1]; c[n].n_lo = lo; lo = p.l1[1].n_hi == 0xFFFFFFFFu ? lo : p.l1[1].n_hi + 1u; n++; }
        if (p.n_l1 == 3) { c[n] = p.l1[2]; c[n].n_lo = lo; n++; }
        for (int i = 0; i < n; i++) p.l1[i] = c[i];
        p.n_l1 = n;
      }
      static const bool dbg_l1 = getenv("FA_DEBUG_L1") != nullptr;
      if (dbg_l1) fprintf(stderr, "k_l1 classes: need=%u tiny=%.3f small=%.3f mid=%.3f prefilter=%d -> %d launch(es)\n", need, sp.l1_tiny_share, sp.l1_small_share, sp.l1_mid_share, (int)p.l1_prefilter, p.n_l1);
      p.seed_slots = p.l1[p.n_l1 - 1].slots;                       // "fits LDS" for seed_totals and k_l1_big: the last class's slots
    }
    ln.l_frag.ensure((size_t)l_cap); ln.l_seq.ensure((size_t)l_cap); ln.l_start.ensure((size_t)l_cap); ln.l_end.ensure((size_t)l_cap + 4);
    ln.l_rfirst.ensure((size_t)l_cap); ln.l_rlast.ensure((size_t)l_cap + 4); ln.l_rpart.ensure((size_t)l_cap);
    ln.l_group.ensure((size_t)l_cap); ln.l_shared.ensure((size_t)l_cap); ln.l_pos.ensure((size_t)l_cap);
    ln.group_best.ensure((size_t)l_cap + 2);
    ln.l_beg.ensure((size_t)l_cap); ln.l_end0.ensure((size_t)l_cap); ln.l_last.ensure((size_t)l_cap); ln.l_ndrop.ensure((size_t)l_cap);
    ln.l_nev.ensure((size_t)l_cap); ln.l_ioff.ensure((size_t)l_cap); ln.l_redo.ensure((size_t)l_cap + 4);
    // A wave of k_l2_scan lasts as long as the longest of its 64 slides, and in k_l1's numbering it holds the loci of one fragment --
    // streams of every length the divergences of the index produce (lane utilisation 85 %).  When the last accepted part had loci
    // for a wave per SIMD and more, the scan takes the loci of every region sorted by stream length (profiles/r06_scan_order.txt:
    // L2 stage -8 % on config 3, -9 % on config 4, -10 % on genome-like inputs and in the (16, 1000) cell, -6 % at 16 queries per
    // launch, -1 % on one 5 Mb query).  FA_L2_SCAN_ORDER = 0 / 1: never / always.
    static const int scan_order_env = getenv("FA_L2_SCAN_ORDER") ? atoi(getenv("FA_L2_SCAN_ORDER")) : -1;
    p.scan_sorted = scan_order_env < 0 ? sp.l2_loci_last >= 1024 * 64 : scan_order_env != 0;
    ln.scan_hist.ensure((size_t)2 * LOCI_REGIONS * SCAN_CLASSES);
    if (p.scan_sorted) ln.scan_order.ensure((size_t)l_cap + 64);
    ln.ovf_buf.ensure((size_t)sp.scratch_words + 4);
    // the locus numbering: one region per sixteen fragments (64 at most), each the largest power of two that fits its share
    p.loci.count = ln.status.p->loci_region;
    p.loci.n = std::min<uint32_t>(LOCI_REGIONS, ev_regions_for(F));
    p.loci.shift = (uint32_t)floor_log2((int)std::max<int64_t>(1, l_cap / p.loci.n));
    ln.loci_n = p.loci.n; ln.loci_shift = p.loci.shift;
    ln.l1_pf = p.l1_prefilter; ln.l1_small_class = p.n_l1 > 1 && p.l1[0].nt == 256;
    p.wide = smax + 1 >= (1 << EvBits<uint16_t>::RANK);        // slot = rank + 1 must fit the slot field of the 16-bit event
    ln.items.ensure(((size_t)sp.items_cap + 8) * (p.wide ? 4 : 2));

    launch_sketch_stage(r, p);
    prepare_order(p);
    launch_l1_stage(p);
    launch_l2_stage(r, p);
    launch_cgi_stage_and_hand_over(r, p);
    ran[r.lane] = true;
  }

  // Passes of several genomes run the workgroups of k_l2_events in offset-major, XCD-aware
  // order (build_frag_order); the order is built on the host, cached per (batch, fragment range) and uploaded behind K1.
  void prepare_order(Part &p) {
    Workspace &ln = p.ln;
    static const bool order_on = !(getenv("FA_FRAG_ORDER") && atoi(getenv("FA_FRAG_ORDER")) == 0);
    static const bool order_one = !(getenv("FA_FRAG_ORDER_ONE") && atoi(getenv("FA_FRAG_ORDER_ONE")) == 0);   // (A/B of the one-genome order)
    if (!(order_on && (NQ >= 2 || order_one) && p.F >= 64)) return;
    if (ln.order_batch != g.serial || ln.order_f0 != p.f0 || ln.order_f1 != p.f1) {
      std::vector<int32_t> ord;
      ln.order_len = build_frag_order(g, g0, p.f0, p.f1, ord);       // 0: the lists cannot be balanced, identity order
      if (ln.order_len) {
        ln.pin_order.ensure(ord.size() * sizeof(int32_t));
        memcpy(ln.pin_order.p, ord.data(), ord.size() * sizeof(int32_t));
        ln.frag_order.ensure(ord.size());
        FA_HIP(hipMemcpyAsync(ln.frag_order.p, ln.pin_order.p, ord.size() * sizeof(int32_t), hipMemcpyHostToDevice, p.st));
      }
      ln.order_batch = g.serial; ln.order_f0 = p.f0; ln.order_f1 = p.f1;
    }
    if (ln.order_len) { p.frag_order = ln.frag_order.p; p.order_len = ln.order_len; }
  }

  // K1 (its extra workgroups zero the tables of the part) + per-fragment sort / unique / index lookup
  void launch_sketch_stage(Run &r, Part &p) {
    Workspace &ln = p.ln;
    hipStream_t st = p.st;
    const fa_mapper::Spec &sp = p.sp;
    const int64_t f0 = p.f0, F = p.F;
    const int64_t l_cap = p.l_cap;
    const int t0 = p.t0, ntiles = p.ntiles;
    {
      ClearList cl;
      cl.add(ln.status.p, offsetof(PassStatus, stamp));
      cl.a.stamp = &ln.status.p->stamp[0];
      cl.add(ln.l_end.p, (size_t)l_cap * sizeof(int32_t)); cl.add(ln.l_rlast.p, (size_t)l_cap * sizeof(int32_t));
      cl.add(ln.group_best.p, (size_t)l_cap * sizeof(unsigned long long));
      if (p.scan_sorted) cl.add(ln.scan_hist.p, (size_t)2 * LOCI_REGIONS * SCAN_CLASSES * sizeof(uint32_t));
      if (!bins_cleared) { cl.add(w.bins.p, (size_t)NQ * std::max(m.total_bins, 1) * sizeof(unsigned long long)); bins_cleared = true; }
      QuerySketchArgs a;
      a.frag_tile_lo = g.d_frag_tile_lo + f0;
      a.tile_count = ln.sk.tile_count.p; a.stage_hash = ln.sk.stage_hash.p; a.stage_wpos = ln.sk.stage_wpos.p;
      a.tile_base = t0;                              // frag_tile_lo holds batch-wide tile numbers
      a.q_hash = ln.q_hash.p; a.q_size = ln.q_size.p; a.qcap = qcap;
      a.ix = ix; a.q_off = ln.q_off.p; a.q_cnt = ln.q_cnt.p; a.n_seeds = ln.n_seeds.p;
      a.sort_cap = (int32_t)(qs_lds / 4);
      a.rec_cap = getenv("FA_QF_CAP") ? std::max(0, std::min(QF_CAP, atoi(getenv("FA_QF_CAP")))) : QF_CAP;
      a.exc_tiles = 0;                                 // (set by launch_sketch_tiles for the fused kernel)
      // ---- K1 (its extra workgroups zero the ranges above beside the hashing) + per-fragment sort / unique / index lookup:
      //      one launch where the pass qualifies (k_query_fused), else k_sketch_fast / k_sketch_tiles, then k_query_sketch ----
      const bool fused = launch_sketch_tiles(m.P, g.store, g.tiles + t0, ntiles, ln.sk.stage_hash.p, ln.sk.stage_wpos.p, ln.sk.tile_count.p, st, &cl.a,
                                             (sp.fuse_off || r.forced_unfused) ? nullptr : &a, F);
      r.fused = fused;
      if (!fused) {
        if (qs_lds > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k_query_sketch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qs_lds));
        hipLaunchKernelGGL(k_query_sketch, dim3((unsigned)F), dim3(MAP_THREADS), qs_lds, st, a);
      }
    }
    debug_sync(st, "sketch");
  }

  // seed totals, then the candidate regions
  void launch_l1_stage(Part &p) {
    Workspace &ln = p.ln;
    hipStream_t st = p.st;
    const fa_mapper::Spec &sp = p.sp;
    const int64_t F = p.F;
    const int smax = p.smax;
    const int64_t l_cap = p.l_cap;
    int32_t *const d_stats = ln.status.p->stats;
    uint64_t *const d_totals = ln.status.p->totals;
    uint32_t *const d_counters = ln.status.p->counters;
    unsigned long long *const d_pinfo = ln.status.p->pinfo;
    const uint32_t seed_slots = p.seed_slots;
    // ---- seed totals and speculation checks (the lookup itself is the tail of k_query_sketch).  A kernel of its own
    //      only where k_l1 / k_l1_big need the scratch offsets it produces; else workgroup F of k_l1's launch ----
    const bool fold_totals = sp.scratch_words == 0;
    if (!fold_totals)
      hipLaunchKernelGGL(k_seed_totals, dim3(1), dim3(1024), 0, st, ln.n_seeds.p, F, seed_slots, d_totals, ln.ovf_off.p,
                         d_stats, smax, sp.scratch_words, d_pinfo, &ln.status.p->stamp[1], ln.q_size.p);
    debug_sync(st, "lookup");
    // ---- L1 ----
    {
      L1Args a;
      a.fold_totals = fold_totals ? 1 : 0; a.spec_smax = smax; a.F = F; a.totals = d_totals; a.stats = d_stats;
      a.spec_scratch_words = sp.scratch_words; a.stamp = &ln.status.p->stamp[1];
      a.ix = ix; a.q_size = ln.q_size.p; a.q_off = ln.q_off.p; a.q_cnt = ln.q_cnt.p; a.n_seeds = ln.n_seeds.p;
      a.ovf_off = ln.ovf_off.p; a.ovf_buf = ln.ovf_buf.p; a.min_hits_lut = w.lut_min_hits;
      a.l_frag = ln.l_frag.p; a.l_seq = ln.l_seq.p; a.l_start = ln.l_start.p; a.l_end = ln.l_end.p; a.l_group = ln.l_group.p;
      a.l_rfirst = ln.l_rfirst.p; a.l_rlast = ln.l_rlast.p; a.l_rpart = ln.l_rpart.p;
      a.counters = d_counters; a.loci = p.loci; a.qcap = qcap; a.frag_len = m.P.fragment_length; a.l_cap = (int32_t)l_cap;
      a.lds_seed_cap = seed_slots; a.totals_seed_cap = seed_slots; a.n_lo = 0; a.n_hi = 0xFFFFFFFFu; a.pinfo = d_pinfo; a.lut_smax = smax; a.scratch_words = sp.scratch_words;
      a.f_loci_lo = ln.f_loci_lo.p; a.f_loci_n = ln.f_loci_n.p;
      static const bool l1_block_sort_on = !(getenv("FA_L1_BLOCK_SORT") && atoi(getenv("FA_L1_BLOCK_SORT")) == 0);
      static const bool l1_stats = getenv("FA_L1_STATS") && atoi(getenv("FA_L1_STATS")) != 0;
      // Hits that cannot be an end of a candidate skip the fetch of their padded global coordinate (k_l1, scan_run<., NEAR>).  The
      // hits it saves are the chance hits, whose number grows with the index (~500 per fragment at 4 x 10^8 records, ~50 at
      // 4 x 10^7), and it costs a second LDS read per hit: lookup + L1 70.8 -> 67.0 ms per step on 1000 x 1000 genomes, 28.6 ->
      // 28.6 on 500 x 500, 10.1 -> 10.6 on 200 x 200 (profiles/r05_l1_near_time.txt) -- so it is on from 3 x 10^8 records.
      // FA_L1_NEAR = 0 / 1: never / always (the A/B of the HBM fetch: profiles/r05_l1_near_fetch.txt).
      static const int l1_near_env = getenv("FA_L1_NEAR") ? atoi(getenv("FA_L1_NEAR")) : -1;
      const bool l1_near_on = l1_near_env < 0 ? m.N >= 300000000LL : l1_near_env != 0;
      const bool l1_pf_on = p.l1_prefilter;             // (decided with the size classes, when the part was planned)
      a.block_sort = (l1_block_sort_on ? 1 : 0) | (l1_stats ? 2 : 0) | (l1_near_on ? 4 : 0) | (l1_pf_on ? 8 : 0);
      const uint32_t l1_grid = (uint32_t)F;             // (the offset-major order of k_l2_events applied here measured nothing: 75.9 / 75.5 ms on config 3)
      a.dbg = ln.status.p->dbg;
      // fragments with more hits than LDS holds (seen before on this mapper: scratch is reserved for them) are cut
      // into LDS-sized chunks at contig boundaries by k_l1_big first; what it cannot cut stays with k_l1's HBM path
      static const bool l1_big = !(getenv("FA_L1_BIG") && atoi(getenv("FA_L1_BIG")) == 0);
      a.big_state = nullptr; a.big_enabled = 0; a.big_cap = 0;
      const int64_t big_room = (int64_t)160 * 1024 - 2048 - ((int64_t)smax + 2) * 16;   // LDS left for a chunk's seeds
      if (l1_big && sp.scratch_words > 0 && big_room >= 4 * 2048) {
        ln.big_state.ensure((size_t)F);
        a.big_cap = (uint32_t)std::min<int64_t>((int64_t)L1_BIG_E * L1_BIG_THREADS, big_room / 4 / 256 * 256);
        a.big_state = ln.big_state.p; a.big_enabled = 1;
        const size_t lds = l1_big_lds_bytes(a.big_cap, smax);
        if (lds > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k_l1_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_l1_big, dim3((unsigned)F), dim3(L1_BIG_THREADS), lds, st, a);
      }
      // one launch per size class (Part::L1Class); the seed totals ride in the first one
      auto go = [&](auto nt_tag, const Part::L1Class &c, bool fold) {
        constexpr int NTT = decltype(nt_tag)::value;
        const size_t lds = l1_lds_bytes(c.slots, smax, NTT);
        FA_REQUIRE(lds + 1024 <= 160 * 1024, FA_ERR_UNSUPPORTED, "query sketch too large for the LDS tables of the L1 kernel");
        static const bool dbg = getenv("FA_DEBUG_L1") != nullptr;
        if (dbg) fprintf(stderr, "k_l1<%d>: F=%lld hits %u..%u seed_slots=%u smax=%d lds=%zu\n", NTT, (long long)F, c.n_lo, c.n_hi, c.slots, smax, lds);
        L1Args b = a;
        b.lds_seed_cap = c.slots; b.n_lo = c.n_lo; b.n_hi = c.n_hi; b.fold_totals = fold ? 1 : 0;
        if (c.slots <= 16 * (uint32_t)NTT) {
          if (lds > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k_l1<NTT, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL((k_l1<NTT, 16>), dim3(l1_grid + (fold ? 1u : 0u)), dim3(NTT), lds, st, b);
        } else {
          if (lds > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k_l1<NTT, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL((k_l1<NTT, 32>), dim3(l1_grid + (fold ? 1u : 0u)), dim3(NTT), lds, st, b);
        }
      };
      for (int c = 0; c < p.n_l1; c++) {
        const bool fold = fold_totals && c == 0;
        if (p.l1[c].nt >= 1024) go(std::integral_constant<int, 1024>(), p.l1[c], fold);
        else if (p.l1[c].nt >= 512) go(std::integral_constant<int, 512>(), p.l1[c], fold);
        else go(std::integral_constant<int, 256>(), p.l1[c], fold);
      }
    }
    debug_sync(st, "l1");
  }

  // event streams, then the sequential slide
  void launch_l2_stage(Run &r, Part &p) {
    Workspace &ln = p.ln;
    hipStream_t st = p.st;
    const fa_mapper::Spec &sp = p.sp;
    const int64_t F = p.F;
    const int smax = p.smax;
    const int64_t l_cap = p.l_cap;
    uint64_t *const d_totals = ln.status.p->totals;
    uint32_t *const d_counters = ln.status.p->counters;
    unsigned long long *const d_pinfo = ln.status.p->pinfo;
    const bool wide = p.wide;
    // ---- L2: event streams, then the sequential slide (uint8 state, uint16 redo) ----
    {
      if (m.stage_events) FA_HIP(hipEventRecord(ln.ev[2], st));
      L2Args a;
      a.stamp = &ln.status.p->stamp[2];
      a.ix = ix; a.q_hash = ln.q_hash.p; a.q_size = ln.q_size.p;
      a.l_frag = ln.l_frag.p; a.l_seq = ln.l_seq.p; a.l_start = ln.l_start.p; a.l_end = ln.l_end.p; a.l_group = ln.l_group.p;
      a.l_rfirst = ln.l_rfirst.p; a.l_rlast = ln.l_rlast.p; a.l_rpart = ln.l_rpart.p; a.frag_len = m.P.fragment_length;
      a.l_beg = ln.l_beg.p; a.l_end0 = ln.l_end0.p; a.l_last = ln.l_last.p; a.l_nev = ln.l_nev.p; a.l_ioff = ln.l_ioff.p; a.l_ndrop = ln.l_ndrop.p;
      a.items = ln.items.p; a.items_cap = sp.items_cap; a.pinfo = d_pinfo; a.l_cap = (int32_t)l_cap;
      a.l_shared = ln.l_shared.p; a.l_pos = ln.l_pos.p; a.pass_lut = w.lut_pass; a.group_best = ln.group_best.p;
      a.counters = d_counters; a.loci = p.loci; a.qcap = qcap; a.cmw = m.cmw;
      a.cnt_slots = smax + 1;
      a.rec_total = (unsigned long long *)(d_totals + 3);
      a.ev_region = ln.status.p->ev_region; a.rec_region = ln.status.p->rec_region;
      a.n_regions = ev_regions_for(F);
      a.region_cap = (sp.items_cap / a.n_regions) & ~7ULL;
      a.l_redo = ln.l_redo.p;
      a.redo_count = d_counters + 3;
      {
        // classes of the counting sort: the longest streams hold the records of ~2.6 windows twice (see ev_stage below), so six
        // windows' worth of events over the classes
        const int per_window_ev = std::max(1, 2 * m.P.fragment_length / (m.P.window_size + 1));
        a.scan_class_div = p.scan_sorted ? std::max(8, (per_window_ev * 6 / SCAN_CLASSES + 7) & ~7) : 0;
        a.scan_hist = ln.scan_hist.p; a.scan_cursor = ln.scan_hist.p + LOCI_REGIONS * SCAN_CLASSES;
        a.scan_order = p.scan_sorted ? ln.scan_order.p : nullptr;
      }
      a.f_loci_lo = ln.f_loci_lo.p; a.f_loci_n = ln.f_loci_n.p;
      // several genomes in the pass: the workgroups of k_l2_events in offset-major order (prepare_order)
      a.frag_order = p.frag_order;
      const uint32_t ev_grid = p.frag_order ? p.order_len : (uint32_t)F;
      if (p.frag_order) r.ordered = true;
#ifdef FA_EXPERIMENTS
      static const int fused_dbg = (int)env_u64("FA_FUSED_DEBUG", 0);
      a.dbg = fused_dbg;
#endif
      // events of one locus staged in LDS per wave of k_l2_events (longer streams are stored directly): a stream holds
      // the records of about 2.6 windows twice, minus the first window -- 5.3 windows' worth at the longest in the bench;
      // the LDS this costs decides how many workgroups a CU holds (2048: 6, 1408: 7; 0.41 vs 0.38 ms for the L2 stage)
      static const int ev_stage_env = (int)exp_u64("FA_EV_STAGE", 0);
      const int per_window = std::max(1, 2 * m.P.fragment_length / (m.P.window_size + 1));
      a.ev_stage = ev_stage_env ? (ev_stage_env & ~7) : std::min(2048, std::max(512, (per_window * 11 / 2 + 127) & ~127));
      const size_t ev_lds = ev_sketch_bytes(a.cnt_slots) + (size_t)a.ev_stage * (wide ? 4 : 2) * (EV_THREADS / 64) + 16;
      FA_REQUIRE(ev_lds <= 150 * 1024, FA_ERR_UNSUPPORTED, "query sketch too large for the LDS-staged event kernel");
      // fast pass: one state byte per rank; redo pass: two bytes per rank, only for loci whose counts overflowed
      auto scan_lds = [&](int ln, int bytes) { return ((size_t)(a.cnt_slots + 1) * ln * bytes + 15) / 16 * 16; };
      auto pick_lanes = [&](int bytes) {
        int ln = L2_THREADS;
        while (ln > 1 && scan_lds(ln, bytes) > 144 * 1024) ln >>= 1;   // large sketches (tiny windows): fewer loci per workgroup
        FA_REQUIRE(scan_lds(ln, bytes) <= 160 * 1024, FA_ERR_UNSUPPORTED, "query sketch too large for the LDS-resident L2 state");
        return ln;
      };
      const int lanes8 = pick_lanes(1), lanes16 = pick_lanes(2);
      static const size_t scan_pad = (size_t)exp_u64("FA_SCAN_LDS_PAD", 0);   // experiment: fewer scan workgroups per CU
      const size_t lds8 = scan_lds(lanes8, 1) + scan_pad, lds16 = scan_lds(lanes16, 2);
      // (workgroup b of a scan takes chunk b / n of region b mod n: every region needs its chunks, however few loci it can hold)
      auto scan_grid = [&](int lanes) { return (unsigned)(p.loci.n * (uint32_t)ceil_div((int64_t)1 << p.loci.shift, lanes)); };
      // the rank structure of k_l2_events (its template parameter RK): 1 = occupancy words (round 6), FA_EV_RANK=0 = the bucket
      // table + four-entry probe of rounds 2-5 (kept for the A/B and for the unpacked record layout)
      static const bool ev_rank_occ = !(getenv("FA_EV_RANK") && atoi(getenv("FA_EV_RANK")) == 0);
      auto launch = [&](auto ev_kernel, auto scan8, auto scan8_rt, auto scan16, auto scan16_rt) {
        if (ev_lds > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)ev_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ev_lds));
        hipLaunchKernelGGL(ev_kernel, dim3(ev_grid), dim3(EV_THREADS), ev_lds, st, a);
        debug_sync(st, "l2 events");
        if (p.scan_sorted)
          hipLaunchKernelGGL(k_l2_order, dim3((unsigned)(p.loci.n * (uint32_t)ceil_div((int64_t)1 << p.loci.shift, 256))), dim3(256), 0, st, a);
        // the number of loci is only known on the device: launch for the capacity, surplus workgroups exit at once
        a.lanes = lanes8;
        if (lanes8 == L2_THREADS) {
          if (lds8 > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)scan8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
          hipLaunchKernelGGL(scan8, dim3(scan_grid(lanes8)), dim3(L2_THREADS), lds8, st, a);
        } else {
          if (lds8 > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)scan8_rt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
          hipLaunchKernelGGL(scan8_rt, dim3(scan_grid(lanes8)), dim3(L2_THREADS), lds8, st, a);
        }
        // the wide-state pass is only launched once some locus has needed it (a part that finds out too late is repeated)
        if (!sp.redo) return;
        a.lanes = lanes16;
        if (lanes16 == L2_THREADS) {
          if (lds16 > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)scan16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
          hipLaunchKernelGGL(scan16, dim3(scan_grid(lanes16)), dim3(L2_THREADS), lds16, st, a);
        } else {
          if (lds16 > 64 * 1024) FA_HIP(hipFuncSetAttribute((const void *)scan16_rt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
          hipLaunchKernelGGL(scan16_rt, dim3(scan_grid(lanes16)), dim3(L2_THREADS), lds16, st, a);
        }
      };
#ifdef FA_EXPERIMENTS
      // FA_L2_FUSED=1 selects the fused form (events generated into an LDS ring and consumed in place: no event arena in
      // HBM, L2-stage traffic 0.45 GB instead of 1.08 GB per bench step).  It is bit-exact but measured SLOWER than the
      // two-kernel form on the bench step (0.81 ms against 0.42 ms, profiles/EXPERIMENTS.md): one or two producer waves per 64
      // loci cannot hide their LDS round trips the way the 32 waves per CU of k_l2_events do, and the slide state (16-18 KB
      // per 64 loci) leaves no LDS for more.  Kept as an experiment; the default is k_l2_events + k_l2_scan.
      static const bool fused_on = fused_l2_enabled();
      // ring rows of 16 events per locus unless the 2 KB a row of 8 saves buy one more workgroup per CU AND the
      // fragments of this pass then fit the chip in one round (the slide is a latency-bound chain: rounds add up)
      static const int fu_forced = (int)env_u64("FA_FUSED_C", 0);
      const int evb = wide ? 4 : 2;
      auto wg_bytes = [&](int c) { return (fused_lds_bytes<uint8_t>(a.cnt_slots, evb, c) + FU_STATIC_LDS + 511) / 512 * 512; };
      auto per_cu = [&](int c) { return (int64_t)((160 * 1024) / wg_bytes(c)); };
      int fu_c = 16;
      if (fu_forced == 8 || fu_forced == 16) fu_c = fu_forced;
      else if (per_cu(8) > per_cu(16) && F > per_cu(16) * 256 && F <= per_cu(8) * 256) fu_c = 8;
      static const size_t fused_pad = (size_t)env_u64("FA_FUSED_LDS_PAD", 0);   // development: unused LDS behind the ring
      const size_t fl8 = fused_lds_bytes<uint8_t>(a.cnt_slots, evb, fu_c) + fused_pad, fl16 = fused_lds_bytes<uint16_t>(a.cnt_slots, evb, fu_c);
      const bool fused = fused_on && fl16 <= 150 * 1024 && m.cmw < 65535;   // rec_hf keeps same-hash distances in 16 bits
      static const int fu_nprod = (int)env_u64("FA_FUSED_PRODUCERS", 2) == 1 ? 1 : 2;
      const int fu_threads = 64 * (1 + fu_nprod);
      auto launch_fused = [&](auto k8, auto k16) {
        if (fl8 > 60 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl8));
        const unsigned grid = (unsigned)(((F + 7) / 8) * 8);                // blockIdx -> (XCD, fragment of that XCD)
        static const bool dbg_launch = getenv("FA_DEBUG_FUSED") != nullptr;
        if (dbg_launch) {
          int nb = -1;
          (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k8, fu_threads, fl8);
          hipFuncAttributes fa_;
          (void)hipFuncGetAttributes(&fa_, (const void *)k8);
          fprintf(stderr, "k_l2_fused: F=%lld grid=%u lds=%zu (+%zu static) fu_c=%d cnt_slots=%d redo=%d | occupancy API %d blocks/CU, numRegs %d, sharedSizeBytes %zu, maxDynShared %d\n",
                  (long long)F, grid, fl8, FU_STATIC_LDS, fu_c, a.cnt_slots, (int)sp.redo, nb, fa_.numRegs, fa_.sharedSizeBytes, fa_.maxDynamicSharedSizeBytes);
        }
        hipLaunchKernelGGL(k8, dim3(grid), dim3(fu_threads), fl8, st, a, F);
        if (!sp.redo) return;
        if (fl16 > 60 * 1024) FA_HIP(hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl16));
        hipLaunchKernelGGL(k16, dim3(grid), dim3(fu_threads), fl16, st, a, F);
      };
#else
      const bool fused = false;
#endif
      const bool pk = m.packed_geo && !getenv("FA_NO_PACKED_GEO");
      if (fused) {
#ifdef FA_EXPERIMENTS
        auto pick = [&](auto ev_tag) {
          using EV = decltype(ev_tag);
          if (fu_nprod == 2) {
            if (fu_c == 8) launch_fused(k_l2_fused<EV, uint8_t, false, 8, 2>, k_l2_fused<EV, uint16_t, true, 8, 2>);
            else launch_fused(k_l2_fused<EV, uint8_t, false, 16, 2>, k_l2_fused<EV, uint16_t, true, 16, 2>);
          } else {
            if (fu_c == 8) launch_fused(k_l2_fused<EV, uint8_t, false, 8, 1>, k_l2_fused<EV, uint16_t, true, 8, 1>);
            else launch_fused(k_l2_fused<EV, uint8_t, false, 16, 1>, k_l2_fused<EV, uint16_t, true, 16, 1>);
          }
        };
        if (wide) pick(uint32_t()); else pick(uint16_t());
        w.last_ms[14] = (float)smax; w.last_ms[15] = (float)fu_c;
#endif
      } else if (wide) {
        if (pk && ev_rank_occ) launch(k_l2_events<uint32_t, true, 1>, k_l2_scan<uint32_t, uint8_t, 64>, k_l2_scan<uint32_t, uint8_t, 0>, k_l2_scan<uint32_t, uint16_t, 64>, k_l2_scan<uint32_t, uint16_t, 0>);
        else if (pk) launch(k_l2_events<uint32_t, true, 0>, k_l2_scan<uint32_t, uint8_t, 64>, k_l2_scan<uint32_t, uint8_t, 0>, k_l2_scan<uint32_t, uint16_t, 64>, k_l2_scan<uint32_t, uint16_t, 0>);
        else launch(k_l2_events<uint32_t, false, 0>, k_l2_scan<uint32_t, uint8_t, 64>, k_l2_scan<uint32_t, uint8_t, 0>, k_l2_scan<uint32_t, uint16_t, 64>, k_l2_scan<uint32_t, uint16_t, 0>);
      } else {
        if (pk && ev_rank_occ) launch(k_l2_events<uint16_t, true, 1>, k_l2_scan<uint16_t, uint8_t, 64>, k_l2_scan<uint16_t, uint8_t, 0>, k_l2_scan<uint16_t, uint16_t, 64>, k_l2_scan<uint16_t, uint16_t, 0>);
        else if (pk) launch(k_l2_events<uint16_t, true, 0>, k_l2_scan<uint16_t, uint8_t, 64>, k_l2_scan<uint16_t, uint8_t, 0>, k_l2_scan<uint16_t, uint16_t, 64>, k_l2_scan<uint16_t, uint16_t, 0>);
        else launch(k_l2_events<uint16_t, false, 0>, k_l2_scan<uint16_t, uint8_t, 64>, k_l2_scan<uint16_t, uint8_t, 0>, k_l2_scan<uint16_t, uint16_t, 64>, k_l2_scan<uint16_t, uint16_t, 0>);
      }
    }
    debug_sync(st, "l2 scan");
    if (m.stage_events) FA_HIP(hipEventRecord(ln.ev[3], st));
  }

  // core-genome identity and the one hand-over of the part
  void launch_cgi_stage_and_hand_over(Run &r, Part &p) {
    Workspace &ln = p.ln;
    hipStream_t st = p.st;
    const fa_mapper::Spec &sp = p.sp;
    const int64_t f0 = p.f0;
    const int64_t l_cap = p.l_cap;
    uint32_t *const d_counters = ln.status.p->counters;
    // ---- core-genome identity ----
    if (npairs > 0) {
      CgiArgs a;
      a.stamp = &ln.status.p->stamp[3];
      a.ix = ix; a.group_best = ln.group_best.p; a.counters = d_counters; a.l_frag = ln.l_frag.p; a.l_seq = ln.l_seq.p;
      a.l_pos = ln.l_pos.p; a.q_size = ln.q_size.p; a.ident_lut = w.lut_ident;
      a.frag_query = g.d_frag_query + f0; a.frag_qseq = g.d_frag_qseq + f0; a.bins = w.bins.p;
      a.bin_len = m.P.fragment_length - 20;
      a.query_base = g0;                             // frag_query holds batch-wide genome numbers
      a.wide_launched = sp.redo ? 1 : 0;
      a.group_bound = p.loci.n << p.loci.shift;
      hipLaunchKernelGGL(k_cgi_bins, dim3(ceil_div(l_cap, 256)), dim3(256), 0, st, a);
    }
    // ---- the one hand-over of the part: results, statistics and the speculation verdict (publish_pass) ----
    PassStatus *h_dev = nullptr;
    FA_HIP(hipHostGetDevicePointer((void **)&h_dev, ln.h_status, 0));
    const bool to_host = r.with_rows && host_rows != nullptr;
    const PublishArgs pub = publish_args(ln.status.p, h_dev, ++ln.seq, rows_dev + row_base, to_host ? host_rows + row_base : nullptr, cap - row_base);
    bool published = false;
    if (npairs > 0 && r.with_rows) { join_lanes(r.lane); published = launch_rows(ln, pub); rows_lane = r.lane; rows_valid = true; }
    FA_HIP(hipGetLastError());
    debug_sync(st, "cgi");
    if (n_lanes > 1) FA_HIP(hipEventRecord(ln.ev[4], st));         // (join_lanes: the bins of this part)
    if (!published) hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(256), 0, st, pub);
  }

  // ================================================ the verdict on a part ======================================
  // waits for a part and reads its verdict: true = accepted, false = void (its range has to run again)
  bool judge_part(Run &r) {
    Workspace &ln = *lanes[r.lane];
    fa_mapper::Spec &sp = r.sp;
    const int64_t F = r.f1 - r.f0;
    wait_published(ln.h_status, ln.seq, ln.stream);
    const int32_t *h_stats = ln.h_status->stats;
    const uint64_t *h_totals = ln.h_status->totals;
    const uint32_t *h_counters = ln.h_status->counters;
    const unsigned long long *h_pinfo = ln.h_status->pinfo;
    const uint64_t total_seeds = h_totals[0], max_seeds = h_totals[1];
    uint64_t events_total = h_pinfo[0], records_total = h_totals[3], ev_region_max = 0;
    for (int i = 0; i < EV_REGIONS; i++) {
      events_total += ln.h_status->ev_region[i]; records_total += ln.h_status->rec_region[i];
      ev_region_max = std::max<uint64_t>(ev_region_max, ln.h_status->ev_region[i]);
    }
    const unsigned long long flags = h_pinfo[1];
    // loci: reserved per region (LociRegions); a region asked for more than it holds = SPEC_LOCI
    uint64_t loci_total = 0, loci_region_max = 0;
    for (uint32_t i = 0; i < ln.loci_n; i++) {
      const uint64_t c = ln.h_status->loci_region[i];
      loci_region_max = std::max(loci_region_max, c);
      loci_total += std::min<uint64_t>(c, 1ULL << ln.loci_shift);
    }
    // a part whose seeds / loci / slide events cannot be addressed with 32-bit offsets is cut down and run again
    if (flags || (h_counters[3] > 0 && !sp.redo)) w.last_ms[9] += 1.0f;   // repeated attempts of this call (speculation misses)
    auto shrink_part = [&](double have, double limit, const char *what) {
      FA_REQUIRE(F > 1, FA_ERR_UNSUPPORTED, std::string("a single query fragment produces too many ") + what);
      sp.part_frags = std::max<int64_t>(1, std::min<int64_t>(F / 2, (int64_t)((double)F * limit / have * 0.8)));
    };
    if (total_seeds >= (1ULL << 31)) { shrink_part((double)total_seeds, 2147483648.0, "seed hits"); publish_spec(sp); return false; }
    // bounds for the next pass (or the repeat of this one)
    // (a multiple of 8 just above the largest sketch seen: every slot of the bound costs k_l2_scan 64 bytes of LDS per wave,
    // and on the bench workload -- largest sketch 263 -- 272 slots let nine of its workgroups share a CU where 288 let eight)
    // The first growth is that tight bound; if the workload keeps producing larger sketches (every raise voids a pass and
    // rebuilds the O(s^2) LUTs) later ones take an eighth of headroom, cut back to the largest bound that leaves k_l2_scan and
    // k_l2_events the workgroups per CU the tight bound would (scan_occupancy).
    if (h_stats[0] > sp.smax) {
      const int tight = (h_stats[0] + 4 + 7) / 8 * 8;
      int bound = tight;
      if (sp.smax_misses > 0) {
        const int roomy = (h_stats[0] + h_stats[0] / 8 + 7) / 8 * 8;
        const int want = scan_occupancy(tight);
        bound = want > 0 ? tight : roomy;
        for (int s2 = tight + 8; want > 0 && s2 <= roomy && scan_occupancy(s2) == want; s2 += 8) bound = s2;
      }
      sp.smax = bound;
      sp.smax_misses++;
    }
    // LDS slots for the seed sort: a quarter of headroom over the largest fragment seen, (LDS per workgroup sets how many fragments a CU works on at once)
    uint32_t want_slots = std::min<uint32_t>(lds_seed_cap_max(sp.smax), std::max<uint32_t>(1024, (uint32_t)((std::min<uint64_t>(max_seeds + max_seeds / 4, 1u << 30) + 255) / 256 * 256)));
    const bool slots_changed = want_slots != sp.seed_slots;
    if (flags & SPEC_SCRATCH) sp.scratch_words = std::max<uint64_t>(sp.scratch_words, h_totals[2] + h_totals[2] / 4);
    if (flags & SPEC_LOCI) {
      // every region has to hold its share: size the arrays for the fullest one.  A region holds the largest power of two
      // below its share of l_cap, i.e. more than half of it: twice the need is what makes the repeat fit for certain
      const int64_t need = (int64_t)(loci_region_max * ln.loci_n);
      const int64_t want = std::max<int64_t>(sp.l_cap * 2, need * 2);
      const int64_t l_max = (1LL << 31) - 64;
      // a region holds the largest power of two below its share: at the cap that is 2^floor_log2(l_max / n), which the fullest
      // region must fit -- otherwise the repeat would overflow again at the same capacity, for ever
      const int64_t region_at_cap = (int64_t)1 << floor_log2((int)std::max<int64_t>(1, l_max / (int64_t)ln.loci_n));
      if (need > l_max || (want >= l_max && (int64_t)loci_region_max > region_at_cap)) {
        shrink_part((double)loci_region_max, (double)region_at_cap, "candidate loci"); publish_spec(sp); return false;
      }
      sp.l_cap = std::min(want, l_max);
    }
    if (flags & SPEC_QFUSE) {
      r.forced_unfused = true;                           // this range again, through the two kernels
      std::lock_guard<std::mutex> lock(m.mtx);
      m.spec.fuse_penalty = m.spec.fuse_penalty ? std::min(64, m.spec.fuse_penalty * 2) : 1;
      m.spec.fuse_skip = m.spec.fuse_penalty - 1;
    }
    if (flags & SPEC_EVENTS) {
      // every region has to hold its share: size the arena for the fullest one (the fused form reserves from one counter)
      const uint64_t need = std::max<uint64_t>(h_pinfo[0], ev_region_max * ev_regions_for(F));
      if (need > items_max) { shrink_part((double)need, (double)items_max, "slide events"); publish_spec(sp); return false; }
      sp.items_cap = std::min<uint64_t>(items_max, std::max<uint64_t>(sp.items_cap * 2, need + need / 4));
    }
    if (flags) { if (slots_changed && (flags & SPEC_SCRATCH)) sp.seed_slots = want_slots; publish_spec(sp); return false; }   // void part: run it again
    if (h_counters[3] > 0 && !sp.redo) { sp.redo = true; publish_spec(sp); return false; }   // loci overflowed the byte state and the wide pass was not launched
    if (slots_changed) {
      // fragments that do not fit the LDS slots use HBM scratch, which must exist: size it for the new slot count lazily
      sp.seed_slots = want_slots;
    }
    sp.l2_loci_last = (int64_t)loci_total;
    if (F > 0) { sp.l1_small_share = (float)h_stats[1] / (float)F; sp.l1_mid_share = (float)h_stats[2] / (float)F; sp.l1_tiny_share = (float)h_stats[3] / (float)F; }
    // fragments whose hits were too scattered for the block sort (they took the merge, at twice the time): from one in two hundred
    // on, the passes that follow drop the hits that cannot belong to a candidate before the sort (launch_l1_stage)
    // (with the filter on and the 256-thread class in use they are fragments whose kept chance hits -- the hashed bits keep about a
    // third of them -- still overfill that class's table of 1 365 entries: an index of 1.6 x 10^9 records leaves ~700 of 2 000, and
    // 47 % of the fragments of the 4000 x 4000 run fell back; the class is folded into the 512-thread form from then on)
    if (F > 0 && (double)h_counters[0] > 0.005 * (double)F) { if (ln.l1_pf && ln.l1_small_class) sp.l1_no_small = true; sp.l1_prefilter = true; }
    publish_spec(sp);
    // ---- accepted ----
    {
      std::lock_guard<std::mutex> lock(m.mtx);
      if (r.fused) m.spec.fuse_penalty = 0;                                             // the fused form holds again
      else if (sp.fuse_off && !r.forced_unfused && m.spec.fuse_skip > 0) m.spec.fuse_skip--;   // one pass of the back-off served
    }
    w.last_ms[r.fused ? 17 : 18] += 1.0f;
    if (r.ordered) w.last_ms[19] += 1.0f;
    if (m.stage_events) {
      float ev_ms = 0;
      FA_HIP(hipEventSynchronize(ln.ev[3]));
      FA_HIP(hipEventElapsedTime(&ev_ms, ln.ev[2], ln.ev[3]));
      w.last_ms[16] += ev_ms;
    }
    const unsigned long long *stamp = ln.h_status->stamp;              // 100 MHz ticks
    for (int i = 0; i < 4; i++) w.last_ms[i] += (float)((double)(stamp[i + 1] - stamp[i]) * 1e-5);
    t_begin = std::min(t_begin, stamp[0]); t_end = std::max(t_end, stamp[4]);
    ln.last_F = F;
    ln.last_loci = (uint32_t)loci_total;
    for (uint32_t i = 0; i < LOCI_REGIONS; i++)
      ln.last_region_count[i] = i < ln.loci_n ? (uint32_t)std::min<uint64_t>(ln.h_status->loci_region[i], 1ULL << ln.loci_shift) : 0u;
    ln.last_items = events_total;
    w.last_ms[5] += (float)records_total;   // reference records inside the locus ranges of this call (roofline line)
    w.last_ms[6] += (float)loci_total;
    w.last_ms[7] += (float)events_total;  // slide events
    w.last_ms[8] += (float)h_counters[3]; // loci that needed the wide L2 state
    w.last_ms[20] += (float)h_counters[5]; w.last_ms[21] += (float)h_counters[6];   // FA_L1_STATS=1: fragments block-sorted / merged by k_l1
    w.last_ms[22] += (float)(h_counters[0] + h_counters[1]);   // fragments that left k_l1's fast form (exact): merged in LDS, HBM road, k_l1_big
    if (h_counters[5] + h_counters[6] > 0) {
      const double nf = (double)(h_counters[5] + h_counters[6]);
      fprintf(stderr, "[fa] k_l1 phases, shader-clock ticks per fragment (thread 0):");
      for (int i = 0; i < 8; i++) fprintf(stderr, " %.0f", (double)ln.h_status->dbg[i] / nf);
      fprintf(stderr, "  (%u block-sorted, %u merged; one workgroup in 64 sampled; gave up on probes / blocks / counts: %llu %llu %llu; expansion: bitmaps+places %.0f, pairs+bitmaps %.0f, bits %.0f)\n",
              h_counters[5], h_counters[6], ln.h_status->dbg[8], ln.h_status->dbg[9], ln.h_status->dbg[10], (double)ln.h_status->dbg[11] / nf, (double)ln.h_status->dbg[12] / nf,
              (double)ln.h_status->dbg[13] / nf);
    }
    return true;
  }


  // ================================================ the pass ===================================================
  int64_t run() {
    w.last_F = 0; w.last_f0 = range_f0; w.last_loci = 0; w.last_genomes = &g;
    for (int i = 0; i < 6; i++) if (!w.ev[i]) FA_HIP(hipEventCreate(&w.ev[i]));
    if (range_f1 == range_f0) return 0;
    FA_REQUIRE(m.P.fragment_length > 20, FA_ERR_UNSUPPORTED, "fragment_length must exceed 20 (the reference bins by fragment_length - 20)");
    fetch_spec();
    plan();
    todo.push_back(Range{range_f0, range_f1, false});
    while (!todo.empty() || !flight.empty()) {
      // launch on every free lane
      for (int i = 0; i < n_lanes && !todo.empty(); i++) {
        if (busy[i]) continue;
        FA_REQUIRE(attempts < 40 + 4 * (int)(F_total / std::max<int64_t>(1, std::min(auto_part, sp.part_frags)) + 1), FA_ERR_INTERNAL,
                   "query pass did not converge on its buffer sizes");
        attempts++;
        fetch_spec();
        const Range range = todo.front(); todo.pop_front();
        const int64_t f1 = std::min(range.f1, range.f0 + std::max<int64_t>(1, std::min(auto_part, sp.part_frags)));
        if (f1 < range.f1) todo.push_front(Range{f1, range.f1, range.unfused});
        Run r{i, range.f0, f1, sp, todo.empty() && npairs > 0};
        r.forced_unfused = range.unfused;
        if (r.with_rows) rows_valid = false;
        launch_part(r);
        busy[i] = true;
        flight.push_back(r);
      }
      // the oldest part in flight
      Run r = flight.front(); flight.pop_front();
      const bool ok = judge_part(r);
      busy[r.lane] = false;
      if (!ok) {
        todo.push_front(Range{r.f0, r.f1, r.forced_unfused});
        rows_valid = false;                     // (rows formed meanwhile lack this part)
      }
    }
    int64_t nrows = 0;
    if (npairs > 0) {
      if (!rows_valid) {
        // a part was repeated after the rows had been formed: form them again, behind everything
        Workspace &ln = w;
        join_lanes(0);
        FA_HIP(hipMemsetAsync(&ln.status.p->counters[4], 0, sizeof(uint32_t), ln.stream));
        FA_HIP(hipMemsetAsync(&ln.status.p->total_rows, 0, sizeof(int32_t), ln.stream));
        PassStatus *h_dev = nullptr;
        FA_HIP(hipHostGetDevicePointer((void **)&h_dev, ln.h_status, 0));
        const PublishArgs pub = publish_args(ln.status.p, h_dev, ++ln.seq, rows_dev + row_base, host_rows ? host_rows + row_base : nullptr, cap - row_base);
        if (!launch_rows(ln, pub)) hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(256), 0, ln.stream, pub);
        wait_published(ln.h_status, ln.seq, ln.stream);
        rows_lane = 0;
      }
      nrows = lanes[rows_lane]->h_status->total_rows;
    }
    if (t_end > t_begin) w.last_ms[4] += (float)((double)(t_end - t_begin) * 1e-5);   // device wall time of the pass
    FA_REQUIRE(nrows <= cap - row_base, FA_ERR_INVALID, "row buffer too small");
    return nrows;
  }
};

static int64_t run_query_pass(fa_mapper &m, Workspace &w, const fa_genomes &g, int32_t g0, int32_t g1, fa_cgi_row *rows_dev, int64_t cap,
                              int64_t row_base, fa_cgi_row *host_rows = nullptr) {
  return QueryPass(m, w, g, g0, g1, rows_dev, cap, row_base, host_rows).run();
}

static int64_t run_query(fa_mapper &m, Workspace &w, const fa_genomes &g, int32_t first, int32_t count, fa_cgi_row *rows, int64_t cap, bool rows_device) {
  require_device();
  FA_REQUIRE(first >= 0 && count >= 0 && first + count <= g.n_genomes, FA_ERR_INVALID, "genome range out of bounds");
  for (float &x : w.last_ms) x = 0;
  fa_cgi_row *dst = rows;
  if (!rows_device) { w.rows_dev.ensure((size_t)std::max<int64_t>(cap, 1)); dst = w.rows_dev.p; }
  // a call that is ONE pass and returns a modest number of rows to the host gets them written into pinned memory by the
  // pass's last kernel (k_publish_status): no device-to-host copy, no second synchronisation
  const bool one_pass = count > 0 && g.genome_frag_lo[first + count] - g.genome_frag_lo[first] <= pass_fragments();
  fa_cgi_row *host_rows = nullptr;
  if (!rows_device && one_pass && cap <= 65536) {
    w.pin_rows.ensure(std::max<size_t>((size_t)cap * sizeof(fa_cgi_row), 4096));
    FA_HIP(hipHostGetDevicePointer((void **)&host_rows, w.pin_rows.p, 0));
  }
  int64_t nrows = 0;
  int32_t g0 = first;
  while (g0 < first + count) {
    int32_t g1 = g0 + 1;
    while (g1 < first + count && g.genome_frag_lo[g1 + 1] - g.genome_frag_lo[g0] <= pass_fragments()) g1++;
    // frag_query is batch-wide: the bins of a pass are indexed by (genome - g0), handled through the pointer offset below
    nrows += run_query_pass(m, w, g, g0, g1, dst, cap, nrows, host_rows);
    g0 = g1;
  }
  if (!rows_device && nrows) {
    const size_t bytes = (size_t)nrows * sizeof(fa_cgi_row);
    if (!host_rows || nrows > ROWS_INLINE_MAX) {
      // through pinned memory: a device-to-pageable copy of a few KB costs more in staging than the copy itself
      // (host_rows set: the publishing workgroup leaves more than ROWS_INLINE_MAX rows to this copy)
      w.pin_rows.ensure(std::max<size_t>(bytes, 4096));
      FA_HIP(hipMemcpyAsync(w.pin_rows.p, w.rows_dev.p, bytes, hipMemcpyDeviceToHost, w.stream));
      FA_HIP(hipStreamSynchronize(w.stream));
    }
    memcpy(rows, w.pin_rows.p, bytes);
  }
  return nrows;
}

// A query call borrows one workspace of the mapper for its duration (blocks while all are busy): fa_lease.h.
struct WorkspaceLease : Lease<fa_mapper, Workspace> {
  explicit WorkspaceLease(fa_mapper &mm)
      : Lease<fa_mapper, Workspace>((bind_device(mm.device), mm), [](Workspace &w) {
          if (!w.stream && hipStreamCreate(&w.stream) != hipSuccess) {
            (void)hipGetLastError();
            w.stream = nullptr;
            throw Error(FA_ERR_NO_DEVICE, "hipStreamCreate failed");
          }
        }) {}
};

// pack + cut into fragments + tiles + upload.  `reuse` (a batch object whose device buffers are recycled, contents
// replaced) and `pin` (pinned staging memory the image is built in, so that the one upload is a plain DMA) serve the
// one-query-at-a-time call; without them the image is built in pageable memory.
// `packed` (fa_genomes_upload_fasta): contig c is record packed[c] of a file that read_fasta_packed has packed already -- its
// words are copied where `contigs` would be packed; `contigs` is not read then.
// fill_genomes works in place (fa_genomes_reload_fasta: a batch object whose device buffers, pinned image and upload stream are
// recycled from chunk to chunk); `sync_pinned`: wait for the upload although the image is pinned (the image is reused next).
static void fill_genomes(fa_genomes *g, const fa_params &P, hipStream_t st, const void *const *contigs, const int64_t *lengths,
                         const int32_t *contig_genome, int64_t n_contigs, int32_t n_genomes, int width,
                         float *host_ms, PinnedBuf *pin, const PackedRef *packed, bool sync_pinned) {
  require_device();
  const auto t_begin = std::chrono::steady_clock::now();
  auto lap = [&, last = t_begin](int slot) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (host_ms) host_ms[slot] = std::chrono::duration<float, std::milli>(now - last).count();
    last = now;
  };
  FA_REQUIRE(width == 1 || width == 2 || width == 4, FA_ERR_INVALID, "char_width must be 1, 2 or 4");
  g->P = P;
  g->serial = ++g_batch_serial;
  g->n_genomes = n_genomes;
  g->total_fragments.assign(n_genomes, 0); g->total_length.assign(n_genomes, 0); g->n_short.assign(n_genomes, 0);
  g->total_bases = 0;
  HostStore hs;
  hs.protein = P.alphabet_size != 4;
  g->genome_frag_lo.assign((size_t)n_genomes + 1, 0);
  g->frag_tile_lo.clear();
  const int frag = P.fragment_length;
  const int64_t min_len = std::min<int64_t>(std::min(P.window_size, P.kmer_size), frag);
  // pass 1: which contigs are mapped, and how many whole fragments each holds -- this fixes every size of the image
  std::vector<const void *> use_ptr;
  std::vector<PackedRef> use_ref;
  std::vector<int64_t> use_len;
  int64_t F = 0;
  for (int64_t c = 0; c < n_contigs; c++) {
    const int64_t len = lengths[c];
    if (len < min_len) continue;
    const int64_t nfrag = len / frag;
    if (nfrag > 0) {                                                  // the tail past the last whole fragment is never read
      if (packed) use_ref.push_back(PackedRef{packed[c].file, packed[c].rec, nfrag * frag}); else use_ptr.push_back(contigs[c]);
      use_len.push_back(nfrag * frag); F += nfrag;
    }
  }
  const int64_t npos_frag = (int64_t)frag - P.kmer_size + 1;
  const int64_t tiles_per_frag = npos_frag > 0 ? (npos_frag + TILE - 1) / TILE : 0;
  const int64_t ntiles = F * tiles_per_frag;
  FA_REQUIRE(ntiles < (1LL << 31) - 1 && F < (1LL << 31) - 1, FA_ERR_UNSUPPORTED, "too many fragments in one batch");
  const size_t bases = (size_t)HostStore::padded_bases(use_len.data(), (int64_t)use_len.size());
  auto al = [](size_t x) { return (x + 15) / 16 * 16; };
  const size_t o_packed = 0, n_packed = hs.protein ? 0 : al(bases / 4 + 64);
  const size_t o_bytes = o_packed + n_packed, n_bytes = hs.protein ? al(bases + 64) : 0;
  const size_t o_tiles = o_bytes + n_bytes, n_tiles = al((size_t)std::max<int64_t>(ntiles, 1) * sizeof(Tile));
  const size_t o_ftl = o_tiles + n_tiles, n_ftl = al(((size_t)F + 1) * 4);
  const size_t o_fq = o_ftl + n_ftl, n_fq = al((size_t)std::max<int64_t>(F, 1) * 4);
  const size_t o_fs = o_fq + n_fq, n_fs = n_fq;
  const size_t o_tf = o_fs + n_fs, n_tf = al((size_t)std::max(n_genomes, 1) * 4);
  const size_t image_bytes = o_tf + n_tf;
  unsigned char *img;
  if (pin) { pin->ensure(image_bytes); img = pin->p; }
  else { g->host_image.resize(image_bytes); img = g->host_image.data(); }
  StageTrace tr("upload_genomes");
  // the slack behind the packed words / bytes (the sketch kernel's funnel shift reads one word past the end)
  if (!hs.protein) memset(img + o_packed + bases / 4, 0, n_packed - bases / 4); else memset(img + o_bytes + bases, 0, n_bytes - bases);
  if (packed) place_packed(hs, use_ref.data(), (int64_t)use_ref.size(), (uint32_t *)(img + o_packed), img + o_bytes);
  else hs.pack_many(use_ptr.data(), use_len.data(), (int64_t)use_ptr.size(), width, (uint32_t *)(img + o_packed), img + o_bytes);
  tr.mark("pack", st);
  lap(0);
  // pass 2: fragments, tiles and per-genome bookkeeping, in contig order, written straight into the image
  Tile *tiles = (Tile *)(img + o_tiles);
  int32_t *frag_query = (int32_t *)(img + o_fq), *frag_qseq = (int32_t *)(img + o_fs), *tf = (int32_t *)(img + o_tf);
  // (the bookkeeping per contig on this thread, the per-fragment tables and tiles by the host pool: 1.7 M fragments and 5 M
  // tiles for a thousand genomes were one thread's loop)
  struct ContigJob { int64_t si, nfrag, nf0, q0; int32_t gi; };
  std::vector<ContigJob> jobs;
  int32_t cur = 0;
  size_t used = 0;
  int64_t nf = 0;
  for (int64_t c = 0; c < n_contigs; c++) {
    int32_t gi = contig_genome ? contig_genome[c] : 0;
    FA_REQUIRE(gi >= cur && gi < n_genomes, FA_ERR_INVALID, "contig_genome must be non-decreasing and < n_genomes");
    while (cur < gi) { cur++; g->genome_frag_lo[cur] = nf; }
    const int64_t len = lengths[c];
    if (len < min_len) { g->n_short[gi]++; continue; }               // _fastani.pyx:1061-1070
    const int64_t nfrag = len / frag;                                 // :1097
    if (nfrag > 0) {
      jobs.push_back(ContigJob{(int64_t)used++, nfrag, nf, (int64_t)g->total_fragments[gi], gi});   // querySeqId = fragments before + i, :985
      nf += nfrag;
    }
    g->total_fragments[gi] += (uint64_t)nfrag;                        // :1104
    g->total_length[gi] += (uint64_t)len;                             // :1105
    g->total_bases += (uint64_t)(nfrag * frag);
  }
  g->frag_tile_lo.assign((size_t)F + 1, 0);
  g->contig_frag_lo.clear();
  for (const ContigJob &cj : jobs) g->contig_frag_lo.push_back(cj.nf0);
  HostPool::get().parallel_for(jobs.size(), [&](size_t j) {
    const ContigJob &cj = jobs[j];
    for (int64_t i = 0; i < cj.nfrag; i++) {
      const int64_t f = cj.nf0 + i;
      g->frag_tile_lo[(size_t)f] = (int32_t)(f * tiles_per_frag);
      make_tiles_at(tiles + f * tiles_per_frag, hs, hs.seq_off[cj.si] + i * frag, frag, (int)f, P.kmer_size, P.window_size);
      frag_qseq[f] = (int32_t)(cj.q0 + i);
      frag_query[f] = cj.gi;
    }
  });
  const int64_t nt = nf * tiles_per_frag;
  while (cur < n_genomes) { cur++; g->genome_frag_lo[cur] = nf; }
  FA_REQUIRE(nf == F && nt == ntiles, FA_ERR_INTERNAL, "fragment / tile count mismatch while building the batch image");
  g->F = F;
  g->frag_tile_lo[(size_t)F] = (int32_t)nt;
  g->ntiles = ntiles;
  memcpy(img + o_ftl, g->frag_tile_lo.data(), ((size_t)F + 1) * 4);
  for (int i = 0; i < n_genomes; i++) tf[i] = (int32_t)g->total_fragments[i];
  tr.mark("fragments_tiles", st);
  lap(1);
  g->blob.ensure(image_bytes);
  // The one-query call (image in the workspace's pinned block, untouched until the call returns): the sketch stage reads the
  // packed words STRAIGHT from the pinned image over PCIe -- every word is read once, by the workgroup that hashes it, and the
  // kernel is bound by its hashing, so the 1.25 MB travel inside its 67 us instead of in a 28 us copy in front of it; only the
  // tile and fragment tables (read by every stage) are copied.  FA_QUERY_ZERO_COPY=0: the whole image is copied.
  static const bool zero_copy_on = !(getenv("FA_QUERY_ZERO_COPY") && atoi(getenv("FA_QUERY_ZERO_COPY")) == 0);
  const bool zero_copy = zero_copy_on && pin && !sync_pinned && !hs.protein;
  if (zero_copy) FA_HIP(hipMemcpyAsync(g->blob.p + o_tiles, img + o_tiles, image_bytes - o_tiles, hipMemcpyHostToDevice, st));
  else FA_HIP(hipMemcpyAsync(g->blob.p, img, image_bytes, hipMemcpyHostToDevice, st));
  const int64_t n_exc = (int64_t)hs.exc_pos.size();
  if (n_exc) { g->exc_pos.upload(hs.exc_pos, st); g->exc_val.upload(hs.exc_val, st); }
  g->store = StoreView();
  g->store.packed = hs.protein ? nullptr : (const uint32_t *)((zero_copy ? img : g->blob.p) + o_packed);
  g->store.bytes = hs.protein ? (const uint8_t *)(g->blob.p + o_bytes) : nullptr;
  g->store.exc_pos = g->exc_pos.p; g->store.exc_val = g->exc_val.p; g->store.n_exc = n_exc;
  g->tiles = (const Tile *)(g->blob.p + o_tiles);
  g->d_frag_tile_lo = (const int32_t *)(g->blob.p + o_ftl);
  // the CGI bins of a pass are indexed by the genome number relative to the first genome of the pass; passes start at
  // genome boundaries, so the per-fragment genome numbers are batch-wide and the kernel subtracts the pass's first genome
  g->d_frag_query = (const int32_t *)(g->blob.p + o_fq);
  g->d_frag_qseq = (const int32_t *)(g->blob.p + o_fs);
  g->d_total_frag = (const int32_t *)(g->blob.p + o_tf);
  // the staging image must stay untouched until the copy has left it (pageable copies return after staging, pinned ones
  // are asynchronous).  An image in the caller's pinned block (the one-query call: the block belongs to the workspace and
  // is not touched again before the call returns, and the pass runs on this same stream, behind the copy) needs no
  // synchronisation; otherwise one per upload
  if (!pin || sync_pinned) {
    FA_HIP(hipStreamSynchronize(st));
    std::vector<unsigned char, NoInitAlloc<unsigned char>>().swap(g->host_image);
  }
  tr.mark("uploads", st);
  lap(2);
}

static std::unique_ptr<fa_genomes> upload_genomes(const fa_params &P, hipStream_t st, const void *const *contigs, const int64_t *lengths,
                                                  const int32_t *contig_genome, int64_t n_contigs, int32_t n_genomes, int width,
                                                  float *host_ms = nullptr, std::unique_ptr<fa_genomes> reuse = nullptr, PinnedBuf *pin = nullptr,
                                                  const PackedRef *packed = nullptr) {
  std::unique_ptr<fa_genomes> g = reuse ? std::move(reuse) : std::unique_ptr<fa_genomes>(new fa_genomes());
  fill_genomes(g.get(), P, st, contigs, lengths, contig_genome, n_contigs, n_genomes, width, host_ms, pin, packed, false);
  return g;
}

// one genome per FASTA file, every file read + packed by its own task of the host pool (read_fasta_packed_many), then the
// batch image assembled from the packed records and uploaded on the batch's own stream
// genomes [first, first + count) of files packed already (one genome per file) as the batch `g`
static void fill_genomes_from_packed(fa_mapper *m, fa_genomes *g, const PackedFasta *files, int32_t n_paths, bool pinned) {
  std::vector<PackedRef> refs;
  std::vector<int64_t> lens;
  std::vector<int32_t> genome;
  for (int32_t i = 0; i < n_paths; i++)
    for (size_t r = 0; r < files[i].rec_len.size(); r++) { refs.push_back(PackedRef{&files[i], (int64_t)r, files[i].rec_len[r]}); lens.push_back(files[i].rec_len[r]); genome.push_back(i); }
  bind_device(m->device);
  // (its own stream: a batch may be uploaded while another thread maps the previous one, Mapper.query_fasta_stream)
  if (!g->up_stream) FA_HIP(hipStreamCreateWithFlags(&g->up_stream, hipStreamNonBlocking));
  fill_genomes(g, m->P, g->up_stream, nullptr, lens.data(), genome.data(), (int64_t)refs.size(), n_paths, 1, nullptr,
               pinned ? &g->pin_image : nullptr, refs.data(), true);
}
static void fill_genomes_from_fasta(fa_mapper *m, fa_genomes *g, const char *const *paths, int32_t n_paths, bool pinned) {
  std::vector<PackedFasta> files;
  read_fasta_packed_many(paths, (size_t)n_paths, m->P.alphabet_size != 4, files);
  fill_genomes_from_packed(m, g, files.data(), n_paths, pinned);
}

// FASTA files read and packed ONCE, to be used as references AND as queries (an all-vs-all reads every file one time)
struct fa_packed {
  bool protein = false;
  std::vector<PackedFasta> files;
  // fa_packed_append grows `files` (and moves every PackedFasta) under the exclusive lock; the calls that read them
  // (fa_packed_info, fa_sketch_add_packed, fa_genomes_reload_packed -- all run without the GIL) hold it shared for their duration
  std::shared_mutex mtx;
};
// the bookkeeping of fa_sketch_add_fasta_many over files that are packed already (s->mtx held by the caller)
static void sketch_add_packed_files(fa_sketch *s, const PackedFasta *files, int32_t n_paths, int64_t *n_records, int64_t *n_short) {
  std::vector<PackedRef> refs;
  std::vector<int32_t> contig_ids;
  std::vector<uint64_t> lengths;
  std::vector<int32_t> by_file;
  std::vector<int64_t> shorts((size_t)n_paths, 0);
  int64_t counter = s->counter;
  FA_REQUIRE(s->cur_total == 0 || n_paths == 0, FA_ERR_INVALID, "a genome is still open (add_contig without end_genome)");
  for (int32_t i = 0; i < n_paths; i++) {
    uint64_t total = 0;
    for (size_t r = 0; r < files[i].rec_len.size(); r++) {
      const int64_t length = files[i].rec_len[r];
      FA_REQUIRE(length < (1LL << 31), FA_ERR_INVALID, "contig length must be below 2^31");
      if (length >= s->P.window_size && length >= s->P.kmer_size) {      // _fastani.pyx:648
        refs.push_back(PackedRef{&files[i], (int64_t)r, length});
        contig_ids.push_back((int32_t)counter);
      } else {
        shorts[(size_t)i]++;
      }
      total += (uint64_t)(length / s->P.fragment_length) * s->P.fragment_length;   // :680
      counter += 1;                                                                // :683
    }
    lengths.push_back(total);                            // :687
    by_file.push_back((int32_t)counter);                 // :690
  }
  s->pending_contig.reserve(s->pending_contig.size() + contig_ids.size());
  s->lengths.reserve(s->lengths.size() + lengths.size());
  s->seqs_by_file.reserve(s->seqs_by_file.size() + by_file.size());
  if (!refs.empty()) append_packed(s->pending, refs.data(), (int64_t)refs.size());
  s->pending_contig.insert(s->pending_contig.end(), contig_ids.begin(), contig_ids.end());
  s->counter = counter;
  s->lengths.insert(s->lengths.end(), lengths.begin(), lengths.end());
  s->seqs_by_file.insert(s->seqs_by_file.end(), by_file.begin(), by_file.end());
  for (int32_t i = 0; i < n_paths; i++) {
    if (n_records) n_records[i] = (int64_t)files[i].rec_len.size();
    if (n_short) n_short[i] = shorts[(size_t)i];
  }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
// the live loci of a part, region by region (LociRegions): fn(first locus number, count)
template <class Fn>
static void for_each_locus_slice(const Workspace &x, Fn fn) {
  for (uint32_t r = 0; r < x.loci_n && r < (uint32_t)LOCI_REGIONS; r++)
    if (x.last_region_count[r]) fn((size_t)r << x.loci_shift, (size_t)x.last_region_count[r]);
}

extern "C" {

const char *fa_last_error(void) { return g_last_error.c_str(); }
int fa_version(void) { return 100; }
int fa_device_trim(uint64_t *held_bytes) {
  return guarded([&] {
    if (held_bytes) *held_bytes = (uint64_t)DevPool::get().held();
    DevPool::get().trim();
  });
}

int fa_device_count(int *count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  *count = n;
  return FA_OK;
}
int fa_set_device(int device) {
  return guarded([&] { require_device(); FA_HIP(hipSetDevice(device)); g_device.store(device); });
}

int fa_recommended_window_size(double p_value, int k, int alphabet_size, float identity, int fragment_length,
                               uint64_t reference_size, int *window) {
  return guarded([&] { *window = stat_recommended_window(p_value, k, alphabet_size, identity, fragment_length, reference_size); });
}
int fa_estimate_minimum_hits_relaxed(int s, int k, float identity, int *hits) {
  return guarded([&] { *hits = stat_min_hits_relaxed(s, k, identity); });
}
int fa_mapping_identity(int shared, int s, int k, float *identity, float *upper) {
  return guarded([&] { FA_REQUIRE(s > 0, FA_ERR_INVALID, "sketch_size must be positive"); stat_identity(shared, s, k, identity, upper); });
}

// host twin of the device hash (same published algorithm as fa_sketch.hip.h's Murmur)
uint32_t fa_hash(const void *kmer, int len) {
  const uint8_t *d = (const uint8_t *)kmer;
  auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
  auto fmix = [](uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; };
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = 42, h2 = 42;
  int nb = len / 16;
  for (int i = 0; i < nb; i++) {
    uint64_t k1, k2;
    memcpy(&k1, d + 16 * i, 8); memcpy(&k2, d + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl(k1, 31); k1 *= c2; h1 ^= k1; h1 = rotl(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl(k2, 33); k2 *= c1; h2 ^= k2; h2 = rotl(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *t = d + 16 * nb;
  int rem = len & 15;
  uint64_t k1 = 0, k2 = 0;
  for (int j = 0; j < rem; j++) { if (j < 8) k1 |= (uint64_t)t[j] << (8 * j); else k2 |= (uint64_t)t[j] << (8 * (j - 8)); }
  if (rem > 8) { k2 *= c2; k2 = rotl(k2, 33); k2 *= c1; h2 ^= k2; }
  if (rem > 0) { k1 *= c1; k1 = rotl(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len; h1 += h2; h2 += h1; h1 = fmix(h1); h2 = fmix(h2);
  return (uint32_t)(h1 + h2);
}

int fa_sketch_new(const fa_params *params, fa_sketch **out) {
  return guarded([&] {
    validate_params(*params);
    std::unique_ptr<fa_sketch> s(new fa_sketch());
    s->P = *params;
    s->device = g_device.load();
    s->reset_data();
    *out = s.release();
  });
}
void fa_sketch_free(fa_sketch *s) {
  if (!s) return;
  if (s->stream) (void)hipStreamDestroy(s->stream);
  if (s->up_stream) (void)hipStreamDestroy(s->up_stream);
  delete s;
}
int fa_sketch_add_contig(fa_sketch *s, const void *data, int64_t length, int char_width, int *added) {
  return guarded([&] {
    FA_REQUIRE(char_width == 1 || char_width == 2 || char_width == 4, FA_ERR_INVALID, "char_width must be 1, 2 or 4");
    FA_REQUIRE(length >= 0 && length < (1LL << 31), FA_ERR_INVALID, "contig length must be below 2^31");
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    int ok = 0;
    if (length >= s->P.window_size && length >= s->P.kmer_size) {      // _fastani.pyx:648
      s->pending.append(data, char_width, length);
      s->pending_contig.push_back((int32_t)s->counter);
      ok = 1;
    }
    s->cur_total += (uint64_t)(length / s->P.fragment_length) * s->P.fragment_length;   // :680
    s->counter += 1;                                                                     // :683
    if (added) *added = ok;
  });
}
struct fa_fasta { FastaFile f; };
int fa_fasta_open(const char *path, fa_fasta **out) {
  return guarded([&] {
    FA_REQUIRE(path && out, FA_ERR_INVALID, "null argument");
    std::unique_ptr<fa_fasta> h(new fa_fasta());
    h->f.open(path);
    *out = h.release();
  });
}
int fa_fasta_next(fa_fasta *f, int *has_record, const char **id, int64_t *id_length, const unsigned char **seq, int64_t *seq_length) {
  return guarded([&] {
    const bool ok = f->f.next();
    *has_record = ok ? 1 : 0;
    if (!ok) return;
    *id = f->f.id.data(); *id_length = (int64_t)f->f.id.size();
    *seq = f->f.seq.data(); *seq_length = (int64_t)f->f.seq.size();
  });
}
void fa_fasta_close(fa_fasta *f) { delete f; }

int fa_sketch_add_fasta(fa_sketch *s, const char *path, int64_t *n_records, int64_t *n_short) {
  return guarded([&] {
    FA_REQUIRE(path, FA_ERR_INVALID, "null path");
    std::vector<FastaSeq> seqs;
    read_fasta_records(path, seqs);
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    // everything is validated and staged in locals first and committed only after the packer has succeeded: a record that
    // is refused, or an allocation failure half-way, leaves the sketch exactly as it was (the reference keeps `total` in a
    // local for the same reason, _fastani.pyx:618,680)
    std::vector<const void *> ptrs;
    std::vector<int64_t> lens;
    std::vector<int32_t> contig_ids;
    int64_t shorts = 0, counter = s->counter;
    uint64_t total = 0;
    for (auto &q : seqs) {
      const int64_t length = (int64_t)q.size;
      FA_REQUIRE(length < (1LL << 31), FA_ERR_INVALID, "contig length must be below 2^31");
      if (length >= s->P.window_size && length >= s->P.kmer_size) {      // _fastani.pyx:648
        ptrs.push_back(q.data.get()); lens.push_back(length);
        contig_ids.push_back((int32_t)counter);
      } else {
        shorts++;
      }
      total += (uint64_t)(length / s->P.fragment_length) * s->P.fragment_length;   // :680
      counter += 1;                                                                // :683
    }
    s->pending_contig.reserve(s->pending_contig.size() + contig_ids.size());
    s->lengths.reserve(s->lengths.size() + 1);
    s->seqs_by_file.reserve(s->seqs_by_file.size() + 1);
    if (!ptrs.empty()) s->pending.append_many(ptrs.data(), lens.data(), (int64_t)ptrs.size(), 1);
    // commit (nothing below can throw: the vectors have room)
    s->pending_contig.insert(s->pending_contig.end(), contig_ids.begin(), contig_ids.end());
    s->counter = counter;
    s->lengths.push_back(s->cur_total + total);          // :687
    s->cur_total = 0;
    s->seqs_by_file.push_back((int32_t)s->counter);      // :690
    if (n_records) *n_records = (int64_t)seqs.size();
    if (n_short) *n_short = shorts;
  });
}
// Many reference genomes at once from host buffers: contig c belongs to genome contig_genome[c] (non-decreasing); the effect of
// n_genomes x (fa_sketch_add_contig per contig, fa_sketch_end_genome) with ONE call of the packer over all contigs -- the
// per-genome calls wake the host pool a thousand times for a thousand genomes (0.4-0.5 s of `host_pack_s` on config 3).
int fa_sketch_add_genomes(fa_sketch *s, const void *const *contigs, const int64_t *lengths, const int32_t *contig_genome, int64_t n_contigs,
                          int32_t n_genomes, int char_width, int32_t *n_short) {
  return guarded([&] {
    FA_REQUIRE(char_width == 1 || char_width == 2 || char_width == 4, FA_ERR_INVALID, "char_width must be 1, 2 or 4");
    FA_REQUIRE(n_contigs >= 0 && n_genomes >= 0 && (n_contigs == 0 || (contigs && lengths && contig_genome)), FA_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    FA_REQUIRE(s->cur_total == 0 || n_genomes == 0, FA_ERR_INVALID, "a genome is still open (add_contig without end_genome)");
    std::vector<const void *> ptrs;
    std::vector<int64_t> lens;
    std::vector<int32_t> contig_ids, by_file((size_t)n_genomes, 0), shorts((size_t)n_genomes, 0);
    std::vector<uint64_t> totals((size_t)n_genomes, 0);
    int64_t counter = s->counter;
    int32_t cur = 0;
    for (int64_t c = 0; c < n_contigs; c++) {
      const int32_t gi = contig_genome[c];
      FA_REQUIRE(gi >= cur && gi < n_genomes, FA_ERR_INVALID, "contig_genome must be non-decreasing and < n_genomes");
      while (cur < gi) { by_file[(size_t)cur] = (int32_t)counter; cur++; }
      const int64_t length = lengths[c];
      FA_REQUIRE(length >= 0 && length < (1LL << 31), FA_ERR_INVALID, "contig length must be below 2^31");
      if (length >= s->P.window_size && length >= s->P.kmer_size) {      // _fastani.pyx:648
        ptrs.push_back(contigs[c]); lens.push_back(length); contig_ids.push_back((int32_t)counter);
      } else {
        shorts[(size_t)gi]++;
      }
      totals[(size_t)gi] += (uint64_t)(length / s->P.fragment_length) * s->P.fragment_length;   // :680
      counter += 1;                                                                            // :683
    }
    while (cur < n_genomes) { by_file[(size_t)cur] = (int32_t)counter; cur++; }
    s->pending_contig.reserve(s->pending_contig.size() + contig_ids.size());
    s->lengths.reserve(s->lengths.size() + totals.size());
    s->seqs_by_file.reserve(s->seqs_by_file.size() + by_file.size());
    if (!ptrs.empty()) s->pending.append_many(ptrs.data(), lens.data(), (int64_t)ptrs.size(), char_width);
    // commit (nothing below can throw: the vectors have room)
    s->pending_contig.insert(s->pending_contig.end(), contig_ids.begin(), contig_ids.end());
    s->counter = counter;
    s->lengths.insert(s->lengths.end(), totals.begin(), totals.end());                           // :687
    s->seqs_by_file.insert(s->seqs_by_file.end(), by_file.begin(), by_file.end());               // :690
    if (n_short) for (int32_t i = 0; i < n_genomes; i++) n_short[i] = shorts[(size_t)i];
  });
}
// Many reference genomes at once, one per FASTA file, in the order given: the files are read and packed concurrently (one
// task per file), then appended to the pending store exactly as fa_sketch_add_fasta would have added them one by one.
int fa_sketch_add_fasta_many(fa_sketch *s, const char *const *paths, int32_t n_paths, int64_t *n_records, int64_t *n_short) {
  return guarded([&] {
    FA_REQUIRE(paths && n_paths >= 0, FA_ERR_INVALID, "null paths or negative count");
    StageTrace tr("add_fasta_many");
    std::vector<PackedFasta> files;
    read_fasta_packed_many(paths, (size_t)n_paths, s->P.alphabet_size != 4, files);
    tr.mark("read_pack", nullptr);
    if (tr.on) {
      FastaTaskClock &c = fasta_task_clock();
      fprintf(stderr, "[fa trace] fasta tasks so far: %llu files, per-task sums: read %.1f ms, record search %.1f ms, pack %.1f ms\n", (unsigned long long)c.files.load(),
              c.read_ns.load() * 1e-6, c.scan_ns.load() * 1e-6, c.pack_ns.load() * 1e-6);
    }
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    tr.mark("bookkeeping", nullptr);
    sketch_add_packed_files(s, files.data(), n_paths, n_records, n_short);
    tr.mark("place", nullptr);
  });
}
// ---- files packed once, used many times (fa_packed) ----
int fa_packed_read(const char *const *paths, int32_t n_paths, int protein, fa_packed **out) {
  return guarded([&] {
    FA_REQUIRE(paths && out && n_paths >= 0, FA_ERR_INVALID, "null argument or negative count");
    std::unique_ptr<fa_packed> p(new fa_packed());
    p->protein = protein != 0;
    read_fasta_packed_many(paths, (size_t)n_paths, p->protein, p->files);
    *out = p.release();
  });
}
// more files behind the ones the set holds (read + packed concurrently, like fa_packed_read); all or nothing
int fa_packed_append(fa_packed *p, const char *const *paths, int32_t n_paths) {
  return guarded([&] {
    FA_REQUIRE(p && paths && n_paths >= 0, FA_ERR_INVALID, "null argument or negative count");
    std::vector<PackedFasta> more;
    read_fasta_packed_many(paths, (size_t)n_paths, p->protein, more);
    std::unique_lock<std::shared_mutex> grow(p->mtx);
    p->files.reserve(p->files.size() + more.size());
    for (auto &f : more) p->files.push_back(std::move(f));
  });
}
void fa_packed_free(fa_packed *p) { delete p; }
int fa_packed_info(fa_packed *p, int32_t *n_files, uint64_t *file_bytes, int64_t *records, int64_t *bases) {
  return guarded([&] {
    std::shared_lock<std::shared_mutex> hold(p->mtx);
    if (n_files) *n_files = (int32_t)p->files.size();
    for (size_t i = 0; i < p->files.size(); i++) {
      if (file_bytes) file_bytes[i] = (uint64_t)p->files[i].file_bytes;
      if (records) records[i] = (int64_t)p->files[i].rec_len.size();
      if (bases) { int64_t b = 0; for (int64_t l : p->files[i].rec_len) b += l; bases[i] = b; }
    }
  });
}
int fa_sketch_add_packed(fa_sketch *s, fa_packed *p, int32_t first, int32_t count, int64_t *n_records, int64_t *n_short) {
  return guarded([&] {
    FA_REQUIRE(p, FA_ERR_INVALID, "null packed set");
    std::shared_lock<std::shared_mutex> hold(p->mtx);
    FA_REQUIRE(first >= 0 && count >= 0 && (size_t)first + (size_t)count <= p->files.size(), FA_ERR_INVALID, "file range outside the packed set");
    FA_REQUIRE(p->protein == (s->P.alphabet_size != 4), FA_ERR_INVALID, "the files were packed for the other alphabet");
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    sketch_add_packed_files(s, p->files.data() + first, count, n_records, n_short);
  });
}
int fa_genomes_reload_packed(fa_mapper *m, fa_genomes *g, fa_packed *p, int32_t first, int32_t count) {
  return guarded([&] {
    FA_REQUIRE(g && p, FA_ERR_INVALID, "null argument");
    std::shared_lock<std::shared_mutex> hold(p->mtx);
    FA_REQUIRE(first >= 0 && count >= 0 && (size_t)first + (size_t)count <= p->files.size(), FA_ERR_INVALID, "file range outside the packed set");
    FA_REQUIRE(p->protein == (m->P.alphabet_size != 4), FA_ERR_INVALID, "the files were packed for the other alphabet");
    try {
      fill_genomes_from_packed(m, g, p->files.data() + first, count, true);
    } catch (...) {
      g->n_genomes = 0; g->F = 0; g->ntiles = 0;
      g->genome_frag_lo.assign(1, 0); g->total_fragments.clear(); g->total_length.clear(); g->n_short.clear();
      throw;
    }
  });
}
int fa_sketch_end_genome(fa_sketch *s) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    s->lengths.push_back(s->cur_total);                  // :687
    s->cur_total = 0;
    s->seqs_by_file.push_back((int32_t)s->counter);      // :690
  });
}
int fa_sketch_abort_genome(fa_sketch *s) {
  return guarded([&] { std::lock_guard<std::mutex> lock(s->mtx); s->cur_total = 0; });
}
int fa_sketch_clear(fa_sketch *s) {
  return guarded([&] { std::lock_guard<std::mutex> lock(s->mtx); s->reset_data(); });
}
int fa_sketch_num_minimizers(fa_sketch *s, int64_t *n) {
  return guarded([&] { std::lock_guard<std::mutex> lock(s->mtx); s->flush(); *n = s->nrec; });
}
int fa_sketch_get_minimizers(fa_sketch *s, uint32_t *hash, int32_t *seq_id, int32_t *wpos) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    s->flush();
    if (s->nrec == 0) return;
    s->rec_hash.download(hash, (size_t)s->nrec, s->stream);
    s->rec_seq.download(seq_id, (size_t)s->nrec, s->stream);
    s->rec_wpos.download(wpos, (size_t)s->nrec, s->stream);
    FA_HIP(hipStreamSynchronize(s->stream));
  });
}
int fa_sketch_num_genomes(fa_sketch *s, int64_t *n) {
  return guarded([&] { *n = (int64_t)s->lengths.size(); });
}
int fa_sketch_get_state(fa_sketch *s, uint64_t *lengths, int32_t *sbf, int64_t *counter) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    for (size_t i = 0; i < s->lengths.size(); i++) { lengths[i] = s->lengths[i]; sbf[i] = s->seqs_by_file[i]; }
    *counter = s->counter;
  });
}
int fa_sketch_set_state(fa_sketch *s, int64_t n_genomes, const uint64_t *lengths, const int32_t *sbf, int64_t counter,
                        int64_t n_min, const uint32_t *hash, const int32_t *seq_id, const int32_t *wpos) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    s->reset_data();
    s->lengths.assign(lengths, lengths + n_genomes);
    s->seqs_by_file.assign(sbf, sbf + n_genomes);
    s->counter = counter;
    if (n_min > 0) {
      require_device();
      bind_device(s->device);
      if (!s->stream) FA_HIP(hipStreamCreate(&s->stream));
      s->rec_hash.upload(hash, (size_t)n_min, s->stream);
      s->rec_seq.upload(seq_id, (size_t)n_min, s->stream);
      s->rec_wpos.upload(wpos, (size_t)n_min, s->stream);
      FA_HIP(hipStreamSynchronize(s->stream));
    }
    s->nrec = n_min;
  });
}

// Device-pointer variants of the two calls above: the minimizer records never leave HBM.  Used by the multi-GPU index
// build (every rank sketches a share of the references, the shards are all-gathered over RCCL into
// tensors the caller owns, and every rank loads the merged records).  The caller synchronises its own stream before
// the call; the library synchronises its stream before returning.
int fa_sketch_get_minimizers_device(fa_sketch *s, int64_t cap, uint32_t *d_hash, int32_t *d_seq_id, int32_t *d_wpos) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    s->flush();
    FA_REQUIRE(cap >= s->nrec, FA_ERR_INVALID, "destination holds fewer records than the sketch");
    if (s->nrec == 0) return;
    const size_t n = (size_t)s->nrec;
    FA_HIP(hipMemcpyAsync(d_hash, s->rec_hash.p, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FA_HIP(hipMemcpyAsync(d_seq_id, s->rec_seq.p, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    FA_HIP(hipMemcpyAsync(d_wpos, s->rec_wpos.p, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    FA_HIP(hipStreamSynchronize(s->stream));
  });
}
int fa_sketch_set_state_device(fa_sketch *s, int64_t n_genomes, const uint64_t *lengths, const int32_t *sbf, int64_t counter,
                               int64_t n_min, const uint32_t *d_hash, const int32_t *d_seq_id, const int32_t *d_wpos) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    FA_REQUIRE(n_min >= 0 && n_genomes >= 0, FA_ERR_INVALID, "negative count");
    s->reset_data();
    s->lengths.assign(lengths, lengths + n_genomes);
    s->seqs_by_file.assign(sbf, sbf + n_genomes);
    s->counter = counter;
    if (n_min > 0) {
      require_device();
      bind_device(s->device);
      if (!s->stream) FA_HIP(hipStreamCreate(&s->stream));
      const size_t n = (size_t)n_min;
      s->rec_hash.ensure(n); s->rec_seq.ensure(n); s->rec_wpos.ensure(n);
      FA_HIP(hipMemcpyAsync(s->rec_hash.p, d_hash, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
      FA_HIP(hipMemcpyAsync(s->rec_seq.p, d_seq_id, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
      FA_HIP(hipMemcpyAsync(s->rec_wpos.p, d_wpos, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
      FA_HIP(hipStreamSynchronize(s->stream));
    }
    s->nrec = n_min;
  });
}

int fa_sketch_index(fa_sketch *s, fa_mapper **out) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(s->mtx);
    bind_device(s->device);
    require_device();
    s->flush();
    bind_device(s->device);
    std::unique_ptr<fa_mapper> m(new fa_mapper());
    m->P = s->P;
    m->device = s->device;
    if (m->device < 0) FA_HIP(hipGetDevice(&m->device));
    FA_HIP(hipStreamCreate(&m->stream));
    m->rec_hash = std::move(s->rec_hash);
    m->rec_seq = std::move(s->rec_seq);
    m->rec_wpos = std::move(s->rec_wpos);
    m->N = s->nrec;
    m->rec_hash.ensure((size_t)m->N + 4, true, m->stream, (size_t)m->N);
    m->rec_seq.ensure((size_t)m->N + 4, true, m->stream, (size_t)m->N);
    m->rec_wpos.ensure((size_t)m->N + 4, true, m->stream, (size_t)m->N);
    m->lengths = s->lengths;
    m->seqs_by_file = s->seqs_by_file;
    if (!m->seqs_by_file.empty() && m->seqs_by_file.back() < (int32_t)s->counter) {
      // contigs added after the last end_genome belong to no genome; keep the tables consistent
      m->seqs_by_file.back() = (int32_t)s->counter;
    }
    build_index(*m);
    s->reset_data();                                     // _fastani.pyx:803-804
    *out = m.release();
  });
}

void fa_mapper_free(fa_mapper *m) {
  if (!m) return;
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;   // the workspaces release their own streams, events and pinned blocks
}
int fa_mapper_freq_threshold(fa_mapper *m, int *thr) { *thr = m->freq_threshold; return FA_OK; }
int fa_mapper_lookup_size(fa_mapper *m, int64_t *n) { *n = m->U; return FA_OK; }
int fa_mapper_device(fa_mapper *m, int *device) { *device = m->device; return FA_OK; }
int fa_mapper_lookup_export_device(fa_mapper *m, int64_t cap, uint32_t *d_keys, int32_t *d_counts) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    FA_REQUIRE(cap >= m->U, FA_ERR_INVALID, "destination smaller than the lookup index");
    if (m->U == 0) return;
    FA_HIP(hipMemcpyAsync(d_keys, m->uniq_hash.p, (size_t)m->U * sizeof(uint32_t), hipMemcpyDeviceToDevice, m->stream));
    hipLaunchKernelGGL(k_list_lengths, dim3(ceil_div(m->U, 256)), dim3(256), 0, m->stream, m->uniq_off.p, (int64_t)m->U, d_counts);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(m->stream));
  });
}
int fa_mapper_set_global_frequency(fa_mapper *m, int threshold, int64_t n_drop, const uint32_t *d_drop_keys) {
  return guarded([&] {
    FA_REQUIRE(threshold >= 0 && n_drop >= 0, FA_ERR_INVALID, "negative threshold or key count");
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    m->freq_threshold = threshold;
    if (n_drop > 0 && m->U > 0) {
      hipLaunchKernelGGL(k_drop_keys, dim3(ceil_div(n_drop, 256)), dim3(256), 0, m->stream, d_drop_keys, n_drop, m->table_bits, m->table.p);
      FA_HIP(hipGetLastError());
    }
    FA_HIP(hipStreamSynchronize(m->stream));
  });
}
int fa_mapper_lookup_keys(fa_mapper *m, uint32_t *keys) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    m->uniq_hash.download(keys, (size_t)m->U, m->stream);
    FA_HIP(hipStreamSynchronize(m->stream));
  });
}
static int64_t host_find(fa_mapper *m, uint32_t hash, uint32_t *off, uint32_t *cnt) {
  // binary search with single-element reads (introspection path, not performance critical)
  int64_t lo = 0, hi = m->U;
  while (lo < hi) {
    int64_t mid = (lo + hi) / 2;
    uint32_t v;
    FA_HIP(hipMemcpy(&v, m->uniq_hash.p + mid, 4, hipMemcpyDeviceToHost));
    if (v < hash) lo = mid + 1; else hi = mid;
  }
  if (lo >= m->U) return -1;
  uint32_t v, o[2];
  FA_HIP(hipMemcpy(&v, m->uniq_hash.p + lo, 4, hipMemcpyDeviceToHost));
  if (v != hash) return -1;
  FA_HIP(hipMemcpy(o, m->uniq_off.p + lo, 8, hipMemcpyDeviceToHost));
  *off = o[0]; *cnt = o[1] - o[0];
  return lo;
}
int fa_mapper_lookup_count(fa_mapper *m, uint32_t hash, int64_t *count) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    uint32_t off, cnt;
    *count = host_find(m, hash, &off, &cnt) < 0 ? -1 : (int64_t)cnt;
  });
}
int fa_mapper_lookup_get(fa_mapper *m, uint32_t hash, int32_t *seq_id, int32_t *wpos, int64_t cap) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    uint32_t off, cnt;
    FA_REQUIRE(host_find(m, hash, &off, &cnt) >= 0, FA_ERR_INVALID, "hash not in the lookup index");
    std::vector<uint32_t> ridx(cnt);
    FA_HIP(hipMemcpy(ridx.data(), m->pos_ridx.p + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < cnt && (int64_t)i < cap; i++) {
      FA_HIP(hipMemcpy(&seq_id[i], m->rec_seq.p + ridx[i], 4, hipMemcpyDeviceToHost));
      FA_HIP(hipMemcpy(&wpos[i], m->rec_wpos.p + ridx[i], 4, hipMemcpyDeviceToHost));
    }
  });
}
int fa_mapper_num_minimizers(fa_mapper *m, int64_t *n) { *n = m->N; return FA_OK; }
int fa_mapper_get_minimizers(fa_mapper *m, uint32_t *hash, int32_t *seq_id, int32_t *wpos) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    if (m->N == 0) return;
    m->rec_hash.download(hash, (size_t)m->N, m->stream);
    m->rec_seq.download(seq_id, (size_t)m->N, m->stream);
    m->rec_wpos.download(wpos, (size_t)m->N, m->stream);
    FA_HIP(hipStreamSynchronize(m->stream));
  });
}
int fa_mapper_num_genomes(fa_mapper *m, int64_t *n) { *n = (int64_t)m->lengths.size(); return FA_OK; }
int fa_mapper_get_state(fa_mapper *m, uint64_t *lengths, int32_t *sbf) {
  for (size_t i = 0; i < m->lengths.size(); i++) { lengths[i] = m->lengths[i]; sbf[i] = m->seqs_by_file[i]; }
  return FA_OK;
}

int fa_genomes_upload(fa_mapper *m, const void *const *contigs, const int64_t *lengths, const int32_t *contig_genome,
                      int64_t n_contigs, int32_t n_genomes, int char_width, fa_genomes **out) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    *out = upload_genomes(m->P, m->stream, contigs, lengths, contig_genome, n_contigs, n_genomes, char_width).release();
  });
}
int fa_genomes_upload_fasta(fa_mapper *m, const char *const *paths, int32_t n_paths, fa_genomes **out) {
  return guarded([&] {
    FA_REQUIRE(n_paths >= 0, FA_ERR_INVALID, "negative count");
    std::unique_ptr<fa_genomes> g(new fa_genomes());
    fill_genomes_from_fasta(m, g.get(), paths, n_paths, false);
    *out = g.release();
  });
}
int fa_genomes_reload_fasta(fa_mapper *m, fa_genomes *g, const char *const *paths, int32_t n_paths) {
  return guarded([&] {
    FA_REQUIRE(g && n_paths >= 0, FA_ERR_INVALID, "null batch or negative count");
    try {
      fill_genomes_from_fasta(m, g, paths, n_paths, true);
    } catch (...) {
      g->n_genomes = 0; g->F = 0; g->ntiles = 0;                      // a failed refill leaves an empty (but valid) batch
      g->genome_frag_lo.assign(1, 0); g->total_fragments.clear(); g->total_length.clear(); g->n_short.clear();
      throw;
    }
  });
}
void fa_genomes_free(fa_genomes *g) { delete g; }
int fa_genomes_info(fa_genomes *g, int32_t *n_genomes, uint64_t *tf, uint64_t *tl, int32_t *ns) {
  if (n_genomes) *n_genomes = g->n_genomes;
  for (int i = 0; i < g->n_genomes; i++) {
    if (tf) tf[i] = g->total_fragments[i];
    if (tl) tl[i] = g->total_length[i];
    if (ns) ns[i] = g->n_short[i];
  }
  return FA_OK;
}
int fa_mapper_query_genomes(fa_mapper *m, fa_genomes *g, int32_t first, int32_t count, fa_cgi_row *rows, int64_t cap,
                            int64_t *n_rows, int rows_device) {
  return guarded([&] {
    WorkspaceLease lease(*m);
    *n_rows = run_query(*m, *lease.w, *g, first, count, rows, cap, rows_device != 0);
  });
}
int fa_mapper_query(fa_mapper *m, const void *const *contigs, const int64_t *lengths, int n_contigs, int char_width,
                    fa_cgi_row *rows, int64_t cap, int64_t *n_rows, int *n_short, uint64_t *total_fragments,
                    uint64_t *total_length) {
  return guarded([&] {
    WorkspaceLease lease(*m);
    std::vector<int32_t> cg((size_t)std::max(n_contigs, 1), 0);
    float host_ms[3] = {0, 0, 0};
    auto g = upload_genomes(m->P, lease.w->stream, contigs, lengths, cg.data(), n_contigs, 1, char_width, host_ms,
                            std::move(lease.w->query_batch), &lease.w->pin_image);
    if (n_short) *n_short = g->n_short[0];
    if (total_fragments) *total_fragments = g->total_fragments[0];
    if (total_length) *total_length = g->total_length[0];
    const auto t0 = std::chrono::steady_clock::now();
    *n_rows = run_query(*m, *lease.w, *g, 0, 1, rows, cap, false);
    // host-side split of the boundary call (wall clock): packing, fragment/tile tables, H2D, pass + rows D2H
    for (int i = 0; i < 3; i++) lease.w->last_ms[10 + i] = host_ms[i];
    lease.w->last_ms[13] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    lease.w->last_genomes = nullptr;
    lease.w->query_batch = std::move(g);                 // keep the device buffers for the next call
  });
}

// The parts of the last pass, in fragment order: a pass is pipelined over up to three lanes (run_query_pass), every
// lane still holds the intermediates of the last part it ran.
static std::vector<Workspace *> lanes_of_last_pass(Workspace &w) {
  std::vector<Workspace *> v;
  for (Workspace *x : {&w, w.sub[0].get(), w.sub[1].get()}) if (x && x->serial == w.serial && x->last_F > 0) v.push_back(x);
  std::sort(v.begin(), v.end(), [](const Workspace *a, const Workspace *b) { return a->last_f0 < b->last_f0; });
  // a pass that ran in more parts than there are lanes (several passes of one genome, or parts that were repeated) has
  // left only its last parts behind: say so instead of handing out a fraction of the pass as if it were all of it
  int64_t covered = 0;
  for (const Workspace *x : v) covered += x->last_F;
  FA_REQUIRE(covered == w.pass_F, FA_ERR_UNSUPPORTED,
             "the stage getters hold the last part of every lane only, and the last pass ran in more parts than that");
  return v;
}
int fa_mapper_debug_mappings(fa_mapper *m, fa_mapping *out, int64_t cap, int64_t *n) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    Workspace &w = m->ws[m->last_ws];
    ensure_luts(*m, 1);
    int64_t k = 0;
    for (Workspace *x : lanes_of_last_pass(w)) {
      const uint32_t L = x->last_loci;
      std::vector<int32_t> lf(L), ls(L), lp(L), lsh(L), qs((size_t)x->last_F);
      size_t at = 0;
      for_each_locus_slice(*x, [&](size_t first, size_t count) {
        FA_HIP(hipMemcpyAsync(lf.data() + at, x->l_frag.p + first, count * 4, hipMemcpyDeviceToHost, x->stream));
        FA_HIP(hipMemcpyAsync(ls.data() + at, x->l_seq.p + first, count * 4, hipMemcpyDeviceToHost, x->stream));
        FA_HIP(hipMemcpyAsync(lp.data() + at, x->l_pos.p + first, count * 4, hipMemcpyDeviceToHost, x->stream));
        FA_HIP(hipMemcpyAsync(lsh.data() + at, x->l_shared.p + first, count * 4, hipMemcpyDeviceToHost, x->stream));
        at += count;
      });
      x->q_size.download(qs.data(), (size_t)x->last_F, x->stream);
      FA_HIP(hipStreamSynchronize(x->stream));
      for (uint32_t i = 0; i < L; i++) {
        int s = qs[lf[i]];
        if (lsh[i] < m->stats.pass_shared[s]) continue;
        if (k < cap) {
          fa_mapping r;
          r.query_seq_id = lf[i] + (int32_t)(x->last_f0 - w.pass_f0); r.ref_seq_id = ls[i]; r.ref_start_pos = lp[i]; r.sketch_size = s;
          r.conserved = lsh[i]; r.query_id = 0;
          out[k] = r;
        }
        k++;
      }
    }
    *n = k;
  });
}
int fa_mapper_debug_l1(fa_mapper *m, int32_t *frag, int32_t *seq_id, int32_t *rs, int32_t *re, int64_t cap, int64_t *n) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    Workspace &w = m->ws[m->last_ws];
    int64_t k = 0;
    for (Workspace *x : lanes_of_last_pass(w)) {
      int64_t kk = k;
      for_each_locus_slice(*x, [&](size_t first, size_t count) {
        const size_t c = (size_t)std::max<int64_t>(0, std::min<int64_t>((int64_t)count, cap - kk));
        if (c) {
          FA_HIP(hipMemcpyAsync(frag + kk, x->l_frag.p + first, c * 4, hipMemcpyDeviceToHost, x->stream));
          FA_HIP(hipMemcpyAsync(seq_id + kk, x->l_seq.p + first, c * 4, hipMemcpyDeviceToHost, x->stream));
          FA_HIP(hipMemcpyAsync(rs + kk, x->l_start.p + first, c * 4, hipMemcpyDeviceToHost, x->stream));
          FA_HIP(hipMemcpyAsync(re + kk, x->l_end.p + first, c * 4, hipMemcpyDeviceToHost, x->stream));
        }
        kk += (int64_t)count;
      });
      FA_HIP(hipStreamSynchronize(x->stream));
      for (int64_t i = k; i < std::min<int64_t>(kk, cap); i++) frag[i] += (int32_t)(x->last_f0 - w.pass_f0);
      k = kk;
    }
    *n = k;
  });
}
int fa_mapper_debug_query_sketch(fa_mapper *m, int64_t fragment, uint32_t *hashes, int32_t cap, int32_t *sketch_size) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    Workspace &w = m->ws[m->last_ws];
    for (Workspace *x : lanes_of_last_pass(w)) {
      const int64_t local = fragment - (x->last_f0 - w.pass_f0);
      if (local < 0 || local >= x->last_F) continue;
      int32_t s = 0;
      FA_HIP(hipMemcpy(&s, x->q_size.p + local, 4, hipMemcpyDeviceToHost));
      *sketch_size = s;
      int c = std::min(s, cap);
      if (c > 0) FA_HIP(hipMemcpy(hashes, x->q_hash.p + (size_t)local * m->qcap, (size_t)c * 4, hipMemcpyDeviceToHost));
      return;
    }
    throw Error(FA_ERR_INVALID, "fragment out of range");
  });
}
int fa_debug_sketch_sequence(const fa_params *params, const void *data, int64_t length, int char_width, uint32_t *hash,
                             int32_t *wpos, int64_t cap, int64_t *n) {
  return guarded([&] {
    validate_params(*params);
    fa_sketch s;
    s.P = *params;
    s.reset_data();
    if (length >= params->kmer_size) {
      s.pending.append(data, char_width, length);
      s.pending_contig.push_back(0);
    }
    s.flush();
    *n = s.nrec;
    size_t c = (size_t)std::min<int64_t>(s.nrec, cap);
    if (c) {
      s.rec_hash.download(hash, c, s.stream);
      s.rec_wpos.download(wpos, c, s.stream);
      FA_HIP(hipStreamSynchronize(s.stream));
    }
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.up_stream) (void)hipStreamDestroy(s.up_stream);
  });
}

// development probe: how many 128-thread workgroups with `lds_bytes` of dynamic LDS does the chip hold at once, right now?
__global__ void k_probe_occupancy(unsigned *alive, unsigned *peak, int spin) {
  extern __shared__ unsigned char probe_lds[];
  if (threadIdx.x == 0) { unsigned a = atomicAdd(alive, 1u) + 1u; atomicMax(peak, a); }
  probe_lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(alive, 1u);
  if (probe_lds[(threadIdx.x + 1) & 63] == 255 && spin < 0) alive[1] = 1;
}
int fa_debug_probe_occupancy(int lds_bytes, int *peak_alive) {
  return guarded([&] {
    require_device();
    DevBuf<unsigned> d;
    d.ensure(16);
    FA_HIP(hipMemset(d.p, 0, 64));
    hipLaunchKernelGGL(k_probe_occupancy, dim3(4096), dim3(128), (size_t)lds_bytes, 0, d.p, d.p + 4, 200000);
    FA_HIP(hipDeviceSynchronize());
    unsigned h[8];
    FA_HIP(hipMemcpy(h, d.p, 32, hipMemcpyDeviceToHost));
    *peak_alive = (int)h[4];
  });
}
int fa_mapper_debug_items(fa_mapper *m, void *out, int64_t bytes) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    Workspace &w = m->ws[m->last_ws];
    FA_REQUIRE(bytes >= 0 && (size_t)bytes <= w.items.cap, FA_ERR_INVALID, "more bytes than the event arena holds");
    FA_HIP(hipMemcpy(out, w.items.p, (size_t)bytes, hipMemcpyDeviceToHost));
  });
}
int fa_mapper_debug_links(fa_mapper *m, int32_t *prev, int32_t *fwd, int32_t *bwd, uint8_t *flags, int64_t cap, int64_t *n) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    const size_t c = (size_t)std::max<int64_t>(0, std::min<int64_t>(m->N, cap));
    if (c) {
      FA_HIP(hipMemcpy(prev, m->rec_prev.p, c * 4, hipMemcpyDeviceToHost));
      FA_HIP(hipMemcpy(fwd, m->rec_fwd.p, c * 4, hipMemcpyDeviceToHost));
      FA_HIP(hipMemcpy(bwd, m->rec_bwd.p, c * 4, hipMemcpyDeviceToHost));
      FA_HIP(hipMemcpy(flags, m->rec_flags.p, c, hipMemcpyDeviceToHost));
    }
    *n = m->N;
  });
}
int fa_mapper_debug_locus_events(fa_mapper *m, uint32_t *events, int64_t cap, int64_t *n) {
  return guarded([&] {
    std::lock_guard<std::mutex> lock(m->mtx);
    bind_device(m->device);
    Workspace &w = m->ws[m->last_ws];
    int64_t k = 0;
    for (Workspace *x : lanes_of_last_pass(w)) {
      for_each_locus_slice(*x, [&](size_t first, size_t count) {
        const size_t c = (size_t)std::max<int64_t>(0, std::min<int64_t>((int64_t)count, cap - k));
        if (c) FA_HIP(hipMemcpyAsync(events + k, x->l_nev.p + first, c * 4, hipMemcpyDeviceToHost, x->stream));
        k += (int64_t)count;
      });
      FA_HIP(hipStreamSynchronize(x->stream));
    }
    *n = k;
  });
}
int fa_mapper_set_stage_events(fa_mapper *m, int on) {
  std::lock_guard<std::mutex> lock(m->mtx);
  m->stage_events = on != 0;
  return FA_OK;
}
int fa_mapper_last_timings(fa_mapper *m, float *ms, int n) {
  std::lock_guard<std::mutex> lock(m->mtx);
  for (int i = 0; i < n && i < 24; i++) ms[i] = m->ws[m->last_ws].last_ms[i];
  return FA_OK;
}
int fa_mapper_stream(fa_mapper *m, void **stream) { *stream = (void *)m->stream; return FA_OK; }

int fa_bench_sketch_kernel(fa_mapper *m, fa_genomes *g, int repeat, float *ms_per_launch, uint64_t *bases, uint64_t *minimizers) {
  return guarded([&] {
    WorkspaceLease lease(*m);
    Workspace &w = *lease.w;
    require_device();
    int ntiles = (int)g->ntiles;
    FA_REQUIRE(ntiles > 0 && repeat > 0, FA_ERR_INVALID, "nothing to sketch");
    // the genome as REFERENCE sketching sees it: the fragments of a contig lie back to back in the batch's store, so they are
    // joined into whole sequences again and cut into the tiles reference sketching uses (k1_tile_len: positions + halo = whole
    // hashing trips); the batch's own tiles are per query fragment, for k_query_fused.  Batches with exceptions keep their tiles.
    const Tile *tiles = g->tiles;
    DevBuf<Tile> retiled;
    const int tile_len = k1_tile_len(m->P.window_size);
    if (g->store.n_exc == 0 && m->P.alphabet_size == 4) {
      std::vector<Tile> host((size_t)ntiles), cut;
      FA_HIP(hipMemcpy(host.data(), g->tiles, (size_t)ntiles * sizeof(Tile), hipMemcpyDeviceToHost));
      int64_t run_base = -1, run_len = 0;
      int32_t run_id = 0, last_seq = -1;
      auto close = [&] {
        if (run_base < 0) return;
        const int64_t npos = run_len - m->P.kmer_size + 1;
        for (int64_t p0 = 0; p0 < npos; p0 += tile_len)
          cut.push_back(Tile{run_base, (int32_t)run_len, (int32_t)p0, (int32_t)std::min<int64_t>(tile_len, npos - p0), run_id, 0, 0});
        run_id++; run_base = -1; run_len = 0;
      };
      // (a run ends where a contig ends, not only where the addresses break: a contig whose whole fragments fill a multiple of
      // 64 bases is followed by the next one without a gap, and k-mers / windows across that seam are not what reference
      // sketching hashes)
      size_t next_contig = 0;
      for (const Tile &t : host) {
        if (t.seq == last_seq) continue;                               // (the other tiles of a fragment seen already)
        last_seq = t.seq;
        bool starts_contig = false;
        while (next_contig < g->contig_frag_lo.size() && g->contig_frag_lo[next_contig] <= (int64_t)t.seq) { starts_contig = g->contig_frag_lo[next_contig] == (int64_t)t.seq; next_contig++; }
        if (!starts_contig && run_base >= 0 && t.base == run_base + run_len && run_len + t.seq_len < (1LL << 31)) run_len += t.seq_len;
        else { close(); run_base = t.base; run_len = t.seq_len; }
      }
      close();
      retiled.upload(cut, w.stream);
      tiles = retiled.p; ntiles = (int)cut.size();
    }
    w.last_ms[23] = (float)tile_len;
    w.sk.stage_hash.ensure((size_t)ntiles * TILE);
    w.sk.stage_wpos.ensure((size_t)ntiles * TILE);
    w.sk.tile_count.ensure((size_t)ntiles + 1);
    hipEvent_t e0, e1;
    FA_HIP(hipEventCreate(&e0)); FA_HIP(hipEventCreate(&e1));
    launch_sketch_tiles(m->P, g->store, tiles, ntiles, w.sk.stage_hash.p, w.sk.stage_wpos.p, w.sk.tile_count.p, w.stream);
    FA_HIP(hipEventRecord(e0, w.stream));
    for (int i = 0; i < repeat; i++)
      launch_sketch_tiles(m->P, g->store, tiles, ntiles, w.sk.stage_hash.p, w.sk.stage_wpos.p, w.sk.tile_count.p, w.stream);
    FA_HIP(hipEventRecord(e1, w.stream));
    FA_HIP(hipEventSynchronize(e1));
    float ms = 0;
    FA_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / repeat;
    std::vector<int32_t> counts((size_t)ntiles);
    w.sk.tile_count.download(counts.data(), (size_t)ntiles, w.stream);
    FA_HIP(hipStreamSynchronize(w.stream));
    uint64_t tot = 0;
    for (int c : counts) tot += (uint64_t)c;
    *minimizers = tot;
    *bases = g->total_bases;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  });
}

}  // extern "C"
