// fa_map.hip.h -- index construction and the fragment mapper (lookup, L1, L2, core-genome identity) on gfx950.
//
// What it replaces (reference = pyfastani + the FastANI C++ it links):
//   skch::Sketch::index / computeFreqHist / searchIndex   include/fastani/map/win_sketch.pxd:38-41
//   Mapper._do_l1_mappings                                src/pyfastani/_fastani.pyx:885-954
//   skch::Map::computeL1CandidateRegions                  include/fastani/map/compute_map.pxd:33
//   skch::Map::doL2Mapping / computeL2MappedRegions       compute_map.pxd:34-35 (+ slidingMap.hpp, MIIteratorL2.hpp)
//   cgi::computeCGI                                       include/fastani/cgi/compute_core_identity.pxd:28-37
//
// HBM layout of an indexed reference (all SoA, resident for the life of the mapper)
//   rec_hash/rec_seq/rec_wpos[N]   the minimizer records in (contig, window) order  = skch::Sketch::minimizerIndex
//   rec_prev[N]                    index of the previous record of the same contig with the same hash, or -1
//   rec_flags[N]                   bit0: the previous same-hash record is still inside the L2 super-window when this one
//                                  is admitted; bit1: the next same-hash record is admitted before this one is dropped
//   uniq_hash[U], uniq_off[U+1]    sorted distinct hashes and CSR offsets into pos_ridx = minimizerPosLookupIndex
//   pos_ridx[N]                    record indices grouped by hash, ascending inside a group
//   table[2^B]                     open-addressing hash table {hash, CSR offset, length+1}: one 16-byte probe per lookup
//   contig_rec[C+1], contig_genome[C], contig_bin[C+1], genome_bin[G+1]
//
// Query side, per batch of fragments: q_hash[F*QCAP] sorted distinct query minimizers, q_size[F] = sketchSize.
#pragma once

#include "fa_common.h"
#include "fa_sketch.hip.h"
#include "fa_sketch_fast.hip.h"

namespace fa {



constexpr int MAP_THREADS = 256;
// loci of one fragment merged in LDS by k_l1 (more fall back to a second pass that writes them to HBM).  128, not 256: with
// 3 KB of stage instead of 6 the kernel's 21.7 KB let seven workgroups share a CU, i.e. the 1666 fragments of a 5 Mb query run
// in ONE resident round (85 -> 81 us; 928 -> 872 us at 16 queries per launch); a fragment has one locus per related contig
constexpr int L1_STAGE = 128;
// the size classes of k_l1 (hits of a fragment): up to 16 per thread of the 256-thread form; up to the slots the 512-thread,
// 16-per-thread form is given (a kilobyte short of 16 x 512: see Part::L1Class); everything beyond
constexpr uint32_t L1_SMALL_HITS = 16u * 256u, L1_MID_HITS = 16u * 512u - 256u;
constexpr uint32_t SEED_PAD = 0xFFFFFFFFu;

// ----------------------------------------------------------------------------------------------------------
// in-place bitonic sort of n32 (power of two) uint32 keys by one workgroup; works on LDS or global memory
// ----------------------------------------------------------------------------------------------------------
__device__ inline void block_bitonic_sort(uint32_t *a, uint32_t n32) {
  for (uint32_t size = 2; size <= n32; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < n32 / 2; t += blockDim.x) {
        uint32_t lo = 2 * t - (t & (stride - 1));
        uint32_t hi = lo + stride;
        bool up = (lo & size) == 0;
        uint32_t x = a[lo], y = a[hi];
        if ((x > y) == up) { a[lo] = y; a[hi] = x; }
      }
    }
  }
  __syncthreads();
}

// ----------------------------------------------------------------------------------------------------------
// index construction helpers (the radix sort / run-length / scan primitives come from hipCUB)
// ----------------------------------------------------------------------------------------------------------
__global__ void k_iota(uint32_t *a, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (uint32_t)i;
}

// Open-addressing lookup table over the distinct hashes (= unordered_map::find of _fastani.pyx:943 in one 16-byte
// probe): entry = {hash, offset into pos_ridx, list length + 1, unused}; .z == 0 marks an empty slot.  Slots are
// chosen by a multiplicative mix because minimizer hashes are biased towards small values.
__device__ __forceinline__ uint32_t ht_slot(uint32_t h, int bits) { return (h * 0x9E3779B1u) >> (32 - bits); }

__global__ void k_build_table(const uint32_t *uniq_hash, const uint32_t *uniq_off, int64_t U, int bits, uint4 *table) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const uint32_t h = uniq_hash[u], off = uniq_off[u], cnt = uniq_off[u + 1] - off;
  const uint32_t mask = (1u << bits) - 1u;
  for (uint32_t slot = ht_slot(h, bits);; slot = (slot + 1) & mask) {
    if (atomicCAS(&table[slot].z, 0u, cnt + 1u) == 0u) { table[slot].x = h; table[slot].y = off; return; }
  }
}

// Reference-sharded index ("when the index does not fit"): hashes whose position lists, summed over the
// shards of all ranks, reach the global frequency threshold are ignored by the lookup although the local list is short.
__global__ void k_drop_keys(const uint32_t *keys, int64_t n, int bits, uint4 *table) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = keys[i], mask = (1u << bits) - 1u;
  for (uint32_t slot = ht_slot(h, bits);; slot = (slot + 1) & mask) {
    const uint4 e = table[slot];
    if (e.z == 0u) return;                       // not in this shard
    if (e.x == h) { table[slot].w = 1u; return; }
  }
}
__global__ void k_list_lengths(const uint32_t *uniq_off, int64_t U, int32_t *counts) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < U) counts[u] = (int32_t)(uniq_off[u + 1] - uniq_off[u]);
}

// computeFreqHist needs the largest list lengths only (the top 0.001 % of the distinct hashes): a histogram of the lengths
// below FREQ_BINS (LDS per workgroup, then one atomic per occupied bin) and the few longer ones verbatim, instead of a
// descending sort of all lengths.  `over_n` counts every long list; those beyond `over_cap` are not stored (the caller sorts then).
constexpr int FREQ_BINS = 4096;
__global__ __launch_bounds__(256) void k_length_histogram(const uint32_t *counts, int64_t U, unsigned long long *hist, uint32_t *over,
                                                           uint32_t over_cap, uint32_t *over_n) {
  __shared__ uint32_t sh[FREQ_BINS];
  for (int i = threadIdx.x; i < FREQ_BINS; i += 256) sh[i] = 0;
  __syncthreads();
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < U; u += (int64_t)gridDim.x * 256) {
    const uint32_t c = counts[u];
    if (c < (uint32_t)FREQ_BINS) atomicAdd(&sh[c], 1u);
    else {
      const uint32_t at = atomicAdd(over_n, 1u);
      if (at < over_cap) over[at] = c;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FREQ_BINS; i += 256) if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

// contig_rec[c] = first record with rec_seq >= c
__global__ void k_contig_ranges(const int32_t *rec_seq, int64_t N, int C, int32_t *contig_rec) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  int64_t lo = 0, hi = N;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (rec_seq[mid] < c) lo = mid + 1; else hi = mid; }
  contig_rec[c] = (int32_t)lo;
}

// rec_prev / rec_flags from the hash-grouped order.  A record i of contig q is inside the L2 super-window at window
// position p for p in [a_i, b_i], a_i = wpos_i - cmw + 1, b_i = wpos_{i+1} - 1 (next record of the contig; +inf for
// the last one).  Two consecutive same-hash records (j, i) are `linked` when b_j >= a_i - 1: the hash never leaves
// the window between them, so admitting i / dropping j must not change the set (slidingMap.hpp REV / NOOP cases).
//
// The pass streams the sorted (hash, record) pairs and touches the records themselves only for a pair inside ONE contig, which
// is the rare case (a repeat inside a contig; relatives in other genomes are the common one).  "Same contig" is p >= first
// record of cur's contig (p < cur: the sort is stable), and that first record comes from a table with one entry per 2^shift
// consecutive records (k_block_contig; 1.6 MB for 4 x 10^8 records -- it lives in L2) instead of two gathers behind the TLB;
// blocks that straddle a contig boundary say so and take the gather.  rec_prev is pre-filled with -1 by the caller, rec_flags
// hold FLAG_SAME_STEP or 0 (k_window_links runs first).
constexpr uint8_t FLAG_INS_LINKED = 1, FLAG_DEL_LINKED = 2;
constexpr uint32_t BLOCK_STRADDLES = 0xFFFFFFFFu;
__global__ void k_block_contig(const int32_t *rec_seq, const int32_t *contig_rec, int64_t N, int shift, uint32_t *blk_lo) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t first = b << shift;
  if (first >= N) return;
  const int64_t last = min(N - 1, first + ((int64_t)1 << shift) - 1);
  const int c = rec_seq[first];
  blk_lo[b] = rec_seq[last] == c ? (uint32_t)contig_rec[c] : BLOCK_STRADDLES;
}
__global__ void k_link_duplicates(const uint32_t *sorted_hash, const uint32_t *pos_ridx, int64_t N, const int32_t *rec_seq,
                                  const int32_t *rec_wpos, const int32_t *contig_rec, const uint32_t *blk_lo, int shift, int cmw,
                                  int32_t *rec_prev, uint8_t *rec_flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || i == 0) return;
  if (sorted_hash[i - 1] != sorted_hash[i]) return;
  const uint32_t cur = pos_ridx[i], p = pos_ridx[i - 1];
  uint32_t lo = blk_lo[cur >> shift];
  if (lo == BLOCK_STRADDLES) lo = (uint32_t)contig_rec[rec_seq[cur]];
  if (p < lo) return;                            // the earlier record with this hash lies in another contig
  rec_prev[cur] = (int32_t)p;
  // p + 1 <= cur exists and lies in the same contig (records of a contig are contiguous)
  const int64_t b_prev = (int64_t)rec_wpos[p + 1] - 1;
  const int64_t a_cur = (int64_t)rec_wpos[cur] - cmw + 1;
  if (b_prev >= a_cur - 1) {
    atomicOr((unsigned int *)(rec_flags + (cur & ~3u)), (unsigned int)FLAG_INS_LINKED << (8 * (cur & 3u)));
    atomicOr((unsigned int *)(rec_flags + (p & ~3u)), (unsigned int)FLAG_DEL_LINKED << (8 * (p & 3u)));
  }
}

// Query-independent geometry of the L2 slide, computed once per index (cmw = windows per fragment):
//   rec_fwd[i] = first record of the contig with wpos >= wpos[i] + cmw   (end of the super-window that starts at i)
//   rec_bwd[i] = last record of the contig with wpos <= wpos[i] - cmw + 1 (the record active when i is admitted)
//   FLAG_SAME_STEP on record r: the window position that drops r (wpos[r+1]) also admits a record
// With these, the position of every admit / drop event in the time-ordered event stream of a locus is plain
// arithmetic (no merge search per query).
//
// (Midpoints as x + (y - x) / 2 throughout: record numbers reach 2^31 and x + y does not fit -- an index beyond 2^30 records
// never came back from this kernel, nor from the prologue of k_l2_events, before round 5 tried one.)
// Three binary searches per record over the window positions of at most cmw records either side (wpos is strictly increasing
// inside a contig).  A workgroup takes WL_TILE consecutive records and holds their window positions plus `halo` records either
// side in LDS, so the ~36 dependent reads of a record are LDS reads; whatever a search needs outside (cmw > WL_HALO_MAX: the
// caller passes the largest halo that fits) is read from memory -- `at()` is the only difference from a search over the array.
// The flags are written as whole bytes (FLAG_SAME_STEP or 0): this kernel runs before k_link_duplicates ORs its bits in.
constexpr uint8_t FLAG_SAME_STEP = 4;
constexpr int WL_TILE = 4096, WL_THREADS = 512, WL_HALO_MAX = 5120;   // 56 KB of LDS at most
// (ALL_LDS: the halo covers cmw + 2 records either side, every read of a search is an LDS read -- no range test per step: the
//  kernel is bound by the vector instructions of its ~36 search steps per record, 15 each with the test and 6 without)
template <bool ALL_LDS>
__global__ __launch_bounds__(WL_THREADS) void k_window_links(const int32_t *rec_seq, const int32_t *rec_wpos, const int32_t *contig_rec, int64_t N,
                                                           int cmw, int halo, int32_t *rec_fwd, int32_t *rec_bwd, uint8_t *rec_flags) {
  extern __shared__ int32_t wl_w[];
  const int64_t t0 = (int64_t)blockIdx.x * WL_TILE;
  const int64_t first = t0 - halo;                                      // record of wl_w[0]
  const int held = WL_TILE + 2 * halo;
  for (int j = threadIdx.x; j < held; j += WL_THREADS) {
    const int64_t r = first + j;
    wl_w[j] = r >= 0 && r < N ? rec_wpos[r] : 0;
  }
  __syncthreads();
  const int64_t s_lo = max((int64_t)0, first), s_hi = min(N, first + held);
  const int first32 = (int)first;                                       // (ALL_LDS: N < 2^31 and halo < 2^13, no wrap)
  auto at = [&](int r) -> int { return ALL_LDS ? wl_w[r - first32] : (r >= s_lo && r < s_hi ? wl_w[r - first] : rec_wpos[r]); };
  for (int q = 0; q < WL_TILE / WL_THREADS; q++) {
    const int64_t i = t0 + q * WL_THREADS + threadIdx.x;
    if (i >= N) return;
    const int c = rec_seq[i];
    const int lo = contig_rec[c], hi = contig_rec[c + 1];
    const int w = wl_w[i - first];
    {
      int x = (int)i + 1, y = min(hi, (int)i + 1 + cmw), key = w + cmw;   // wpos is strictly increasing: at most cmw records ahead
      while (x < y) { int mid = x + ((y - x) >> 1); if (at(mid) < key) x = mid + 1; else y = mid; }
      rec_fwd[i] = x;
    }
    {
      int x = max(lo, (int)i - cmw), y = (int)i + 1, key = w - cmw + 1;    // first index with wpos > key, minus one
      while (x < y) { int mid = x + ((y - x) >> 1); if (at(mid) <= key) x = mid + 1; else y = mid; }
      rec_bwd[i] = x - 1;
    }
    uint8_t flag = 0;
    if (i + 1 < hi) {
      const int key = at((int)i + 1) + cmw - 1;
      int x = (int)i + 1, y = min(hi, (int)i + 2 + cmw);                   // wpos[i + 1 + cmw] >= wpos[i + 1] + cmw > key
      while (x < y) { int mid = x + ((y - x) >> 1); if (at(mid) < key) x = mid + 1; else y = mid; }
      if (x < hi && at(x) == key) flag = FLAG_SAME_STEP;
    }
    rec_flags[i] = flag;
  }
}

// Padded global coordinate of every record, for the candidate pass of k_l1: contig c starts at
//   base[c] = sum over c' < c of (last window position of c' + 1 + fragment_length),
// and G(r) = base[rec_seq[r]] + rec_wpos[r].  Records are ordered by (contig, window), so G grows with the record number;
// inside a contig G differences ARE window differences, and two records of different contigs are at least a fragment
// length apart -- "same contig and wb - wa < fragment_length" (computeL1CandidateRegions) is ONE unsigned compare of a G
// difference, with no contig number per seed hit.  rec_gpos keeps the low 32 bits (4 bytes gathered per hit where
// (seqId, wpos) took 8); the high word of G(r) is the number of entries of wrap_rec (first record of every 2^32 boundary
// crossed, a handful for the largest index HBM holds) that are <= r.
constexpr int GPOS_MAX_WRAPS = 256;
__global__ void k_contig_span(const int32_t *contig_rec, const int32_t *rec_wpos, int C, int pad, unsigned long long *span) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  unsigned long long v = 0;
  if (c < C) {
    const int lo = contig_rec[c], hi = contig_rec[c + 1];
    v = (unsigned long long)(hi > lo ? rec_wpos[hi - 1] + 1 : 0) + (unsigned long long)pad;
  }
  span[c] = v;
}
// (`bits` = width of the low word: 32, except in the tests, which shrink it so that a small index spans many boundaries)
__global__ void k_rec_gpos(const int32_t *rec_seq, const int32_t *rec_wpos, const unsigned long long *contig_base, int64_t N, int bits,
                           uint32_t *rec_gpos, uint32_t *wrap_rec, int32_t *n_wraps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const unsigned long long g = contig_base[rec_seq[i]] + (unsigned long long)rec_wpos[i];
  rec_gpos[i] = (uint32_t)(g & ((1ULL << bits) - 1ULL));
  const uint32_t hi = (uint32_t)(g >> bits);
  const uint32_t hi_prev = i ? (uint32_t)((contig_base[rec_seq[i - 1]] + (unsigned long long)rec_wpos[i - 1]) >> bits) : 0u;
  for (uint32_t h = hi_prev + 1; h <= hi; h++) if (h - 1 < (uint32_t)GPOS_MAX_WRAPS) wrap_rec[h - 1] = (uint32_t)i;
  if (i == N - 1) *n_wraps = (int32_t)hi;
}

// The same geometry packed for the hot loop of k_l2_events (fewer bytes per record, ONE load and one address per record):
//   hg[i] = { hash,  (rec_fwd[i+1] - (i+1)) | (i - rec_bwd[i]) << 13 | flags << 26 },   both distances are <= cmw < 8192,
// and prev16[i] = min(i - rec_prev[i], 65535) (65535 also for "no earlier record with this hash in the contig": only
// compared against distances < 8192).  Built only when cmw + 1 < 8192; otherwise the kernels read the plain arrays.
constexpr int GEO_BITS = 13;
__global__ void k_pack_geometry(const uint32_t *rec_hash, const int32_t *rec_prev, const int32_t *rec_fwd, const int32_t *rec_bwd, const uint8_t *rec_flags,
                                int64_t N, uint2 *rec_hg, uint16_t *rec_prev16) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t fwd1 = i + 1 < N ? (uint32_t)(rec_fwd[i + 1] - (int32_t)(i + 1)) : 0u;
  const uint32_t bwd = (uint32_t)((int32_t)i - rec_bwd[i]);
  rec_hg[i] = make_uint2(rec_hash[i], (fwd1 & ((1u << GEO_BITS) - 1u)) | ((bwd & ((1u << GEO_BITS) - 1u)) << GEO_BITS) | ((uint32_t)rec_flags[i] << (2 * GEO_BITS)));
  const int32_t pv = rec_prev[i];
  rec_prev16[i] = (uint16_t)(pv < 0 ? 65535 : min((int32_t)i - pv, 65535));
}

// ----------------------------------------------------------------------------------------------------------
// resident index view passed to the mapping kernels
// ----------------------------------------------------------------------------------------------------------
struct IndexView {
  const uint32_t *rec_hash;
  const int32_t *rec_seq;
  const int32_t *rec_wpos;
  const int32_t *rec_prev;
  const int32_t *rec_fwd, *rec_bwd;
  const uint8_t *rec_flags;
  const uint2 *rec_hg;          // (hash, packed geometry) (k_pack_geometry), null when cmw is too large for it
  const uint16_t *rec_prev16;
  const uint32_t *rec_gpos;     // low word of the padded global coordinate (k_rec_gpos): one 4-byte gather per seed hit in k_l1
  const uint32_t *wrap_rec;     // [n_wraps] first record behind every 2^32 boundary of that coordinate
  int32_t n_wraps;
  int32_t gpos_bits;            // width of the low word (32; FA_GPOS_BITS shrinks it for the tests)
#ifdef FA_EXPERIMENTS
  const uint32_t *ev_bits;      // merged admit / drop order of the L2 slide, one bit per event (k_event_bits)
  const uint2 *rec_hf;          // (hash, flags | distance to the previous record of the hash << 8) for k_l2_fused
#endif
  const uint32_t *uniq_hash;
  const uint32_t *uniq_off;
  const uint32_t *pos_ridx;
  const uint4 *table;
  const int32_t *contig_rec;
  const int32_t *contig_genome;
  const int32_t *contig_bin;
  const int32_t *genome_bin;
  int64_t N, U;
  int32_t C, G;
  int32_t table_bits;
  int32_t freq_threshold;
  int32_t total_bins;
};

// padded global coordinate of record r (lo = rec_gpos[r], fetched by the caller)
__device__ __forceinline__ uint64_t gpos_make(const IndexView &ix, uint32_t r, uint32_t lo) {
  uint32_t hi = 0;
  for (int w = 0; w < ix.n_wraps; w++) hi += r >= ix.wrap_rec[w] ? 1u : 0u;   // (uniform loop, scalar loads; no trip below 4.29 Gbases)
  return ((uint64_t)hi << ix.gpos_bits) | lo;
}
__device__ __forceinline__ uint64_t gpos_of(const IndexView &ix, uint32_t r) { return gpos_make(ix, r, ix.rec_gpos[r]); }

// position list of hash h: true if present; off / cnt receive the CSR slice
__device__ __forceinline__ bool index_find(const IndexView &ix, uint32_t h, uint32_t &off, uint32_t &cnt) {
  const uint32_t mask = (1u << ix.table_bits) - 1u;
  for (uint32_t slot = ht_slot(h, ix.table_bits);; slot = (slot + 1) & mask) {
    const uint4 e = ix.table[slot];
    if (e.z == 0u) return false;
    if (e.x == h) { off = e.y; cnt = e.w ? 0u : e.z - 1u; return true; }   // w: too frequent over all index shards (k_drop_keys)
  }
}

// ----------------------------------------------------------------------------------------------------------
// lookup: for every query minimizer its position list in the index, frequency-filtered (_fastani.pyx:941-948)
// ----------------------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------------------
// query fragments: gather the staged records of a fragment's tiles, apply the leading-run rule, sort by hash,
// drop duplicates (std::sort + std::unique of _fastani.pyx:929-936).  One workgroup per fragment.
// ----------------------------------------------------------------------------------------------------------
struct QuerySketchArgs {
  const int32_t *frag_tile_lo;  // [F+1]
  const int32_t *tile_count;
  const uint32_t *stage_hash;
  const int32_t *stage_wpos;
  uint32_t *q_hash;             // [F * qcap]
  int32_t *q_size;              // [F]
  int32_t qcap;
  int32_t sort_cap;             // power of two >= max records of a fragment
  int32_t tile_base;            // first tile of this pass (staging is indexed pass-locally)
  // the index lookup of the fragment's minimizers follows in the same workgroup (it was a kernel of its own: k_lookup)
  IndexView ix;
  uint32_t *q_off;              // [F*qcap] start of the list in pos_ridx
  uint32_t *q_cnt;              // [F*qcap] list length (0 when absent or too frequent)
  uint32_t *n_seeds;            // [F]
  int32_t rec_cap;              // k_query_fused: most records of a fragment it accepts (<= QF_CAP; FA_QF_CAP lowers it for the tests)
  int32_t exc_tiles;            // k_query_fused: the batch holds tiles with bytes outside ACGT (their records wait in the staging arrays)
};

constexpr int QS_TILES = 16;      // tiles of a fragment whose counts k_query_sketch fetches in one go
__device__ __forceinline__ void query_sketch_tail(const QuerySketchArgs &a, int f, uint32_t *buf, int n, uint32_t h0_in, int wpos0);
__global__ __launch_bounds__(MAP_THREADS) void k_query_sketch(QuerySketchArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  uint32_t *buf = (uint32_t *)lds;                 // [sort_cap]
  __shared__ uint32_t sh_h0;
  __shared__ int sh_wpos0;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int t0 = a.frag_tile_lo[f] - a.tile_base, t1 = a.frag_tile_lo[f + 1] - a.tile_base;
  // gather in order.  The counts of the (few) tiles of a fragment are fetched together, then all staged records: two
  // dependent round trips to HBM per fragment instead of two per tile
  __shared__ int sh_cnt[QS_TILES + 1];
  const int ntile = t1 - t0;
  int n = 0;
  if (ntile <= QS_TILES) {
    if (tid < ntile) sh_cnt[tid] = a.tile_count[t0 + tid];
    __syncthreads();
    int first = -1;
    for (int q = 0; q < ntile; q++) { if (first < 0 && sh_cnt[q] > 0) first = q; n += sh_cnt[q]; }
    for (int i = tid; i < n; i += blockDim.x) {
      int q = 0, o = i;
      while (o >= sh_cnt[q]) { o -= sh_cnt[q]; q++; }
      buf[i] = a.stage_hash[(size_t)(t0 + q) * TILE + o];
    }
    if (tid == 0 && first >= 0) { sh_h0 = a.stage_hash[(size_t)(t0 + first) * TILE]; sh_wpos0 = a.stage_wpos[(size_t)(t0 + first) * TILE]; }
  } else {
    for (int t = t0; t < t1; t++) {
      int c = a.tile_count[t];
      for (int i = tid; i < c; i += blockDim.x) buf[n + i] = a.stage_hash[(size_t)t * TILE + i];
      if (n == 0 && c > 0 && tid == 0) { sh_h0 = a.stage_hash[(size_t)t * TILE]; sh_wpos0 = a.stage_wpos[(size_t)t * TILE]; }
      n += c;
    }
  }
  __syncthreads();
  query_sketch_tail(a, f, buf, n, sh_h0, sh_wpos0);
}

// The records of a fragment, in position order in buf[0 .. n): the leading-run rule, sort by hash, unique, index lookup.
// (All threads of the workgroup; h0 / wpos0 = hash and window of record 0, ignored when n == 0.)
__device__ __forceinline__ void query_sketch_tail(const QuerySketchArgs &a, int f, uint32_t *buf, int n, uint32_t h0_in, int wpos0) {
  const int tid = threadIdx.x;
  __shared__ int sh_drop, sh_total;
  if (tid == 0) { sh_drop = n; }
  __syncthreads();
  // leading run: records 1..d equal to record 0's hash are dropped when record 0 sits at window 0
  if (n > 1 && wpos0 == 0) {
    uint32_t h0 = h0_in;
    int first_diff = n;
    for (int i = 1 + tid; i < n; i += blockDim.x) if (buf[i] != h0) { first_diff = i; break; }
    atomicMin(&sh_drop, first_diff);
    __syncthreads();
    int d = sh_drop - 1;
    __syncthreads();
    if (d > 0) {
      // overwrite the dropped records with copies of record 0 (duplicates vanish in the unique step)
      for (int i = 1 + tid; i <= d; i += blockDim.x) buf[i] = h0;
    }
  }
  __syncthreads();
  __shared__ uint8_t occ[MAP_THREADS];              // counted form: slot i of the sorted array holds a (distinct) hash
  const bool counted = n <= (int)blockDim.x;
  if (counted) {
    // the usual case (a 3 kb fragment holds ~240 minimizers): one record per thread, ranked by counting -- every thread
    // reads the same LDS words (broadcast, four at a time), two barriers instead of the 36 of a bitonic network.  The
    // rank is the number of SMALLER hashes only (a compare and an add-with-carry per entry; breaking ties by position
    // took twice the instructions, and this loop is a third of the kernel): equal hashes then land on one slot, the
    // slots behind it stay empty, and the occupied slots are exactly what std::unique would keep.
    const int n4 = (n + 3) & ~3;
    for (int i = n + tid; i < n4; i += blockDim.x) buf[i] = SEED_PAD;
    occ[tid] = 0;
    __syncthreads();
    const uint32_t x = tid < n ? buf[tid] : 0u;
    int rank = 0;
    const uint4 *b4 = (const uint4 *)buf;
    for (int j = 0; j < n4; j += 4) {
      const uint4 v = b4[j >> 2];
      rank += v.x < x ? 1 : 0;
      rank += v.y < x ? 1 : 0;
      rank += v.z < x ? 1 : 0;
      rank += v.w < x ? 1 : 0;
    }
    __syncthreads();
    if (tid < n) { buf[rank] = x; occ[rank] = 1; }
    __syncthreads();
  } else {
    uint32_t n32 = 1;
    while (n32 < (uint32_t)n) n32 <<= 1;
    if (n32 < 2) n32 = 2;
    for (uint32_t i = n + tid; i < n32; i += blockDim.x) buf[i] = SEED_PAD;
    __syncthreads();
    // a real hash may equal SEED_PAD (protein mode): harmless, the first n sorted entries are then the same multiset
    block_bitonic_sort(buf, n32);
  }
  // unique: keep buf[i] if i == 0 or differs from predecessor, among the first n sorted entries
  if (tid == 0) sh_total = 0;
  __syncthreads();
  uint32_t *out = a.q_hash + (size_t)f * a.qcap;
  for (int base = 0; base < n; base += blockDim.x) {
    int i = base + tid;
    bool keep = i < n && (counted ? occ[i] != 0 : (i == 0 || buf[i] != buf[i - 1]));
    const uint32_t val = i < n ? buf[i] : 0u;
    uint64_t bal = __ballot(keep);
    __shared__ int wave_cnt[MAP_THREADS / 64];
    int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = sh_total;
    for (int q = 0; q < wv; q++) off += wave_cnt[q];
    // (also kept in LDS, compacted in place -- every read of this round is behind the barrier above, and a slot that a
    // later round still reads is only ever written with its own value -- so that the lookup below need not read the
    // hashes back from global memory)
    if (keep) { const int pos = off + __popcll(bal & ((1ULL << lane) - 1ULL)); out[pos] = val; buf[pos] = val; }
    __syncthreads();
    if (tid == 0) { int tot = 0; for (int q = 0; q < MAP_THREADS / 64; q++) tot += wave_cnt[q]; sh_total += tot; }
    __syncthreads();
  }
  if (tid == 0) a.q_size[f] = sh_total;                             // (the largest of the pass is taken by seed_totals)
  // ---- lookup: one 16-byte table probe per distinct minimizer (unordered_map::find), strict `< freqThreshold` ----
  __syncthreads();                                                  // (the hashes this workgroup wrote to `out`)
  const int s = sh_total;
  __shared__ uint32_t red[MAP_THREADS / 64];
  uint32_t mine = 0;
  for (int j = tid; j < s; j += blockDim.x) {
    const uint32_t h = buf[j];
    uint32_t off = 0, cnt = 0;
    if (index_find(a.ix, h, off, cnt)) {
      if ((int64_t)cnt >= (int64_t)a.ix.freq_threshold) cnt = 0;   // strict `size < threshold` keeps the list
    }
    a.q_off[(size_t)f * a.qcap + j] = off;
    a.q_cnt[(size_t)f * a.qcap + j] = cnt;
    mine += cnt;
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((tid & 63) == 0) red[tid >> 6] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t n = 0;
    for (int q = 0; q < MAP_THREADS / 64; q++) n += red[q];
    a.n_seeds[f] = n;
  }
}

// Pass-level speculation.  A query pass is launched without intermediate host synchronisation, sized by what earlier
// passes needed (largest sketch, LDS seed slots, HBM seed scratch, loci and event capacities).  Kernels check those
// bounds on the device, skip the work that does not fit and raise a flag; the host reads the flags once at the end of
// the pass and, if any is set, grows the bounds and runs the pass again.
constexpr uint32_t SPEC_SMAX = 1, SPEC_SCRATCH = 2, SPEC_LOCI = 4, SPEC_EVENTS = 8, SPEC_QFUSE = 16;

// ----------------------------------------------------------------------------------------------------------
// K1 and the per-fragment sketch in ONE launch (query passes over plain-ACGT genomes, 4 <= w <= 64): workgroup f hashes
// the tiles of fragment f one after the other (skf_tile, fa_sketch_fast.hip.h), collects their records in LDS instead
// of the staging arrays, and goes on with query_sketch_tail.  On one 5 Mb query this is one resident round of 1666
// workgroups where k_sketch_fast (2.4 rounds of 4998 tile workgroups) + k_query_sketch (a round of latency chains) were
// two launches.  Batches with bytes outside ACGT run the byte-path kernel for the tiles that touch them first; this kernel takes
// their staged records over and hashes the plain tiles itself.  A fragment with more than QF_CAP records (low-complexity sequence under a small window) is marked
// (q_size = -1, SPEC_QFUSE via seed_totals): the pass is void and runs again through the two kernels.  The workgroups
// write nothing into the status block -- the zeroing workgroups of this very launch are clearing it.
// ----------------------------------------------------------------------------------------------------------
constexpr int QF_CAP = 1024;        // records of one fragment held in LDS (a power of two: the bitonic fallback sorts in place)
// (waves per SIMD: seven, except the two cells whose window loop needs a 73rd register -- k = 14 with w = 37 / 50 -- at six)
#define QF_WAVES(K, W) (((K) == 14 && (W) > 32) ? 6 : 7)
template <int KT, int WT>
__global__ __launch_bounds__(SK_THREADS, QF_WAVES(KT, WT)) void k_query_fused(SketchArgs a, QuerySketchArgs q, int F) {
  extern __shared__ __align__(16) unsigned char lds[];
  static_assert(SK_THREADS == MAP_THREADS, "one workgroup runs both halves");
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= F) {                                         // the zeroing workgroups of the pass (as in k_sketch_fast)
    if (a.clear.stamp && blockIdx.x == (uint32_t)F && tid == 0) a.clear.stamp[3] = 0;
    clear_ranges(a.clear, blockIdx.x - (uint32_t)F, gridDim.x - (uint32_t)F);
    return;
  }
  if (a.clear.stamp && blockIdx.x == 0 && tid == 0) a.clear.stamp[0] = __builtin_amdgcn_s_memrealtime();   // start of the pass
  constexpr bool TABLES = KT == 14 || KT == 16 || KT == 21;
  const int f = blockIdx.x;
  const int t0 = q.frag_tile_lo[f] - q.tile_base, t1 = q.frag_tile_lo[f + 1] - q.tile_base;
  const SkfLayout L = skf_layout(KT ? KT : a.k, WT ? WT : a.w);
  // the premix tables go to LDS once per fragment (and again behind a tile that used their bytes: one in ~70 at k = 16)
  auto put_tables = [&]() __attribute__((always_inline)) {
    if (TABLES) { const uint4 *src = (const uint4 *)d_premix.v; uint4 *dst = (uint4 *)(lds + L.table_off); dst[tid] = src[tid]; dst[tid + SK_THREADS] = src[tid + SK_THREADS]; }
  };
  put_tables();
  uint32_t *qbuf = (uint32_t *)(lds + L.total);                      // [QF_CAP]
  __shared__ int sh_count, sh_add, sh_wpos0;
  if (tid == 0) { sh_count = 0; sh_wpos0 = -1; }
  __syncthreads();
  struct Collected {                                                 // the records behind those of the fragment's earlier tiles
    uint32_t *qbuf; int base; int *add, *wpos0;
    __device__ __forceinline__ void count(int32_t n) const { *add = n; }
    __device__ __forceinline__ void emit(uint32_t i, uint32_t hash, int32_t wpos) const {
      const uint32_t o = (uint32_t)base + i;
      if (o < (uint32_t)QF_CAP) qbuf[o] = hash;
      if (o == 0) *wpos0 = wpos;
    }
  };
  const uint4 none = make_uint4(0, 0, 0, 0);
  for (int t = t0; t < t1; t++) {
    const Tile tile = a.tiles[t];
    if (tile.exc_n > 0) continue;                                     // (uniform) a tile with other bytes: taken over from the staging arrays below
    const bool intact = skf_tile<KT, WT>(a, tile, lds, false, none, none, Collected{qbuf, sh_count, &sh_add, &sh_wpos0});
    __syncthreads();
    if (tid == 0) sh_count += sh_add;
    if (!intact) put_tables();                                       // (uniform: a tile with k-mer-less positions used the tables' bytes)
    __syncthreads();
  }
  // tiles that touch a byte outside ACGT (an N run, an IUPAC code) were sketched from the byte image by k_sketch_tiles<0, true> in
  // the launch before this one: their staged records join the fragment's (the order of the records is immaterial from here
  // on: they are sorted, and the leading-run rule only ever replaces a hash by itself)
  if (q.exc_tiles) {
    for (int t = t0; t < t1; t++) {
      if (a.tiles[t].exc_n <= 0) continue;                            // (uniform)
      const int cnt = a.tile_count[t], base = sh_count;
      const uint32_t *sh = a.stage_hash + (size_t)t * TILE;
      for (int i = tid; i < cnt; i += SK_THREADS) if (base + i < QF_CAP) qbuf[base + i] = sh[i];
      __syncthreads();
      if (tid == 0) sh_count = base + cnt;
      __syncthreads();
    }
  }
  const int n = sh_count;
  if (n > q.rec_cap) {                                               // (uniform) void pass: seed_totals raises SPEC_QFUSE, the host repeats it unfused
    if (tid == 0) { q.q_size[f] = -1; q.n_seeds[f] = 0; }
    return;
  }
  query_sketch_tail(q, f, qbuf, n, n > 0 ? qbuf[0] : 0u, sh_wpos0);
}

// totals[0] = sum of seeds, [1] = largest fragment, [2] = HBM scratch words for fragments whose seeds do not fit the
// LDS slots of k_l1 (their offsets go to ovf_off).  One workgroup: thousands of same-address atomics from k_lookup cost
// more than this.  Also checks the speculated sketch-size and scratch bounds.
// (one workgroup of any size: a kernel of its own when k_l1 needs the scratch offsets, else an extra workgroup of k_l1)
// stats[0] = the largest query sketch of the pass, taken here from q_size (one writer, in a launch behind the one that
// zeroes the status block: the sketch kernels themselves must not write into that block -- k_query_fused shares its launch
// with the zeroing workgroups).  q_size[f] < 0 is k_query_fused's "more records than my LDS holds": SPEC_QFUSE.
__device__ __forceinline__ void seed_totals(const uint32_t *n_seeds, int64_t F, uint32_t lds_seed_cap, uint64_t *totals,
                                            uint32_t *ovf_off, int32_t *stats, int32_t spec_smax,
                                            uint64_t spec_scratch_words, unsigned long long *pinfo, const int32_t *q_size) {
  __shared__ unsigned long long sh_sum;
  __shared__ unsigned int sh_max, sh_any, sh_small, sh_mid, sh_tiny;
  __shared__ int sh_smax, sh_qf;
  if (threadIdx.x == 0) { sh_sum = 0; sh_max = 0; sh_any = 0; sh_smax = 0; sh_qf = 0; sh_small = 0; sh_mid = 0; sh_tiny = 0; }
  __syncthreads();
  unsigned long long sum = 0;
  unsigned int mx = 0, any = 0;
  int smx = 0, qf = 0;
  unsigned int c_small = 0, c_mid = 0, c_tiny = 0;                     // fragments per size class of k_l1 (L1_SMALL_HITS, L1_MID_HITS); up to half L1_SMALL_HITS
  for (int64_t f = threadIdx.x; f < F; f += blockDim.x) {
    const uint32_t n = n_seeds[f];
    sum += n; mx = max(mx, n); any |= n > lds_seed_cap;
    c_small += n <= L1_SMALL_HITS ? 1u : 0u; c_mid += (n > L1_SMALL_HITS && n <= L1_MID_HITS) ? 1u : 0u; c_tiny += n <= L1_SMALL_HITS / 2u ? 1u : 0u;
    ovf_off[f] = 0;
    const int s = q_size[f];
    smx = max(smx, s); qf |= s < 0 ? 1 : 0;
  }
  for (int d = 32; d > 0; d >>= 1) {
    sum += __shfl_down(sum, d); mx = max(mx, (unsigned int)__shfl_down((int)mx, d)); any |= (unsigned int)__shfl_down((int)any, d);
    smx = max(smx, __shfl_down(smx, d)); qf |= __shfl_down(qf, d);
    c_small += (unsigned int)__shfl_down((int)c_small, d); c_mid += (unsigned int)__shfl_down((int)c_mid, d); c_tiny += (unsigned int)__shfl_down((int)c_tiny, d);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sh_sum, sum); atomicMax(&sh_max, mx); atomicOr(&sh_any, any); atomicMax(&sh_smax, smx); atomicOr(&sh_qf, qf);
    atomicAdd(&sh_small, c_small); atomicAdd(&sh_mid, c_mid); atomicAdd(&sh_tiny, c_tiny);
  }
  __syncthreads();
  // scratch of the oversized fragments, in fragment order: exclusive prefix sum of their padded sizes
  __shared__ unsigned long long sh_words, sh_wave[16];
  if (threadIdx.x == 0) sh_words = 0;
  __syncthreads();
  if (sh_any) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int64_t f0 = 0; f0 < F; f0 += blockDim.x) {
      const int64_t f = f0 + threadIdx.x;
      const uint32_t n = f < F ? n_seeds[f] : 0u;
      unsigned long long n32 = 0;
      if (n > lds_seed_cap) { n32 = 1; while (n32 < n) n32 <<= 1; }
      unsigned long long incl = n32;
      for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) sh_wave[wv] = incl;
      __syncthreads();
      unsigned long long off = sh_words + incl - n32;
      for (int q = 0; q < wv; q++) off += sh_wave[q];
      if (n32) ovf_off[f] = (uint32_t)off;
      __syncthreads();
      if (threadIdx.x == 0) { unsigned long long t = 0; for (int q = 0; q < nw; q++) t += sh_wave[q]; sh_words += t; }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    totals[0] = sh_sum; totals[1] = sh_max;
    const unsigned long long words = sh_words;
    totals[2] = words;
    unsigned long long flags = 0;
    stats[0] = sh_smax;
    stats[1] = (int32_t)sh_small; stats[2] = (int32_t)sh_mid; stats[3] = (int32_t)sh_tiny;   // (what the host picks the size classes of k_l1 by)
    if (sh_smax > spec_smax) flags |= SPEC_SMAX;
    if (sh_qf) flags |= SPEC_QFUSE;
    if (words > spec_scratch_words || words >= (1ULL << 32)) flags |= SPEC_SCRATCH;
    if (flags) atomicOr(&pinfo[1], flags);
  }
}
__global__ __launch_bounds__(1024) void k_seed_totals(const uint32_t *n_seeds, int64_t F, uint32_t lds_seed_cap, uint64_t *totals,
                                                      uint32_t *ovf_off, int32_t *stats, int32_t spec_smax,
                                                      uint64_t spec_scratch_words, unsigned long long *pinfo, unsigned long long *stamp,
                                                      const int32_t *q_size) {
  stage_stamp(stamp);                                                // start of the L1 stage
  seed_totals(n_seeds, F, lds_seed_cap, totals, ovf_off, stats, spec_smax, spec_scratch_words, pinfo, q_size);
}

// ----------------------------------------------------------------------------------------------------------
// L1: gather + sort the seed hits of a fragment, scan for >= minHits hits inside one fragment length, merge
// overlapping candidates (computeL1CandidateRegions).  One workgroup per fragment; loci are appended to global
// arrays in (fragment-local) order, and consecutive loci on the same reference genome share a `group`.
// ----------------------------------------------------------------------------------------------------------
// Loci are numbered per REGION: fragment f reserves in region f mod n (n a power of two, one region per sixteen fragments, 64
// at most), region r holds the loci [r << shift, (r << shift) + count[r]).  One counter for everything made thousands of
// workgroups queue on one address for ~12 ns each -- 20 us at the end of k_l1 on a single query, whose 1 666 workgroups all
// arrive there within one resident round.  A locus number is live iff it lies inside the filled part of its region.
constexpr int LOCI_REGIONS = 64;
struct LociRegions {
  uint32_t *count;              // [n] loci reserved per region (may exceed the capacity: the pass is void then, counters[2])
  uint32_t n, shift;
};
__device__ __forceinline__ bool locus_live(const LociRegions &g, uint32_t l) {
  const uint32_t r = l >> g.shift;
  return r < g.n && (l & ((1u << g.shift) - 1u)) < g.count[r];
}
// `cnt` loci for fragment f: the number of the first one; `ok` = they fit the region
__device__ __forceinline__ uint32_t reserve_loci(const LociRegions &g, int f, uint32_t cnt, bool &ok) {
  const uint32_t r = (uint32_t)f & (g.n - 1u);
  const uint32_t off = atomicAdd(&g.count[r], cnt);
  ok = off + cnt <= (1u << g.shift);
  return (r << g.shift) + off;
}

struct L1Args {
  // the seed totals of the pass (seed_totals) as workgroup number F of this launch, when nothing in the pass needs the
  // scratch offsets they produce (no fragment has outgrown the LDS seed slots on this mapper yet)
  int32_t fold_totals, spec_smax;
  int64_t F;
  uint64_t *totals;
  int32_t *stats;
  uint64_t spec_scratch_words;
  unsigned long long *stamp;     // stage_stamp: start of the L1 stage (when the totals are folded in)
  IndexView ix;
  const int32_t *q_size;
  const uint32_t *q_off;
  const uint32_t *q_cnt;
  const uint32_t *n_seeds;
  uint32_t *ovf_off;
  uint32_t *ovf_buf;
  const int32_t *min_hits_lut;   // [smax+1]
  int32_t *l_frag, *l_seq, *l_start, *l_end, *l_group;   // loci, capacity l_cap
  int32_t *l_rfirst, *l_rlast;   // record index of the first / last seed of the locus
  int32_t *l_rpart;              // record index of the seed that fixed the locus start (the partner of its first seed); -1 = unknown
  uint32_t *counters;            // [2] loci overflow flag
  LociRegions loci;              // where the loci of a fragment are reserved (a group is numbered by its first locus)
  uint32_t *f_loci_lo, *f_loci_n; // [F] loci of each fragment (contiguous)
  unsigned long long *pinfo;     // [1] speculation flags
  int32_t lut_smax;              // sketch sizes the LUTs cover
  uint64_t scratch_words;        // capacity of ovf_buf
  int32_t qcap, frag_len, l_cap;
  uint32_t lds_seed_cap;
  // k_l1 is launched once per SIZE CLASS of fragments (round 6): this launch takes the fragments with n_lo <= hits <= n_hi and
  // leaves the others alone.  One launch sized for the largest fragment of the pass put every fragment of a genome-like
  // workload -- where 3 % of the fragments touch a repeat and collect tens of thousands of hits -- through the 512-thread,
  // 32-hits-per-thread form at two workgroups per CU: 15.2 ms of lookup + L1 per step where the i.i.d. cell of the same shape
  // takes 5.1 (profiles/r06_genome_like_*).  totals_seed_cap: the slots of the LAST class -- what `seed_totals` (folded into
  // the first launch) and k_l1_big mean by "more hits than LDS holds".
  uint32_t n_lo, n_hi, totals_seed_cap;
  unsigned long long *dbg;       // [8] FA_L1_STATS=1: ticks of the phases of k_l1
  int32_t block_sort;            // k_l1: bit 0 = try l1_block_sort before the merge (FA_L1_BLOCK_SORT=0 switches it off), bit 1 = count the roads taken,
                                 // bit 2 = skip the coordinate fetch of hits that cannot be an end of a candidate (FA_L1_NEAR=0: off),
                                 // bit 3 = drop hits that cannot belong to a candidate before the block sort (FA_L1_PREFILTER)
  uint8_t *big_state;            // [F] k_l1_big: 1 = this fragment was handled there, 0 = not (k_l1 takes it)
  int32_t big_enabled;           // k_l1_big ran before k_l1 in this pass
  uint32_t big_cap;              // seed hits per chunk of k_l1_big (<= L1_BIG_E x L1_BIG_THREADS)
};

// dynamic LDS of k_l1: the seed hits [cap], the list offsets and sources [lut_smax + 2 each], the staged loci (6 arrays
// of L1_STAGE)
constexpr int L1_INPLACE_MAX = 32;  // most seeds per thread the in-place merge keeps in registers (template parameter E: 16 or 32)
__host__ __device__ inline size_t l1_off_offset(uint32_t seed_cap) { return ((size_t)seed_cap * 4 + 15) / 16 * 16; }
// (nt = threads of the workgroup: once the lists are merged the same bytes hold (contig, window) of a trip's seeds by thread)
__host__ __device__ inline size_t l1_stage_offset(uint32_t seed_cap, int lut_smax, int nt) {
  // list offsets + list sources; later the saved ballots of the candidate scan (16 bytes per 64-candidate step) and -- ten bytes
  // per thread, so that with the locus stage behind it the region holds a (key, place) pair for 1 024 blocks at 512 threads --
  // the key buffer of l1_block_sort
  const size_t lists = ((size_t)lut_smax + 2) * 8, trip = (size_t)nt * 10;
  return (l1_off_offset(seed_cap) + (lists > trip ? lists : trip) + 15) / 16 * 16;
}
__host__ __device__ inline size_t l1_lds_bytes(uint32_t seed_cap, int lut_smax, int nt) {
  return l1_stage_offset(seed_cap, lut_smax, nt) + (size_t)L1_STAGE * 6 * 4;
}

// Sorting the seed hits of a fragment WITHOUT merging them, for the regime the workloads live in: the hits of a fragment are
// record numbers, records are ordered by (contig, window), and a query fragment that has relatives in the index finds them in
// a few dozen STRETCHES of consecutive records (one per related genome, about a fragment's worth of records each) -- 4 000
// hits fall into some 250 blocks of 64 consecutive records.  So the hits are not sorted, their blocks are:
//   1. every hit sets its bit in an open-addressing LDS table {block number + 1, 64-bit bitmap} (one returning and one
//      plain LDS atomic per hit; a record sits in one position list only, so no bit is set twice);
//   2. the block numbers of the occupied entries -- a few hundred -- are bitonic-sorted in REGISTERS (lane exchanges by DPP /
//      swizzle, no LDS round trip per stage; bs_sizes);
//   3. every sorted block number fetches its bitmap back from the table, a prefix sum over the population counts gives it
//      its place, and the bits are expanded into the sorted array of record numbers -- the array the merge produces, bit
//      for bit, for a fraction of the instructions and, what counts more in this kernel, of the dependent LDS round trips
//      (eight merge levels: two binary searches and 8-9 sequential merge steps per thread and level).
// The table and the sorted hits share the LDS of the seed slots (12 bytes per entry: a third as many entries as slots; any
// number -- slots are picked by multiply-high, not by a mask); the keys pass through the bytes behind them (list offsets +
// locus stage).
// Returns false -- the caller then gathers and merges as before -- when the hits are too scattered (a probe sequence longer
// than BS_MAX_PROBES, or more blocks than the key buffer and the registers of the sort hold: 1 024 at 256 threads, 2 048 at
// 512) or the counts do not add up.  The verdict is uniform over the workgroup.

// inclusive prefix sum over the 64 lanes of a wave by DPP (row shifts inside the rows of 16 lanes, then the row totals
// broadcast into the rows behind them): six vector instructions, no LDS crossbar -- `__shfl_up` is a ds_bpermute each
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1 (lanes without a source add 0)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// lane exchange across a power-of-two distance below 64 (the value of lane ^ X)
template <int X>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if constexpr (X == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (X == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (X < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1F);           // bit mode: lane ^ X inside 32 lanes
  else return (uint32_t)__shfl_xor((int)v, 32);
}
// one stage (SIZE, STRIDE) of the bitonic network over keys in registers: KPL keys per lane, key j of lane l of wave w at
// position pos0 + 64 j with pos0 = w x 64 KPL + l
template <int NT, int KPL, int SIZE, int STRIDE>
__device__ __forceinline__ void bs_strides(uint32_t (&k)[KPL], uint32_t pos0, bool active, uint32_t *Kk) {
  if constexpr (STRIDE >= 64 * KPL) {
    // across wave segments: through the key buffer
    if (active) {
#pragma unroll
      for (int j = 0; j < KPL; j++) Kk[pos0 + (uint32_t)j * 64u] = k[j];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < KPL; j++) {
        const uint32_t pos = pos0 + (uint32_t)j * 64u;
        const uint32_t p = Kk[pos ^ (uint32_t)STRIDE];
        const bool up = (pos & (uint32_t)SIZE) == 0, lower = (pos & (uint32_t)STRIDE) == 0;
        k[j] = (up == lower) ? min(k[j], p) : max(k[j], p);
      }
    }
    __syncthreads();
  } else if constexpr (STRIDE >= 64) {
    if (active) {
      constexpr int dj = STRIDE / 64;
#pragma unroll
      for (int j = 0; j < KPL; j++) {
        if ((j & dj) == 0) {
          const bool up = ((pos0 + (uint32_t)j * 64u) & (uint32_t)SIZE) == 0;
          const uint32_t x = k[j], y = k[j | dj];
          const uint32_t lo = min(x, y), hi = max(x, y);
          k[j] = up ? lo : hi; k[j | dj] = up ? hi : lo;
        }
      }
    }
  } else {
    if (active) {
#pragma unroll
      for (int j = 0; j < KPL; j++) {
        const uint32_t pos = pos0 + (uint32_t)j * 64u;
        const uint32_t p = lane_xor<STRIDE>(k[j]);
        const bool up = (pos & (uint32_t)SIZE) == 0, lower = (pos & (uint32_t)STRIDE) == 0;
        k[j] = (up == lower) ? min(k[j], p) : max(k[j], p);
      }
    }
  }
  if constexpr (STRIDE > 1) bs_strides<NT, KPL, SIZE, STRIDE / 2>(k, pos0, active, Kk);
}
template <int NT, int KPL, int SIZE>
__device__ __forceinline__ void bs_sizes(uint32_t (&k)[KPL], uint32_t pos0, bool active, uint32_t nb32, uint32_t *Kk) {
  if ((uint32_t)SIZE <= nb32) {                                          // (uniform over the workgroup)
    bs_strides<NT, KPL, SIZE, SIZE / 2>(k, pos0, active, Kk);
    if constexpr (SIZE < 64 * KPL * (NT / 64)) bs_sizes<NT, KPL, SIZE * 2>(k, pos0, active, nb32, Kk);
  }
}

constexpr uint32_t BS_MAX_PROBES = 48;
// THE PRE-FILTER (bit 3 of L1Args::block_sort; the host sets it for indices of 10^9 records and more, FA_L1_PREFILTER=0 / 1 forces it).
// In an index of thousands of genomes a fragment collects ~2 000 chance hits, each a block of its own, next to the ~250 blocks of
// its relatives: more blocks than the register sort holds, and the fragment fell back to the merge (4000 x 4000 genomes: 55 % of
// the step).  A hit can belong to a candidate only if minHits - 1 other hits lie within a fragment length of WINDOW positions,
// hence of RECORDS (window positions grow by at least one per record), hence inside one cell of one of two staggered grids of
// cells twice a fragment length wide.  With minHits >= 2 a hit whose two cells hold no other hit is dead, and dropping dead hits
// neither removes a run of minHits hits that spans less than a fragment nor creates one (tests/test_l1_prefilter_model.py checks
// the claim against the oracle's candidate regions).  A first sweep over the hits marks "seen once" / "seen twice" bits over a
// hash of the cell number (collisions only keep more); the second sweep inserts the hits with a "seen twice" cell and counts them:
// `n_live` replaces n for everything behind the sort.  The bit arrays sit at the end of the key buffer, which the sweeps do not
// use yet.  One more sweep over the position lists (from L2 the second time) where it is on; nothing where it is off.
template <int NT, int SPT, int KPL>
__device__ __forceinline__ bool l1_block_sort(const L1Args &a, int s, uint32_t n, uint32_t cap, uint32_t *A, const uint32_t *off, const uint32_t *qo,
                                              uint32_t *Kk, uint32_t kl, uint32_t &n_live, int frag_len) {
  __shared__ uint32_t bs_fail;
  __shared__ uint32_t bs_wsum[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t capT = cap / 3u;                                        // table entries (12 bytes each; any number: slots are picked by multiply-high)
  unsigned long long *Tb = (unsigned long long *)A;                      // [capT] bitmaps of 64 consecutive records
  uint32_t *Tk = (uint32_t *)(Tb + capT);                                // [capT] block number + 1 (0 = empty)
  auto slot_of = [&](uint32_t key) __attribute__((always_inline)) { return __umulhi(key * 0x9E3779B1u, capT); };
  auto next_slot = [&](uint32_t h) __attribute__((always_inline)) { return h + 1u == capT ? 0u : h + 1u; };
  const bool dbg = (a.block_sort & 2) && tid == 0 && (blockIdx.x & 63) == 0;
  long long tk = dbg ? clock64() : 0;
  auto phase = [&](int k) __attribute__((always_inline)) {
    if (dbg) { const long long now = clock64(); atomicAdd(&a.dbg[k], (unsigned long long)(now - tk)); tk = now; }
  };
  for (uint32_t i = tid; i < capT; i += NT) { Tb[i] = 0ULL; Tk[i] = 0u; }
  if (tid == 0) bs_fail = 0;
  __shared__ uint32_t bs_live;
  // two bit arrays of PF_WORDS words each at the end of the key buffer, behind the list offsets and sources the sweeps read
  // (what is left of the region: ~700 words each at 512 threads and sketches of ~270)
  const uint32_t lists = 2u * (uint32_t)(a.lut_smax + 2);
  const uint32_t PF_WORDS = kl > lists ? ((kl - lists) / 2u) & ~31u : 0u, PF_BITS = PF_WORDS * 32u;
  int mh = a.min_hits_lut[s];
  const bool prefilter = (a.block_sort & 8) && PF_WORDS >= 256u && frag_len > 0 && mh >= 2;   // (minHits 1: every hit is a candidate)
  uint32_t *B1 = Kk + (kl - 2u * PF_WORDS), *B2 = B1 + PF_WORDS;
  int cell_shift = 1;                                                    // cells of 2^cell_shift records, 2^(cell_shift - 1) >= frag_len
  while ((1 << (cell_shift - 1)) < frag_len && cell_shift < 30) cell_shift++;
  auto cell_bit = [&](uint32_t r, uint32_t g) __attribute__((always_inline)) {
    const uint32_t c = (((r + (g << (cell_shift - 1))) >> cell_shift) << 1) | g;
    return __umulhi(c * 0x9E3779B1u, PF_BITS);
  };
  n_live = n;
  if (prefilter) {
    for (uint32_t i = tid; i < 2u * PF_WORDS; i += NT) B1[i] = 0u;
    if (tid == 0) bs_live = 0;
  }
  __syncthreads();
  if (prefilter) {
    // the loop of step 1 below, twice: marks, then insertions of the live hits
    constexpr int IB = 8;
    uint32_t top = 1;
    while (top * 2u < (uint32_t)s) top <<= 1;
    if (s < 2) top = 0;
    auto place = [&](uint32_t r) __attribute__((always_inline)) {
      const uint32_t key1 = (r >> 6) + 1u;
      const unsigned long long bit = 1ULL << (r & 63u);
      uint32_t h = slot_of(r >> 6);
      for (uint32_t probes = 0;; probes++) {
        const uint32_t old = atomicCAS(&Tk[h], 0u, key1);
        if (old == 0u || old == key1) { atomicOr(&Tb[h], bit); break; }
        if (probes >= BS_MAX_PROBES) { bs_fail = 1; break; }
        h = next_slot(h);
      }
    };
    uint32_t mine_live = 0;
#pragma unroll 1
    for (int sweep = 0; sweep < 2; sweep++) {
      for (uint32_t i0 = tid; i0 < n; i0 += IB * NT) {
        uint32_t jj[IB], r[IB];
#pragma unroll
        for (int u = 0; u < IB; u++) jj[u] = 0;
        for (uint32_t step = top; step; step >>= 1) {
          uint32_t v[IB];
#pragma unroll
          for (int u = 0; u < IB; u++) v[u] = off[min(jj[u] + step, (uint32_t)s)];
#pragma unroll
          for (int u = 0; u < IB; u++) jj[u] += v[u] <= i0 + (uint32_t)u * NT ? step : 0u;
        }
#pragma unroll
        for (int u = 0; u < IB; u++) {
          const uint32_t i = i0 + (uint32_t)u * NT;
          r[u] = i < n ? a.ix.pos_ridx[qo[jj[u]] + (i - off[jj[u]])] : 0u;
        }
#pragma unroll
        for (int u = 0; u < IB; u++) {
          if (i0 + (uint32_t)u * NT >= n) continue;
          const uint32_t b0 = cell_bit(r[u], 0u), b1 = cell_bit(r[u], 1u);
          if (sweep == 0) {
            const uint32_t o0 = atomicOr(&B1[b0 >> 5], 1u << (b0 & 31u));
            if (o0 & (1u << (b0 & 31u))) atomicOr(&B2[b0 >> 5], 1u << (b0 & 31u));
            const uint32_t o1 = atomicOr(&B1[b1 >> 5], 1u << (b1 & 31u));
            if (o1 & (1u << (b1 & 31u))) atomicOr(&B2[b1 >> 5], 1u << (b1 & 31u));
          } else if (((B2[b0 >> 5] >> (b0 & 31u)) | (B2[b1 >> 5] >> (b1 & 31u))) & 1u) {
            place(r[u]);
            mine_live++;
          }
        }
        if (sweep == 1 && *(volatile uint32_t *)&bs_fail) break;
      }
      __syncthreads();
    }
    {
      const uint32_t incl = wave_incl_scan(mine_live);
      if (lane == 63 && incl) atomicAdd(&bs_live, incl);
    }
    __syncthreads();
    n_live = bs_live;
    n = n_live;                                                          // (everything below counts the live hits)
    if (dbg) atomicAdd(&a.dbg[14], (unsigned long long)n_live);
  } else
  // ---- 1. the hits, flat: hit i of the fragment is entry i - off[j] of list j (off = prefix sums of the list lengths, qo = where
  //      every list starts in the index; the caller left both in LDS).  Eight hits per thread and trip, so that eight index
  //      reads are in flight: the workgroup's time is a chain of memory round trips, not instructions ----
  {
    auto place = [&](uint32_t r) __attribute__((always_inline)) {
      const uint32_t key1 = (r >> 6) + 1u;
      const unsigned long long bit = 1ULL << (r & 63u);
      uint32_t h = slot_of(r >> 6);
      for (uint32_t probes = 0;; probes++) {
        const uint32_t old = atomicCAS(&Tk[h], 0u, key1);
        if (old == 0u || old == key1) { atomicOr(&Tb[h], bit); break; }
        if (probes >= BS_MAX_PROBES) { bs_fail = 1; break; }
        h = next_slot(h);
      }
    };
    constexpr int IB = 8;
    uint32_t top = 1;                                                    // largest power of two below s (0 steps when s == 1)
    while (top * 2u < (uint32_t)s) top <<= 1;
    if (s < 2) top = 0;
    for (uint32_t i0 = tid; i0 < n; i0 += IB * NT) {
      // the list of every hit: the last j with off[j] <= i, found for all eight hits in LOCK STEP -- a fixed number of probes, so
      // that eight LDS reads are in flight per probe instead of eight searches one behind the other
      uint32_t jj[IB], r[IB];
#pragma unroll
      for (int u = 0; u < IB; u++) jj[u] = 0;
      for (uint32_t step = top; step; step >>= 1) {
        uint32_t v[IB];
#pragma unroll
        for (int u = 0; u < IB; u++) v[u] = off[min(jj[u] + step, (uint32_t)s)];   // (unconditional reads, one block: all eight in flight; off[s] = n > i)
#pragma unroll
        for (int u = 0; u < IB; u++) jj[u] += v[u] <= i0 + (uint32_t)u * NT ? step : 0u;
      }
#pragma unroll
      for (int u = 0; u < IB; u++) {
        const uint32_t i = i0 + (uint32_t)u * NT;
        r[u] = i < n ? a.ix.pos_ridx[qo[jj[u]] + (i - off[jj[u]])] : 0u;
      }
#pragma unroll
      for (int u = 0; u < IB; u++) if (i0 + (uint32_t)u * NT < n) place(r[u]);
      if (*(volatile uint32_t *)&bs_fail) break;
    }
  }
  __syncthreads();
  phase(5);
  if (bs_fail) { if (dbg) atomicAdd(&a.dbg[8], 1ULL); __syncthreads(); return false; }
  // ---- 2. the keys of the occupied entries, compacted into the key buffer (the table stays where it is) ----
  uint32_t mine = 0;                                                     // (counted now, read again for the copy: eight keys held across
#pragma unroll                                                           //  the scan below were the registers this kernel was short of)
  for (int q = 0; q < SPT; q++) {
    const uint32_t slot = (uint32_t)q * NT + tid;
    mine += (slot < capT && Tk[slot] != 0u) ? 1u : 0u;
  }
  auto block_exclusive = [&](uint32_t v, uint32_t &total) __attribute__((always_inline)) {
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) bs_wsum[wv] = incl;
    __syncthreads();
    // the waves' totals: one read per lane and a scan (a loop over the waves kept all of them in registers at once)
    const uint32_t wt = lane < NT / 64 ? bs_wsum[lane] : 0u;
    const uint32_t wincl = wave_incl_scan(wt);
    const uint32_t base = incl - v + (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wt), wv);
    total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
    __syncthreads();
    return base;
  };
  uint32_t nb = 0;
  uint32_t at = block_exclusive(mine, nb);
  uint32_t nb32 = 64u * KPL; while (nb32 < nb) nb32 <<= 1;               // (whole wave segments of 64 x KPL keys)
  if (nb32 > kl || nb32 > 64u * KPL * (NT / 64)) { if (dbg) atomicAdd(&a.dbg[9], 1ULL); return false; }   // more blocks than the registers of the sort hold
#pragma unroll
  for (int q = 0; q < SPT; q++) {
    const uint32_t slot = (uint32_t)q * NT + tid;
    const uint32_t key1 = slot < capT ? Tk[slot] : 0u;
    if (key1 != 0u) Kk[at++] = key1;
  }
  for (uint32_t i = nb + tid; i < nb32; i += NT) Kk[i] = 0xFFFFFFFFu;
  __syncthreads();
  phase(6);
  // ---- 3. bitonic sort of the keys IN REGISTERS: wave w holds keys [w x 64 KPL, (w+1) x 64 KPL), lane l the keys l, 64 + l, ...
  //      of that segment; strides below 64 are lane exchanges (DPP / swizzle: no LDS memory, no barrier), 64 and up to the
  //      segment are compare-exchanges between a lane's own registers, and only strides that cross segments go through the
  //      key buffer (two barriers each: one stage for 512 keys on two waves, none for 256).  The same network stage by
  //      stage through LDS took a third of this kernel: every stage was an LDS round trip under load ----
  const uint32_t pos0 = (uint32_t)wv * 64u * KPL + (uint32_t)lane;
  const bool active = pos0 < nb32;                                       // (uniform per wave)
  uint32_t k[KPL];
#pragma unroll
  for (int j = 0; j < KPL; j++) k[j] = active ? Kk[pos0 + (uint32_t)j * 64u] : 0xFFFFFFFFu;
  bs_sizes<NT, KPL, 2>(k, pos0, active, nb32, Kk);
  phase(7);
  // ---- 4. the bitmap of every sorted key (one more probe of the table), places from the population counts, bits expanded
  //      into the sorted record numbers.  The sort leaves the keys on a few waves (256 per wave); where the key buffer has
  //      room the (key, place) pairs go back through it so that EVERY thread expands its share of the entries ----
  auto bitmap_of = [&](uint32_t key1) __attribute__((always_inline)) {
    uint32_t h = slot_of(key1 - 1u);
    while (Tk[h] != key1) h = next_slot(h);
    return Tb[h];
  };
  uint32_t place_of[KPL];
  uint32_t wave_total = 0;
#pragma unroll
  for (int j = 0; j < KPL; j++) {
    const uint32_t c = (active && k[j] != 0xFFFFFFFFu) ? (uint32_t)__popcll(bitmap_of(k[j])) : 0u;   // (the bitmap itself is fetched again where it is expanded)
    const uint32_t incl = wave_incl_scan(c);
    place_of[j] = wave_total + incl - c;
    wave_total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  if (lane == 0) bs_wsum[wv] = wave_total;
  __syncthreads();                                                       // (also: these probes of the table are done)
  const uint32_t wt = lane < NT / 64 ? bs_wsum[lane] : 0u;
  const uint32_t wincl = wave_incl_scan(wt);
  uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wt), wv);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
  if (total != n) { if (dbg) atomicAdd(&a.dbg[10], 1ULL); return false; }   // (cannot happen: a record sits in one list only)
  phase(11);
  // (an entry is expanded as two 32-bit halves: a 64-bit lowest-set-bit loop costs twice the instructions per hit)
  auto expand = [&](uint32_t half, uint32_t first, uint32_t o) __attribute__((always_inline)) {
    while (half) { A[o++] = first | (uint32_t)(__ffs((int)half) - 1); half &= half - 1u; }
  };
  constexpr int HPT = 4;                                                 // halves per thread at most (registers)
  const bool spread = 2u * nb32 <= kl && 2u * nb32 <= (uint32_t)(HPT * NT);  // room for the pairs, and at most HPT halves per thread
  if (spread) {
    uint2 *KP = (uint2 *)Kk;
    if (active) {
#pragma unroll
      for (int j = 0; j < KPL; j++) KP[pos0 + (uint32_t)j * 64u] = make_uint2(k[j], base + place_of[j]);
    }
    __syncthreads();
    uint32_t half[HPT], first[HPT], at[HPT];
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      const uint32_t e = (uint32_t)j * NT + tid;                         // half e & 1 of entry e >> 1
      const uint2 kp = (e >> 1) < nb ? KP[e >> 1] : make_uint2(0xFFFFFFFFu, 0u);
      const unsigned long long b = kp.x != 0xFFFFFFFFu ? bitmap_of(kp.x) : 0ULL;
      const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
      half[j] = (e & 1u) ? hi : lo;
      first[j] = ((kp.x - 1u) << 6) | ((e & 1u) << 5);
      at[j] = kp.y + ((e & 1u) ? (uint32_t)__popc(lo) : 0u);
    }
    __syncthreads();                                                     // (every probe of the table is done)
    phase(12);
#pragma unroll
    for (int j = 0; j < HPT; j++) expand(half[j], first[j], at[j]);
  } else {
    unsigned long long bits[KPL];
#pragma unroll
    for (int j = 0; j < KPL; j++) bits[j] = (active && k[j] != 0xFFFFFFFFu) ? bitmap_of(k[j]) : 0ULL;
    __syncthreads();                                                     // (every probe of the table is done)
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const uint32_t lo = (uint32_t)bits[j], hi = (uint32_t)(bits[j] >> 32), f0 = (k[j] - 1u) << 6, o = base + place_of[j];
      expand(lo, f0, o);
      expand(hi, f0 | 32u, o + (uint32_t)__popc(lo));
    }
  }
  __syncthreads();
  phase(13);
  return true;
}

// NT threads per workgroup (256 measured best: wider workgroups pay more for the cross-wave scans and barriers).
// E = elements per thread the in-place merge can hold (16 for small fragments: fewer registers, more workgroups per CU).
template <int NT, int E>
__global__ __launch_bounds__(NT, (E == 16 ? 8 : 4)) void k_l1(L1Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  if (a.fold_totals) {
    stage_stamp(a.stamp);
    if ((int64_t)blockIdx.x == a.F) {
      seed_totals(a.n_seeds, a.F, a.totals_seed_cap, a.totals, a.ovf_off, a.stats, a.spec_smax, a.spec_scratch_words, a.pinfo, a.q_size);
      return;
    }
  }
  __shared__ uint32_t sh_scan[NT / 64];
  __shared__ uint32_t sh_run;       // running offset (gather)
  // candidate pass: running head count and the last flagged candidate so far, double-buffered by trip parity so that
  // thread 0 can publish the next trip's values while slower waves still read this trip's
  __shared__ uint32_t sh_heads[2];
  __shared__ int sh_has_prev[2];
  __shared__ uint64_t sh_prev_g[2];
  __shared__ uint32_t sh_base, sh_gbase;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int s = a.q_size[f];
  uint32_t n = a.n_seeds[f];                     // (the LIVE hits once the pre-filter of the block sort has dropped the dead ones)
  if (n < a.n_lo || n > a.n_hi) return;          // a fragment of another size class: another launch of this pass takes it
  if (a.big_enabled && a.big_state[f]) return;   // more hits than LDS holds: cut into chunks by k_l1_big, which ran before
  if (tid == 0) { a.f_loci_lo[f] = 0; a.f_loci_n[f] = 0; }
  if (s == 0 || n == 0) return;
  if (s > a.lut_smax) return;                    // SPEC_SMAX was raised by k_seed_totals: the pass will be repeated
  uint32_t n32 = 1; while (n32 < n) n32 <<= 1;
  if (n32 < 2) n32 = 2;
  const bool in_lds = n <= a.lds_seed_cap;                               // (the host keeps lds_seed_cap <= E x NT)
  if (!in_lds && (uint64_t)a.ovf_off[f] + n32 > a.scratch_words) return;   // SPEC_SCRATCH, same
  uint32_t *seeds;
  const bool l1_dbg = (a.block_sort & 2) && tid == 0 && (f & 63) == 0;     // (one workgroup in 64: the atomics below must not become the workload)
  long long tk = l1_dbg ? clock64() : 0;
  auto phase = [&](int k) __attribute__((always_inline)) {
    if (l1_dbg) { const long long now = clock64(); atomicAdd(&a.dbg[k], (unsigned long long)(now - tk)); tk = now; }
  };
  if (in_lds) {
    // ---- the position lists of the query minimizers are each sorted already (CSR order = record order): gather them
    //      back to back and merge them pairwise, bottom-up (record indices are unique: no ties).  A run is a group of
    //      2^k consecutive lists, so its bounds come from the prefix sums of the list lengths. ----
    const uint32_t cap = a.lds_seed_cap;
    uint32_t *A = (uint32_t *)lds;
    uint32_t *off = (uint32_t *)(lds + l1_off_offset(cap));             // [s + 1] first seed of every list
    uint32_t *qo = off + a.lut_smax + 2;                                 // [s] where every list starts in the index
    auto list_offsets = [&]() __attribute__((always_inline)) {
      if (tid == 0) sh_run = 0;
      __syncthreads();
      for (int j0 = 0; j0 < s; j0 += NT) {
        const int j = j0 + tid;
        const uint32_t cnt = j < s ? a.q_cnt[(size_t)f * a.qcap + j] : 0;
        const uint32_t src = (j < s && cnt) ? a.q_off[(size_t)f * a.qcap + j] : 0u;
        const uint32_t incl = wave_incl_scan(cnt);
        if (lane == 63) sh_scan[wv] = incl;
        __syncthreads();
        uint32_t o = sh_run + incl - cnt;
        for (int q = 0; q < wv; q++) o += sh_scan[q];
        if (j < s) { off[j] = o; qo[j] = src; }
        __syncthreads();
        if (tid == 0) { uint32_t tot = 0; for (int q = 0; q < NT / 64; q++) tot += sh_scan[q]; sh_run += tot; }
        __syncthreads();
      }
      if (tid == 0) off[s] = n;
      __syncthreads();
    };
    list_offsets();
    phase(0);
    // hits that cluster in stretches of the index (the usual case) are sorted block-wise: l1_block_sort (its key buffer is
    // everything behind the seed slots -- the list offsets, which it has used by then, and the locus stage)
    bool block_sorted = false;
    if ((a.block_sort & 1) && cap >= 1024u) {
      const uint32_t kl = (uint32_t)((l1_lds_bytes(cap, a.lut_smax, NT) - l1_off_offset(cap)) / 4);
      uint32_t n_live = n;
      block_sorted = l1_block_sort<NT, E / 2, 4>(a, s, n, cap, A, off, qo, off, kl, n_live, a.frag_len);
      if (block_sorted) n = n_live;
      if (!block_sorted) list_offsets();                                // (the key buffer may have overwritten them)
    }
    if (l1_dbg) atomicAdd(&a.counters[block_sorted ? 5 : 6], 1u);   // FA_L1_STATS=1: which road the fragments took
    // fragments that left the fast form (block sort -> gather + merge), counted exactly and always: one atomic on the rare road
    if (!block_sorted && tid == 0) atomicAdd(&a.counters[0], 1u);
    phase(1);
    if (!block_sorted) {
    // flat gather, two elements per thread and trip so that two index reads are in flight
    auto locate = [&](uint32_t i) __attribute__((always_inline)) {
      int lo = 0, hi = s - 1;                                              // the list j with off[j] <= i < off[j + 1]
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (off[mid + 1] <= i) lo = mid + 1; else hi = mid; }
      return lo;
    };
    for (uint32_t i0 = tid; i0 < n; i0 += 2 * NT) {
      const uint32_t i1 = i0 + NT;
      const int j0 = locate(i0), j1 = i1 < n ? locate(i1) : 0;
      const uint32_t v0 = a.ix.pos_ridx[qo[j0] + (i0 - off[j0])];
      const uint32_t v1 = i1 < n ? a.ix.pos_ridx[qo[j1] + (i1 - off[j1])] : 0u;
      A[i0] = v0;
      if (i1 < n) A[i1] = v1;
    }
    __syncthreads();
    // Merge path, in place.  Thread t produces the outputs [t x per, (t + 1) x per) of every level: it finds where that
    // range starts on the merge path of its pair of runs (ONE binary search along the diagonal), then merges `per`
    // elements sequentially -- one LDS read and a compare per output, the same number of steps in every lane -- keeps
    // them in registers across the barrier that separates the reads of a level from its writes, and writes them back.
    // 4 bytes of LDS per seed; the chain of dependent LDS round trips per level is (two searches) + per.
    // outputs per thread (<= E + 1), made ODD: thread t reads and writes around A[t x per], and with an even stride the
    // 64 lanes of a wave share 32 / gcd(per, 64) ... banks (per = 12: four-way, per = 16: sixteen-way conflicts)
    const uint32_t per = ((n + NT - 1) / NT) | 1u;
    const uint32_t o_lo = (uint32_t)tid * per;
    for (int k = 0; (1 << k) < s; k++) {
      uint32_t out[E + 1];
      const int w2 = 2 << k;                                             // lists per pair of runs
      const int npairs = (s + w2 - 1) / w2;
      uint32_t A0 = 0, A1 = 0, A2 = 0, ia = 0, ib = 0;
      int p = 0;
      if (o_lo < n) {
        // the pair that holds output o_lo: the last one starting at or before it
        int lo = 0, hi = npairs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[min(mid * w2, s)] <= o_lo) lo = mid; else hi = mid - 1; }
        p = lo;
        A0 = off[min(p * w2, s)]; A1 = off[min(p * w2 + (w2 >> 1), s)]; A2 = off[min((p + 1) * w2, s)];
        // merge path: how many of the first (o_lo - A0) outputs of the pair come from its left run
        const uint32_t diag = o_lo - A0, lenA = A1 - A0, lenB = A2 - A1;
        uint32_t l = diag > lenB ? diag - lenB : 0u, h = min(diag, lenA);
        while (l < h) { const uint32_t mid = (l + h) >> 1; if (A[A0 + mid] < A[A1 + diag - 1 - mid]) l = mid + 1; else h = mid; }
        ia = l; ib = diag - l;
      }
      uint32_t ka = (o_lo < n && A0 + ia < A1) ? A[A0 + ia] : 0xFFFFFFFFu;
      uint32_t kb = (o_lo < n && A1 + ib < A2) ? A[A1 + ib] : 0xFFFFFFFFu;
#pragma unroll
      for (int e = 0; e < E + 1; e++) {
        const uint32_t o = o_lo + e;
        out[e] = 0;
        if ((uint32_t)e < per && o < n) {
          while (o >= A2) {                                              // next pair (empty ones are skipped)
            p++;
            A0 = off[min(p * w2, s)]; A1 = off[min(p * w2 + (w2 >> 1), s)]; A2 = off[min((p + 1) * w2, s)];
            ia = 0; ib = 0;
            ka = A0 < A1 ? A[A0] : 0xFFFFFFFFu;
            kb = A1 < A2 ? A[A1] : 0xFFFFFFFFu;
          }
          const bool take_a = ka < kb;                                   // an exhausted run reads as +inf
          out[e] = take_a ? ka : kb;
          ia += take_a ? 1u : 0u; ib += take_a ? 0u : 1u;
          const uint32_t nxt = take_a ? A0 + ia : A1 + ib, end = take_a ? A1 : A2;
          const uint32_t v = nxt < end ? A[nxt] : 0xFFFFFFFFu;
          ka = take_a ? v : ka; kb = take_a ? kb : v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E + 1; e++) {
        const uint32_t o = o_lo + e;
        if ((uint32_t)e < per && o < n) A[o] = out[e];
      }
      __syncthreads();
    }
    }
    phase(2);
    seeds = A;
  } else {
    // ---- more seed hits than LDS holds: lists gathered into HBM scratch and sorted there ----
    seeds = a.ovf_buf + a.ovf_off[f];
    if (tid == 0) { sh_run = 0; atomicAdd(&a.counters[1], 1u); }         // (off the fast form: the HBM road)
    __syncthreads();
    for (int j0 = 0; j0 < s; j0 += blockDim.x) {
      int j = j0 + tid;
      uint32_t cnt = j < s ? a.q_cnt[(size_t)f * a.qcap + j] : 0;
      uint32_t incl = cnt;
      for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) sh_scan[wv] = incl;
      __syncthreads();
      uint32_t off = sh_run + incl - cnt;
      for (int q = 0; q < wv; q++) off += sh_scan[q];
      if (cnt) {
        uint32_t src = a.q_off[(size_t)f * a.qcap + j];
        for (uint32_t t = 0; t < cnt; t++) seeds[off + t] = a.ix.pos_ridx[src + t];
      }
      __syncthreads();
      if (tid == 0) { uint32_t tot = 0; for (int q = 0; q < NT / 64; q++) tot += sh_scan[q]; sh_run += tot; }
      __syncthreads();
    }
    for (uint32_t i = n + tid; i < n32; i += blockDim.x) seeds[i] = SEED_PAD;
    __syncthreads();
    // a record index identifies (seqId, wpos) and the record array is ordered by it: sorting indices == std::sort of hits
    block_bitonic_sort(seeds, n32);
  }

  int m = a.min_hits_lut[s];
  if (m < 1) m = 1;
  if ((uint32_t)m > n) return;
  const uint32_t ncand = n - (uint32_t)m + 1;
  const int len = a.frag_len;

  // ---- ordered passes over the candidates.  Pass 0 merges them into loci held in LDS (up to L1_STAGE per fragment) and
  //      is normally the only one; if a fragment has more loci, pass 0 only counted and pass 1 writes them to HBM.
  //      A seed is handled as its padded global coordinate G (k_rec_gpos): "same contig and wb - wa < len" is Gb - Ga < len,
  //      "the previous candidate's end reaches this one's start" is Gb - Gprev < len (the previous flagged seed precedes
  //      this one, so it shares the contig of the partner iff it shares this seed's).  A locus is carried as three record
  //      numbers -- its first seed, that seed's partner, its last flagged seed -- and turned into (contig, start, end) once,
  //      at the end: 4 bytes gathered per hit instead of 8, three gathers per locus. ----
  int32_t *st_seq = (int32_t *)(lds + l1_stage_offset(a.lds_seed_cap, a.lut_smax, NT));   // [L1_STAGE] each
  int32_t *st_start = st_seq + L1_STAGE, *st_rfirst = st_start + L1_STAGE, *st_end = st_rfirst + L1_STAGE, *st_rlast = st_end + L1_STAGE;
  int32_t *st_rpart = st_rlast + L1_STAGE;
  for (int i = tid; i < L1_STAGE; i += NT) st_rlast[i] = 0;
  const uint64_t len64 = (uint64_t)len;
  // the loci of a fragment that fit the LDS stage (the common case): (contig, start, end) from their three records, groups,
  // ONE reservation, out.  Ends the workgroup's work.
  auto staged_epilogue = [&](const uint32_t cnt0) __attribute__((always_inline)) {
    // The common case: the loci are in LDS.  A group (consecutive loci on the same reference genome) is numbered by the
    // first locus it holds, so loci and groups need ONE reservation -- thousands of workgroups queue up on that address for
    // ~12 ns each (on a single query all 1 666 arrive within one resident round: 20 us of queue.  Issuing the atomic as soon
    // as the candidate scan knows the count does not overlap that queue with the rest of the work: the compiler's atomic
    // optimizer waits for the answer on the spot).
    uint32_t *st_grp = (uint32_t *)(lds + l1_off_offset(a.lds_seed_cap));   // (the list offsets are no longer needed)
    // (contig, start, end) of every locus from its three records
    for (uint32_t q = tid; q < cnt0; q += NT) {
      st_seq[q] = a.ix.rec_seq[st_rfirst[q]];
      st_start[q] = max(0, a.ix.rec_wpos[st_rpart[q]] - len + 1);
      st_end[q] = a.ix.rec_wpos[st_rlast[q]];
    }
    __syncthreads();
    if (wv == 0) {
      uint32_t last_head = 0;
      for (uint32_t i0 = 0; i0 < cnt0; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool gh = false;
        if (i < cnt0) gh = (i == 0) || a.ix.contig_genome[st_seq[i]] != a.ix.contig_genome[st_seq[i - 1]];
        const uint64_t gb = __ballot(gh);
        const uint64_t upto = gb & ((2ULL << lane) - 1ULL);
        if (i < cnt0) st_grp[i] = upto ? i0 + (uint32_t)(63 - __clzll(upto)) : last_head;   // first locus of the group of locus i
        if (gb) last_head = i0 + (uint32_t)(63 - __clzll(gb));
      }
      if (lane == 0) {
        bool fits;
        uint32_t base = reserve_loci(a.loci, f, cnt0, fits), cnt = cnt0;
        if (!fits) { atomicExch(&a.counters[2], 1u); atomicOr(&a.pinfo[1], (unsigned long long)SPEC_LOCI); cnt = 0; }
        sh_base = base;
        sh_gbase = cnt;
        a.f_loci_lo[f] = base; a.f_loci_n[f] = cnt;
      }
    }
    __syncthreads();
    if (sh_gbase == 0) return;
    for (uint32_t q = tid; q < sh_gbase; q += NT) {
      const uint32_t li = sh_base + q;
      a.l_frag[li] = f; a.l_seq[li] = st_seq[q]; a.l_start[li] = st_start[q]; a.l_rfirst[li] = st_rfirst[q];
      a.l_end[li] = st_end[q]; a.l_rlast[li] = st_rlast[q]; a.l_rpart[li] = st_rpart[q];
      a.l_group[li] = (int32_t)(sh_base + st_grp[q]);
    }
    phase(4);
  };
  if (in_lds) {
    // ---- Candidate pass for hits sorted in LDS: every wave takes a CONTIGUOUS run of 64-candidate steps and carries the
    //      state of the scan (last flagged seed, heads so far) in registers from step to step -- no barrier and no LDS
    //      round trip between steps.  The coordinates of a batch of steps are fetched at once (one memory round trip per batch
    //      instead of one per step: the kernel is a chain of latencies, not of instructions); the partner seed i+m-1 and
    //      the previous flagged seed are other lanes of this step or the next, read by lane exchange.  The first flagged
    //      candidate of a wave's run is a head *provisionally*: after ONE barrier every wave reads the other waves'
    //      summaries, learns whether that candidate continues the previous wave's last locus and where its own loci
    //      start, and a second sweep over the saved ballots (no coordinates needed) writes the loci. ----
    constexpr int NW = NT / 64;
    __shared__ uint64_t cw_last_g[NW], cw_first_gb[NW];
    __shared__ uint32_t cw_heads[NW], cw_any[NW];
    uint4 *step_bits = (uint4 *)(lds + l1_off_offset(a.lds_seed_cap));    // [steps] {flag ballot, head ballot} (the list offsets are no longer needed)
    const uint32_t S = (ncand + 63u) / 64u, T = (S + NW - 1) / NW;
    const uint32_t s0 = min(S, (uint32_t)wv * T), s1 = min(S, s0 + T);
    const int mp = m - 1;                                                  // distance to the partner seed
    uint64_t carry_g = 0, first_gb = 0;
    bool has_carry = false, any = false;
    uint32_t heads = 0;
    // (the scan proper, on 32-bit coordinates while the index spans less than 2^32 padded bases -- twice the steps per batch
    // in the same registers, half the lane exchanges -- and on 64-bit ones beyond)
    auto scan_run = [&](auto tag, auto near_tag) __attribute__((always_inline)) {
      using G = decltype(tag);
      constexpr bool NARROW = sizeof(G) == 4;
      constexpr bool NEAR = decltype(near_tag)::value;
      // steps per batch (registers: one or two per step; the partner read of the near test takes one step's worth)
      constexpr int CB = NEAR ? (NARROW ? 8 : 6) : (NARROW ? 9 : 6);
      // The coordinate of a hit is only fetched if the hit can be one end of a candidate (round 5).  Records are ordered by
      // (contig, window) and window positions grow by at least one per record, so the padded global coordinates of two hits
      // differ by at least their distance in RECORDS: hit i can START a candidate (`fwd`) only if its partner, m - 1 hits ahead,
      // is less than a fragment length of records away, and its coordinate is needed only then or if the hit m - 1 places
      // BACK can start one -- a bit of the previous lanes' ballot, of this step or the one before (scalar shifts: no third LDS
      // read, no further live register in a kernel that has none to spare).  Every other hit reads record 0 (one cached
      // line): that is what the ~500 chance hits a fragment collects in a 4 x 10^8-record index are, and each had cost a
      // 64-128 byte line of rec_gpos for 4 bytes nobody used (k_l1's fetch on 500 x 500 genomes: 110 -> 72 GB).  A candidate
      // is only tested where `fwd` holds; both coordinates are real then.  FA_L1_NEAR=0: every hit counts as a possible start.
      auto fwd_of = [&](uint32_t idx, uint32_t r) __attribute__((always_inline)) {
        if constexpr (!NEAR) return true;
        else return idx + (uint32_t)mp < n && seeds[idx + (uint32_t)mp] - r < (uint32_t)len;
      };
      uint64_t prev_fwdb = 0;
      if (NEAR && s0 > 0 && s0 < s1) {
        const uint32_t idx = (s0 - 1u) * 64u + (uint32_t)lane;
        prev_fwdb = __ballot(idx < n && fwd_of(idx, seeds[min(idx, n - 1u)]));
      }
      auto coord = [&](uint32_t step, bool &fwd) __attribute__((always_inline)) -> G {
        const uint32_t idx = step * 64u + (uint32_t)lane;
        if constexpr (!NEAR) {                                             // (the round-4 form, for the A/B: every coordinate is fetched)
          fwd = true;
          if (!(step <= s1 && idx < n)) return (G)0;
          if constexpr (NARROW) return (G)a.ix.rec_gpos[seeds[idx]]; else return (G)gpos_of(a.ix, seeds[idx]);
        }
        const bool in = step <= s1 && idx < n;
        const uint32_t r = in ? seeds[idx] : 0u;
        fwd = in && fwd_of(idx, r);
        const uint64_t fwdb = __ballot(fwd);
        bool need;
        if (mp < 64) {
          const uint64_t bwdb = mp == 0 ? fwdb : ((fwdb << mp) | (prev_fwdb >> (64 - mp)));
          need = (((fwdb | bwdb) >> lane) & 1ULL) != 0ULL;
        } else {                                                           // (huge sketches only)
          need = fwd || (in && idx >= (uint32_t)mp && fwd_of(idx - (uint32_t)mp, seeds[idx - (uint32_t)mp]));
        }
        prev_fwdb = fwdb;
        if (!in) return (G)0;
        const uint32_t rr = need ? r : 0u;
        if constexpr (NARROW) return (G)a.ix.rec_gpos[rr]; else return (G)gpos_of(a.ix, rr);
      };
      auto from_lane = [&](G v, int l) __attribute__((always_inline)) -> G {       // v of lane l (any l)
        if constexpr (NARROW) return (G)__shfl((int)v, l); else return (G)__shfl((long long)v, l);
      };
      auto uniform_lane = [&](G v, int l) __attribute__((always_inline)) -> uint64_t {   // v of lane l, l uniform
        if constexpr (NARROW) return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, l);
        else return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
      };
      const G lenG = (G)len;
      G carry = 0;
      bool fwd_next = false;
      G g_next = s0 < s1 ? coord(s0, fwd_next) : (G)0;
      for (uint32_t sb = s0; sb < s1; sb += CB) {
        G g[CB + 1];
        uint32_t fwd_bits = fwd_next ? 1u : 0u;                            // bit u: the hit of g[u] may start a candidate
        g[0] = g_next;
#pragma unroll
        for (int u = 1; u <= CB; u++) { bool fw; g[u] = coord(sb + (uint32_t)u, fw); if constexpr (NEAR) fwd_bits |= fw ? (1u << u) : 0u; }
#pragma unroll
        for (int u = 0; u < CB; u++) {
          const uint32_t step = sb + (uint32_t)u;
          if (step < s1) {
            const uint32_t i = step * 64u + (uint32_t)lane;
            const G ga = g[u];
            G gb;
            if (mp < 64) {
              const G x = from_lane(g[u], (lane + mp) & 63), y = from_lane(g[u + 1], (lane + mp) & 63);
              gb = lane + mp < 64 ? x : y;
            } else {                                                       // (huge sketches only)
              gb = 0;
              if (i < ncand && (!NEAR || ((fwd_bits >> u) & 1u))) { if constexpr (NARROW) gb = (G)a.ix.rec_gpos[seeds[i + mp]]; else gb = (G)gpos_of(a.ix, seeds[i + mp]); }
            }
            const bool flag = i < ncand && (!NEAR || ((fwd_bits >> u) & 1u)) && (G)(gb - ga) < lenG;
            const uint64_t bal = __ballot(flag);
            const uint64_t below = bal & ((1ULL << lane) - 1ULL);
            const int src_lane = below ? 63 - __clzll(below) : -1;
            G gp = from_lane(ga, max(src_lane, 0));
            const bool has_prev = src_lane >= 0 || has_carry;
            if (src_lane < 0) gp = carry;
            const bool head = flag && !(has_prev && (G)(gb - gp) < lenG);
            const uint64_t hb = __ballot(head);
            if (lane == 0) step_bits[step] = make_uint4((uint32_t)bal, (uint32_t)(bal >> 32), (uint32_t)hb, (uint32_t)(hb >> 32));
            if (bal) {
              if (!any) { first_gb = uniform_lane(gb, __ffsll((long long)bal) - 1); any = true; }
              carry_g = uniform_lane(ga, 63 - __clzll(bal)); carry = (G)carry_g; has_carry = true;
            }
            heads += (uint32_t)__popcll(hb);
          }
        }
        g_next = g[CB];
        fwd_next = (fwd_bits >> CB) & 1u;
      }
    };
    if (a.block_sort & 4) { if (a.ix.n_wraps == 0) scan_run((uint32_t)0, std::true_type()); else scan_run((uint64_t)0, std::true_type()); }
    else { if (a.ix.n_wraps == 0) scan_run((uint32_t)0, std::false_type()); else scan_run((uint64_t)0, std::false_type()); }
    if (lane == 0) { cw_heads[wv] = heads; cw_any[wv] = any ? 1u : 0u; cw_last_g[wv] = carry_g; cw_first_gb[wv] = first_gb; }
    __syncthreads();
    // every wave resolves the chain of summaries for itself: loci before its run, and whether its first flagged candidate
    // continues the last locus of the waves before it
    uint32_t my_base = 0, my_cont = 0, total = 0;
    {
      bool hp = false;
      uint64_t pg = 0;
      for (int q = 0; q < NW; q++) {
        const uint32_t aq = cw_any[q];
        const uint32_t cont = (aq && hp && cw_first_gb[q] - pg < len64) ? 1u : 0u;
        if (q == wv) { my_base = total; my_cont = cont; }
        total += cw_heads[q] - cont;
        if (aq) { hp = true; pg = cw_last_g[q]; }
      }
    }
    phase(3);
    if (total == 0) return;
    const bool staged = total <= (uint32_t)L1_STAGE;
    if (!staged) {
      // more loci than the stage holds: reserve, and write them straight to HBM
      if (tid == 0) {
        uint32_t cnt = total;
        bool fits;
        uint32_t base = reserve_loci(a.loci, f, cnt, fits);
        if (!fits) { atomicExch(&a.counters[2], 1u); atomicOr(&a.pinfo[1], (unsigned long long)SPEC_LOCI); cnt = 0; }
        sh_base = base;
        sh_gbase = cnt;
        a.f_loci_lo[f] = base; a.f_loci_n[f] = cnt;
      }
      __syncthreads();
      if (sh_gbase == 0) return;
    }
    {
      uint32_t hcount = 0;
      for (uint32_t step = s0; step < s1; step++) {
        const uint4 sbits = step_bits[step];
        const uint64_t bal = ((uint64_t)sbits.y << 32) | sbits.x, hb = ((uint64_t)sbits.w << 32) | sbits.z;
        const uint32_t local = hcount + (uint32_t)__popcll(hb & ((2ULL << lane) - 1ULL));   // provisional heads of this run up to this lane
        if ((bal >> lane) & 1ULL) {
          const uint32_t i = step * 64u + (uint32_t)lane;
          const bool head = ((hb >> lane) & 1ULL) && !(my_cont && local == 1u);
          const uint32_t slot = my_base + local - my_cont;                  // 1-based locus number inside the fragment
          const uint64_t above = (lane == 63) ? 0ULL : (bal & ~((2ULL << lane) - 1ULL));
          const bool last_here = above == 0 || ((hb >> (__ffsll((long long)above) - 1)) & 1ULL);
          if (staged) {
            if (head) { st_rfirst[slot - 1] = (int32_t)seeds[i]; st_rpart[slot - 1] = (int32_t)seeds[i + mp]; }
            if (last_here) atomicMax(&st_rlast[slot - 1], (int32_t)seeds[i]);
          } else {
            const uint32_t li = sh_base + slot - 1;
            if (head) { a.l_frag[li] = f; a.l_rfirst[li] = (int32_t)seeds[i]; a.l_rpart[li] = (int32_t)seeds[i + mp]; }
            if (last_here) atomicMax(&a.l_rlast[li], (int32_t)seeds[i]);
          }
        }
        hcount += (uint32_t)__popcll(hb);
      }
    }
    __syncthreads();
    if (staged) { staged_epilogue(total); return; }
  } else {
  uint64_t *g_trip = (uint64_t *)(lds + l1_off_offset(a.lds_seed_cap));  // [NT] (the list offsets are no longer needed)
  for (int pass = 0; pass < 2; pass++) {
    if (tid == 0) { sh_heads[0] = 0; sh_has_prev[0] = 0; sh_prev_g[0] = 0; }
    __syncthreads();
    int par = 0;
    // coordinate of the seed this thread owns in the first trip; later trips are fetched one trip ahead
    uint32_t ra_n = (uint32_t)tid < n ? seeds[tid] : 0u;
    uint32_t lo_n = (uint32_t)tid < n ? a.ix.rec_gpos[ra_n] : 0u;
    for (uint32_t i0 = 0; i0 < ncand; i0 += NT) {
      uint32_t i = i0 + tid;
      // every lane fetches the coordinate of its own seed once and leaves it in LDS for its wave: the partner seed
      // i+m-1 and the previous flagged candidate are other lanes' seeds (an 8-byte LDS read each where a `__shfl` -- a
      // ds_bpermute, ~18 cycles of the CU's LDS pipe -- per word cost a fifth of this kernel)
      const uint32_t ra = ra_n;
      const uint64_t ga = gpos_make(a.ix, ra, lo_n);
      g_trip[tid] = ga;
      // (the slots written above are read below by OTHER lanes of the same wave only: order the store before those reads
      // explicitly -- no instruction on gfx950, where a wave's LDS operations complete in order -- instead of relying on it)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (i0 + NT < ncand) {
        const uint32_t in = i + NT;
        ra_n = in < n ? seeds[in] : 0u;
        lo_n = in < n ? a.ix.rec_gpos[ra_n] : 0u;
      }
      const int rb = i < ncand ? (int)seeds[i + m - 1] : 0;               // record of the partner seed: it fixes the locus start
      uint64_t gb = g_trip[min(tid + m - 1, NT - 1)];                      // (read by its own wave only: LDS keeps a wave's order)
      if (lane + m - 1 >= 64 && i < ncand) gb = gpos_of(a.ix, (uint32_t)rb);
      const bool flag = i < ncand && gb - ga < len64;
      // previous flagged candidate (in order): inside the wave, else earlier waves, else the carry
      uint64_t bal = __ballot(flag);
      __shared__ uint64_t w_last_g[2][NT / 64];
      __shared__ int w_any[2][NT / 64];
      uint64_t below = bal & ((1ULL << lane) - 1ULL);
      int src_lane = below ? 63 - __clzll(below) : -1;
      uint64_t gp = g_trip[(tid & ~63) + max(src_lane, 0)];
      if (lane == 0) w_any[par][wv] = bal != 0;
      if (bal && lane == 63 - __clzll(bal)) w_last_g[par][wv] = ga;
      __syncthreads();
      bool has_prev = src_lane >= 0;
      if (!has_prev) {
        for (int q = wv - 1; q >= 0 && !has_prev; q--) if (w_any[par][q]) { has_prev = true; gp = w_last_g[par][q]; }
        if (!has_prev && sh_has_prev[par]) { has_prev = true; gp = sh_prev_g[par]; }
      }
      bool head = flag && !(has_prev && gb - gp < len64);
      // inclusive scan of heads -> slot of the locus every flagged candidate belongs to
      uint64_t hb = __ballot(head);
      __shared__ uint32_t w_heads[NT / 64];
      if (lane == 0) w_heads[wv] = __popcll(hb);
      __syncthreads();
      uint32_t slot = sh_heads[par] + __popcll(hb & ((2ULL << lane) - 1ULL));
      for (int q = 0; q < wv; q++) slot += w_heads[q];
      if (flag) {
        // the end of a locus is its last flagged seed; only the last flagged lane of a locus inside this wave
        // touches memory (same-address atomics from every lane would serialise)
        const uint64_t above = (lane == 63) ? 0ULL : (bal & ~((2ULL << lane) - 1ULL));
        const bool last_here = above == 0 || ((hb >> (__ffsll((long long)above) - 1)) & 1ULL);
        if (pass == 0) {
          if (slot <= (uint32_t)L1_STAGE) {
            if (head) { st_rfirst[slot - 1] = (int32_t)ra; st_rpart[slot - 1] = rb; }
            if (last_here) atomicMax(&st_rlast[slot - 1], (int32_t)ra);
          }
        } else {
          uint32_t li = sh_base + slot - 1;
          if (head) { a.l_frag[li] = f; a.l_rfirst[li] = (int32_t)ra; a.l_rpart[li] = rb; }
          if (last_here) atomicMax(&a.l_rlast[li], (int32_t)ra);
        }
      }
      if (tid == 0) {
        // next trip's carry goes to the other parity: nobody reads it before the next trip's first barrier
        uint32_t tot = sh_heads[par];
        for (int q = 0; q < NT / 64; q++) tot += w_heads[q];
        sh_heads[par ^ 1] = tot;
        int hp = sh_has_prev[par];
        uint64_t pg = sh_prev_g[par];
        for (int q = NT / 64 - 1; q >= 0; q--) if (w_any[par][q]) { hp = 1; pg = w_last_g[par][q]; break; }
        sh_has_prev[par ^ 1] = hp; sh_prev_g[par ^ 1] = pg;
      }
      par ^= 1;
    }
    __syncthreads();
    phase(3);
    if (pass == 0) {
      const uint32_t cnt0 = sh_heads[par];
      if (cnt0 > 0 && cnt0 <= (uint32_t)L1_STAGE) {
        staged_epilogue(cnt0);
        return;
      }
      // more loci than the stage holds (or none): reserve, then a second pass writes them straight to HBM
      if (tid == 0) {
        uint32_t cnt = cnt0;
        bool fits = true;
        uint32_t base = cnt ? reserve_loci(a.loci, f, cnt, fits) : 0;
        if (!fits) { atomicExch(&a.counters[2], 1u); atomicOr(&a.pinfo[1], (unsigned long long)SPEC_LOCI); cnt = 0; }
        sh_base = base;
        sh_gbase = cnt;   // reuse: number of loci (0 => skip)
        a.f_loci_lo[f] = base; a.f_loci_n[f] = cnt;
      }
      __syncthreads();
      if (sh_gbase == 0) return;
    }
  }
  }
  // ---- (contig, start, end) of the loci written by pass 1, then the groups: consecutive loci of this fragment on the same
  //      reference genome ----
  __threadfence_block();
  __syncthreads();
  const uint32_t nl = sh_gbase, base = sh_base;
  for (uint32_t q = tid; q < nl; q += NT) {
    const uint32_t li = base + q;
    a.l_seq[li] = a.ix.rec_seq[a.l_rfirst[li]];
    a.l_start[li] = max(0, a.ix.rec_wpos[a.l_rpart[li]] - len + 1);
    a.l_end[li] = a.ix.rec_wpos[a.l_rlast[li]];
  }
  __threadfence_block();
  __syncthreads();
  if (wv == 0) {
    // a group is numbered by the first locus it holds (see staged_epilogue): no counter of its own
    uint32_t last_head = 0;
    for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
      uint32_t i = i0 + lane;
      bool gh = false;
      if (i < nl) {
        int g = a.ix.contig_genome[a.l_seq[base + i]];
        gh = (i == 0) || g != a.ix.contig_genome[a.l_seq[base + i - 1]];
      }
      const uint64_t gb = __ballot(gh);
      const uint64_t upto = gb & ((2ULL << lane) - 1ULL);
      if (i < nl) a.l_group[base + i] = (int32_t)(base + (upto ? i0 + (uint32_t)(63 - __clzll(upto)) : last_head));
      if (gb) last_head = i0 + (uint32_t)(63 - __clzll(gb));
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// L1 for fragments with more seed hits than LDS holds (an index with hundreds of strains of the query's species).
// The hits are cut at CONTIG boundaries -- no candidate spans two contigs -- into chunks that fit LDS: every list
// proposes the record `share` entries ahead of its cursor, the smallest proposal, rounded down to the start of its
// contig, bounds the chunk (no list then contributes more than `share` = cap / s hits).  Every chunk goes through the
// same gather, merge-path merge and candidate pass as in k_l1; the loci are collected in the fragment's HBM scratch
// and moved to their final place at the end.  A fragment that cannot be cut (one contig alone holds more than a fair
// share of some list) or whose loci outgrow the scratch is left to k_l1's HBM path (big_state = 0).
// ----------------------------------------------------------------------------------------------------------
constexpr int L1_BIG_THREADS = 512, L1_BIG_E = 32;
__host__ __device__ inline size_t l1_big_lds_bytes(uint32_t seed_cap, int lut_smax) {
  return ((size_t)seed_cap * 4 + 15) / 16 * 16 + ((size_t)lut_smax + 2) * 16;   // seeds, then off / qo / cur / nxt
}

__global__ __launch_bounds__(L1_BIG_THREADS) void k_l1_big(L1Args a) {
  constexpr int NT = L1_BIG_THREADS, E = L1_BIG_E;
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ uint32_t sh_scan[NT / 64], sh_min[NT / 64], sh_first[NT / 64];
  __shared__ uint32_t sh_run, sh_loci, sh_firstrem, sh_bound, sh_base, sh_cnt;
  __shared__ int sh_has_prev, sh_fail, sh_last;
  __shared__ uint64_t sh_prev_g;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int s = a.q_size[f];
  const uint32_t n = a.n_seeds[f];
  const uint32_t cap = a.big_cap;
  if (tid == 0) a.big_state[f] = 0;
  if (s == 0 || n <= a.lds_seed_cap || s > a.lut_smax) return;
  uint32_t n32 = 1; while (n32 < n) n32 <<= 1;
  if ((uint64_t)a.ovf_off[f] + n32 > a.scratch_words) return;           // SPEC_SCRATCH: the pass is void anyway
  uint32_t *A = (uint32_t *)lds;
  uint32_t *off = (uint32_t *)(lds + ((size_t)cap * 4 + 15) / 16 * 16);  // [s + 1] first hit of every list inside the chunk
  uint32_t *qo = off + a.lut_smax + 2;                                   // [s] where the chunk's part of every list starts in the index
  uint32_t *cur = qo + a.lut_smax + 2, *nxt = cur + a.lut_smax + 2;      // [s] cursor of every list, and its value after the chunk
  // loci collected in the fragment's scratch as three record numbers each (first seed, its partner, last flagged seed: see
  // k_l1), arrays of LC entries
  const uint32_t LC = n32 / 5;
  int32_t *S = (int32_t *)(a.ovf_buf + a.ovf_off[f]);
  int32_t *S_rfirst = S, *S_rpart = S + LC, *S_rlast = S + 2 * (size_t)LC;
  for (uint32_t i = tid; i < LC; i += NT) S_rlast[i] = 0;
  for (int j = tid; j < s; j += NT) cur[j] = 0;
  if (tid == 0) { sh_loci = 0; sh_fail = 0; }
  int m = a.min_hits_lut[s];
  if (m < 1) m = 1;
  const int len = a.frag_len;
  const uint32_t share = max(1u, cap / (uint32_t)s);
  const uint32_t *qcnt = a.q_cnt + (size_t)f * a.qcap, *qoff = a.q_off + (size_t)f * a.qcap;
  __threadfence_block();
  __syncthreads();
  for (;;) {
    // ---- bound of the chunk: smallest proposal, rounded down to a contig start ----
    uint32_t prop = 0xFFFFFFFFu, first = 0xFFFFFFFFu;
    for (int j = tid; j < s; j += NT) {
      const uint32_t c = qcnt[j], cj = cur[j];
      if (cj < c) first = min(first, a.ix.pos_ridx[qoff[j] + cj]);
      if (c - cj > share) prop = min(prop, a.ix.pos_ridx[qoff[j] + cj + share]);
    }
    for (int d = 32; d > 0; d >>= 1) { prop = min(prop, (uint32_t)__shfl_xor((int)prop, d)); first = min(first, (uint32_t)__shfl_xor((int)first, d)); }
    if (lane == 0) { sh_min[wv] = prop; sh_first[wv] = first; }
    __syncthreads();
    if (tid == 0) {
      uint32_t v = 0xFFFFFFFFu, fr = 0xFFFFFFFFu;
      for (int q = 0; q < NT / 64; q++) { v = min(v, sh_min[q]); fr = min(fr, sh_first[q]); }
      uint32_t bound = 0xFFFFFFFFu;                                     // everything that is left
      if (v != 0xFFFFFFFFu) {
        bound = (uint32_t)a.ix.contig_rec[a.ix.rec_seq[v]];             // first record of the contig that holds v
        if (bound <= fr) sh_fail = 1;                                    // that contig alone is too much: cannot cut here
      }
      sh_bound = bound; sh_last = v == 0xFFFFFFFFu; sh_firstrem = fr;
    }
    __syncthreads();
    if (sh_fail || sh_firstrem == 0xFFFFFFFFu) break;                    // cannot cut / nothing left
    const uint32_t bound = sh_bound;
    // ---- the chunk's part of every list: [cur, nxt) with nxt = first entry >= bound ----
    if (tid == 0) sh_run = 0;
    __syncthreads();
    for (int j0 = 0; j0 < s; j0 += NT) {
      const int j = j0 + tid;
      uint32_t cnt = 0;
      if (j < s) {
        const uint32_t c = qcnt[j], cj = cur[j];
        uint32_t lo = cj, hi = bound == 0xFFFFFFFFu ? c : min(c, cj + share + 1);
        if (bound != 0xFFFFFFFFu) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.ix.pos_ridx[qoff[j] + mid] < bound) lo = mid + 1; else hi = mid; } }
        else lo = c;
        nxt[j] = lo; cnt = lo - cj;
        qo[j] = qoff[j] + cj;
      }
      uint32_t incl = cnt;
      for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) sh_scan[wv] = incl;
      __syncthreads();
      uint32_t o = sh_run + incl - cnt;
      for (int q = 0; q < wv; q++) o += sh_scan[q];
      if (j < s) off[j] = o;
      __syncthreads();
      if (tid == 0) { uint32_t tot = 0; for (int q = 0; q < NT / 64; q++) tot += sh_scan[q]; sh_run += tot; }
      __syncthreads();
    }
    const uint32_t nc = sh_run;                                          // hits of the chunk (<= cap by construction)
    if (tid == 0) off[s] = nc;
    __syncthreads();
    if (nc > cap) { if (tid == 0) sh_fail = 1; __syncthreads(); break; }
    // ---- gather + merge-path merge in LDS (as in k_l1) ----
    auto locate = [&](uint32_t i) __attribute__((always_inline)) {
      int lo = 0, hi = s - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (off[mid + 1] <= i) lo = mid + 1; else hi = mid; }
      return lo;
    };
    for (uint32_t i = tid; i < nc; i += NT) { const int j = locate(i); A[i] = a.ix.pos_ridx[qo[j] + (i - off[j])]; }
    __syncthreads();
    {
      const uint32_t per = ((nc + NT - 1) / NT) | 1u;                  // (odd, as in k_l1: no bank conflicts between the lanes' ranges)
      const uint32_t o_lo = (uint32_t)tid * per;
      for (int k = 0; (1 << k) < s; k++) {
        uint32_t out[E + 1];
        const int w2 = 2 << k;
        const int npairs = (s + w2 - 1) / w2;
        uint32_t A0 = 0, A1 = 0, A2 = 0, ia = 0, ib = 0;
        int p = 0;
        if (o_lo < nc) {
          int lo = 0, hi = npairs - 1;
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[min(mid * w2, s)] <= o_lo) lo = mid; else hi = mid - 1; }
          p = lo;
          A0 = off[min(p * w2, s)]; A1 = off[min(p * w2 + (w2 >> 1), s)]; A2 = off[min((p + 1) * w2, s)];
          const uint32_t diag = o_lo - A0, lenA = A1 - A0, lenB = A2 - A1;
          uint32_t l = diag > lenB ? diag - lenB : 0u, h = min(diag, lenA);
          while (l < h) { const uint32_t mid = (l + h) >> 1; if (A[A0 + mid] < A[A1 + diag - 1 - mid]) l = mid + 1; else h = mid; }
          ia = l; ib = diag - l;
        }
        uint32_t ka = (o_lo < nc && A0 + ia < A1) ? A[A0 + ia] : 0xFFFFFFFFu;
        uint32_t kb = (o_lo < nc && A1 + ib < A2) ? A[A1 + ib] : 0xFFFFFFFFu;
#pragma unroll
        for (int e = 0; e < E + 1; e++) {
          const uint32_t o = o_lo + e;
          out[e] = 0;
          if ((uint32_t)e < per && o < nc) {
            while (o >= A2) {
              p++;
              A0 = off[min(p * w2, s)]; A1 = off[min(p * w2 + (w2 >> 1), s)]; A2 = off[min((p + 1) * w2, s)];
              ia = 0; ib = 0;
              ka = A0 < A1 ? A[A0] : 0xFFFFFFFFu;
              kb = A1 < A2 ? A[A1] : 0xFFFFFFFFu;
            }
            const bool take_a = ka < kb;
            out[e] = take_a ? ka : kb;
            ia += take_a ? 1u : 0u; ib += take_a ? 0u : 1u;
            const uint32_t nx = take_a ? A0 + ia : A1 + ib, end = take_a ? A1 : A2;
            const uint32_t v = nx < end ? A[nx] : 0xFFFFFFFFu;
            ka = take_a ? v : ka; kb = take_a ? kb : v;
          }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E + 1; e++) {
          const uint32_t o = o_lo + e;
          if ((uint32_t)e < per && o < nc) A[o] = out[e];
        }
        __syncthreads();
      }
    }
    // ---- candidates of the chunk (computeL1CandidateRegions on its sorted hits; nothing carries over a contig start) ----
    if ((uint32_t)m <= nc) {
      const uint32_t ncand = nc - (uint32_t)m + 1;
      if (tid == 0) { sh_run = 0; sh_has_prev = 0; sh_prev_g = 0; }
      __syncthreads();
      const uint32_t loci0 = sh_loci;
      const uint64_t len64 = (uint64_t)len;
      for (uint32_t i0 = 0; i0 < ncand; i0 += NT) {
        const uint32_t i = i0 + tid;
        uint32_t ra = 0, rb = 0;
        uint64_t ga = 0;
        if (i < nc) { ra = A[i]; ga = gpos_of(a.ix, ra); }
        uint64_t gb = (uint64_t)__shfl((long long)ga, (lane + m - 1) & 63);
        if (i < ncand) rb = A[i + m - 1];
        if (lane + m - 1 >= 64 && i < ncand) gb = gpos_of(a.ix, rb);
        const bool flag = i < ncand && gb - ga < len64;
        uint64_t bal = __ballot(flag);
        __shared__ uint64_t w_last_g[NT / 64];
        __shared__ int w_any[NT / 64];
        uint64_t below = bal & ((1ULL << lane) - 1ULL);
        int src_lane = below ? 63 - __clzll(below) : -1;
        uint64_t gp = (uint64_t)__shfl((long long)ga, src_lane < 0 ? 0 : src_lane);
        if (lane == 0) w_any[wv] = bal != 0;
        if (bal && lane == 63 - __clzll(bal)) w_last_g[wv] = ga;
        __syncthreads();
        bool has_prev = src_lane >= 0;
        if (!has_prev) {
          for (int q = wv - 1; q >= 0 && !has_prev; q--) if (w_any[q]) { has_prev = true; gp = w_last_g[q]; }
          if (!has_prev && sh_has_prev) { has_prev = true; gp = sh_prev_g; }
        }
        bool head = flag && !(has_prev && gb - gp < len64);
        uint64_t hb = __ballot(head);
        __shared__ uint32_t w_heads[NT / 64];
        if (lane == 0) w_heads[wv] = __popcll(hb);
        __syncthreads();
        uint32_t slot = sh_run + __popcll(hb & ((2ULL << lane) - 1ULL));
        for (int q = 0; q < wv; q++) slot += w_heads[q];
        if (flag) {
          const uint64_t above = (lane == 63) ? 0ULL : (bal & ~((2ULL << lane) - 1ULL));
          const bool last_here = above == 0 || ((hb >> (__ffsll((long long)above) - 1)) & 1ULL);
          const uint32_t li = loci0 + slot - 1;
          if (li < LC) {
            if (head) { S_rfirst[li] = (int32_t)ra; S_rpart[li] = (int32_t)rb; }
            if (last_here) atomicMax(&S_rlast[li], (int32_t)ra);
          } else sh_fail = 1;                                           // more loci than the scratch holds
        }
        __syncthreads();
        if (tid == 0) {
          uint32_t tot = 0;
          for (int q = 0; q < NT / 64; q++) tot += w_heads[q];
          sh_run += tot;
          for (int q = NT / 64 - 1; q >= 0; q--) if (w_any[q]) { sh_has_prev = 1; sh_prev_g = w_last_g[q]; break; }
        }
        __syncthreads();
      }
      if (tid == 0) sh_loci = loci0 + sh_run;
    }
    // ---- next chunk ----
    for (int j = tid; j < s; j += NT) cur[j] = nxt[j];
    __syncthreads();
    if (sh_fail || sh_last) break;
  }
  __syncthreads();
  if (sh_fail) return;                                                   // big_state stays 0: k_l1 takes the fragment
  // ---- the loci to their final place, then the groups (as in k_l1) ----
  if (tid == 0) {
    uint32_t cnt = sh_loci;
    bool fits = true;
    uint32_t base = cnt ? reserve_loci(a.loci, f, cnt, fits) : 0;
    if (!fits) { atomicExch(&a.counters[2], 1u); atomicOr(&a.pinfo[1], (unsigned long long)SPEC_LOCI); cnt = 0; }
    sh_base = base; sh_cnt = cnt;
    a.f_loci_lo[f] = base; a.f_loci_n[f] = cnt;
    a.big_state[f] = 1;
    atomicAdd(&a.counters[1], 1u);                                       // (off the fast form: cut into LDS-sized chunks here)
  }
  __threadfence_block();
  __syncthreads();
  const uint32_t nl = sh_cnt, base = sh_base;
  for (uint32_t q = tid; q < nl; q += NT) {
    const uint32_t li = base + q;
    const int32_t rf = S_rfirst[q], rp = S_rpart[q], rl = S_rlast[q];
    a.l_frag[li] = f; a.l_seq[li] = a.ix.rec_seq[rf]; a.l_start[li] = max(0, a.ix.rec_wpos[rp] - len + 1); a.l_rfirst[li] = rf; a.l_rpart[li] = rp;
    a.l_end[li] = a.ix.rec_wpos[rl]; a.l_rlast[li] = rl;
  }
  __threadfence_block();
  __syncthreads();
  if (wv == 0) {
    // a group is numbered by the first locus it holds (see staged_epilogue): no counter of its own
    uint32_t last_head = 0;
    for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
      uint32_t i = i0 + lane;
      bool gh = false;
      if (i < nl) {
        int g = a.ix.contig_genome[a.l_seq[base + i]];
        gh = (i == 0) || g != a.ix.contig_genome[a.l_seq[base + i - 1]];
      }
      const uint64_t gb = __ballot(gh);
      const uint64_t upto = gb & ((2ULL << lane) - 1ULL);
      if (i < nl) a.l_group[base + i] = (int32_t)(base + (upto ? i0 + (uint32_t)(63 - __clzll(upto)) : last_head));
      if (gb) last_head = i0 + (uint32_t)(63 - __clzll(gb));
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// L2: slide the fragment-long super-window over one candidate locus and keep the best winnowed-MinHash
// intersection (computeL2MappedRegions + SlideMapper + MIIteratorL2).  One lane per locus.
//
// The reference keeps an ordered map over Q u W with a pivot at the s-th smallest hash.  Here the same quantity
// lives in rank space: every reference hash is reduced (binary search in the sorted query sketch) to either the
// rank r of the query hash it matches, or -- for a hash absent from Q -- the number c of query hashes below it.
// With cnt[c] = distinct window-only hashes of insertion rank c and f(r) = r + sum_{c<=r} cnt[c], query rank r is
// among the s smallest of the union iff f(r) < s, so shared = #{matched r < r*}, r* = min{r : f(r) >= s}.
// r* moves by at most one per inserted / deleted hash, exactly like the reference's pivot.
// ----------------------------------------------------------------------------------------------------------
// The work is split into two launches so that nothing slow sits on the sequential chain:
//   k_l2_events one workgroup per fragment.  Prologue, one lane per locus: searchIndex(rangeStart) as a bounded binary
//               search, the other two searchIndex() calls as rec_fwd lookups, the event count; the workgroup reserves
//               its slice of the event buffer with one atomicAdd (the order of fragments is irrelevant).  Then one
//               wave per locus: every reference record is reduced to its rank in the query sketch (bucket table + 2-3
//               LDS probes, sketch staged once per fragment) and written as one or two *events* (admit / drop) at
//               its position in the time-ordered event stream, which is plain arithmetic on rec_bwd / rec_fwd;
//   k_l2_scan   one lane per locus: the sequential slide is a fixed-trip, branch-free loop over the event stream (one
//               16-byte load per 8 events), with the per-rank state in lane-interleaved LDS.
// `eval` marks the last event of a window position, i.e. the point where the reference compares
// sharedSketchElements; events that are no-ops by the duplicate-linking rule carry zero deltas.
struct L2Args {
  IndexView ix;
  const uint32_t *q_hash;
  const int32_t *q_size;
  const int32_t *l_frag, *l_seq, *l_start, *l_end, *l_group, *l_rfirst, *l_rlast, *l_rpart;
  int32_t frag_len;
  int32_t *l_beg, *l_end0, *l_last;  // record range of the locus, end of the first super-window
  int32_t *l_ndrop;                  // records dropped before the slide ends
  uint32_t *l_nev;                   // [loci] events of the locus rounded up to a multiple of 8
  uint32_t *l_ioff;                  // [loci] first event of the locus in `items`
  unsigned long long *pinfo;         // [0] events reserved so far, [1] speculation flags
  uint64_t items_cap;                // capacity of `items` in events
  int32_t l_cap;                     // capacity of the loci arrays
  void *items;                       // uint16 or uint32 per event
  int32_t *l_shared, *l_pos;
  const int32_t *pass_lut;           // [smax+1]
  unsigned long long *group_best;    // [groups] (shared<<32 | ~locus)
  const uint32_t *counters;          // [2] loci overflow flag, [3] wide-state loci
  LociRegions loci;                  // which locus numbers are live (k_l2_scan)
  int32_t qcap, cmw;
  int32_t cnt_slots;                 // smax + 1
  int32_t lanes;                     // loci per workgroup of k_l2_scan (power of two <= 64)
  int32_t ev_stage;                  // events of one locus staged in LDS per wave of k_l2_events
  unsigned long long *rec_total;     // sum over loci of the records in their range (for the roofline line)
  // k_l2_events reserves the events of a fragment with ONE atomic that returns its offset; thousands of workgroups doing
  // that on one address within a few microseconds queue up for ~12 ns each (14 us of a 190 us kernel), so the arena is
  // cut into up to EV_REGIONS equal regions with a counter each and a fragment uses region (fragment mod n_regions)
  unsigned long long *ev_region, *rec_region;   // [EV_REGIONS] events reserved / records read per region
  uint64_t region_cap;               // events per region (items_cap / n_regions, a multiple of 8)
  uint32_t n_regions;                // regions in use: a power of two <= EV_REGIONS, about one per sixteen fragments
  uint8_t *l_redo;                   // [loci] set by the uint8-state scan when a count overflowed
  uint32_t *redo_count;              // number of loci sent to the uint16 pass
  const uint32_t *f_loci_lo, *f_loci_n;   // [F] loci of each fragment
  // k_l2_events, passes of several query genomes: workgroup b takes fragment frag_order[b] (-1: none).  The order puts the
  // fragments that sit at the same offset of their genomes next to each other AND on one XCD (workgroups go to XCDs round
  // robin): related genomes are largely co-linear, so those workgroups stream the same stretches of the index at the same
  // time and all but the first find them in the XCD's L2 / the memory-side cache instead of HBM.  nullptr: identity.
  const int32_t *frag_order;
  unsigned long long *stamp;         // stage_stamp: start of the L2 stage
  // k_l2_scan takes its loci through `scan_order` when it is set (round 6): the loci of every region sorted by the length of their
  // event streams, longest first (a counting sort over SCAN_CLASSES classes of `scan_class_div` events: k_l2_events counts,
  // k_l2_order places).  A wave slides 64 loci and lasts as long as the longest; in the numbering of k_l1 a wave holds the loci of
  // ONE fragment -- one per related genome, streams of every length the index's divergences produce (lane utilisation 85 %).
  uint32_t *scan_hist, *scan_cursor; // [loci.n][SCAN_CLASSES] loci per class / placed so far
  uint32_t *scan_order;              // [loci capacity] locus numbers, region by region (nullptr: the identity)
  int32_t scan_class_div;            // events per class; 0 = no ordering
#ifdef FA_EXPERIMENTS
  int32_t dbg;                       // FA_FUSED_DEBUG (timing experiments only, results are void): 1 = slider idles,
                                     // 2 = producer composes no events, 4 = producer issues no loads
#endif
};

constexpr int L2_THREADS = 64;
constexpr int EV_THREADS = 256;
constexpr int SCAN_CLASSES = 32;
__device__ __forceinline__ uint32_t scan_class(uint32_t nev, int32_t div) {   // 0 = the longest streams
  return (uint32_t)(SCAN_CLASSES - 1) - min((uint32_t)(SCAN_CLASSES - 1), nev / (uint32_t)div);
}
constexpr int EV_REGIONS = 64;

// Event word, from the low end: dM:2 | dW:2 | spare | drop | slot (= query rank + 1; 0 is the padding no-op) | no-eval
// (the top bit), where dM / dW are two's complement -1 / 0 / +1: the change of the matched bit of that rank, resp. of
// its count of window-only hashes.  A record whose admit / drop is a no-op by the duplicate-linking rule carries
// dM = dW = 0 but keeps drop / no-eval.  The layout is chosen for the slide (k_l2_scan): the slot field sits at bit 6,
// so with 64 one-byte lanes per slot row the masked event IS the row's LDS offset, and a set top bit turns the shared
// count of an event that carries no comparison into a negative number instead of costing a select.
template <typename T> struct EvBits;
template <> struct EvBits<uint16_t> { static constexpr int RANK = 9; };    // sketches up to 510 minimizers
template <> struct EvBits<uint32_t> { static constexpr int RANK = 24; };
constexpr int EV_DM = 0, EV_DW = 2, EV_DROP = 5, EV_SLOT = 6;
template <typename T> constexpr uint32_t ev_noeval() { return 1u << (8 * sizeof(T) - 1); }
template <typename T> constexpr uint32_t ev_slot_mask() { return ((1u << EvBits<T>::RANK) - 1u) << EV_SLOT; }
static_assert(EV_SLOT + EvBits<uint16_t>::RANK == 15 && EV_SLOT + EvBits<uint32_t>::RANK <= 31, "event fields overlap");

// Bucket table resolution of k_l2_events: 4096 buckets over the hash range of ~250 sketch entries leave at most two
// entries in all but a few buckets, so a rank lookup is one table read and one two-entry probe.
#ifndef FA_EV_QT_BITS
#define FA_EV_QT_BITS 12
#endif
#ifndef FA_EV_WAVES
#define FA_EV_WAVES 8
#endif
constexpr int EV_QT_BITS = FA_EV_QT_BITS;
constexpr int EV_WAVES_PER_SIMD = FA_EV_WAVES;
constexpr int EV_PROBE = 4;                // sketch entries compared at once per rank lookup
#ifndef FA_EV_RPL
#define FA_EV_RPL 4
#endif
constexpr int EV_RPL = FA_EV_RPL;          // records per lane and trip of k_l2_events
static_assert(EV_RPL >= 1 && EV_RPL <= 4, "the last trip of a phase is dispatched on 1..4 records per lane");
__host__ __device__ inline size_t ev_sketch_bytes(int cnt_slots) { return ((size_t)(cnt_slots - 1 + EV_PROBE) * 4 + 15) / 16 * 16; }   // + sentinels

// RK = the rank structure (round 6).  0: the bucket table above + EV_PROBE sketch entries per look-up (rounds 2-5: one table read,
// four sketch reads, four compare-and-adds, a read of the entry at the final rank: seven LDS instructions with the store).
// 1: an OCCUPANCY word per bucket -- 2^EV_OCC_BITS buckets of 32 sub-buckets over the same hash range, i.e. 8 times the resolution
// in three quarters of the LDS: OCC[b] bit j = some sketch hash falls into sub-bucket j of bucket b, R0[b] = sketch entries below
// bucket b.  The rank of a reference hash is R0[b] + popcount(OCC[b] below its sub-bucket) -- a shift and ONE v_bcnt, which adds
// R0 in the same instruction --, exact unless a sketch hash shares its sub-bucket: the TWO sketch entries from that rank on (one
// ds_read2) settle that case and the membership test in the same step; only a hash with two sketch entries of its own sub-bucket
// below it walks on.  Four LDS instructions per record instead of seven.  (64-bit words, 2^16 sub-buckets: the same time, more
// vector instructions -- a 64-bit shift and two counts; profiles/EXPERIMENTS.md, round 6.)
constexpr int EV_OCC_BITS = 10, EV_SUB_BITS = 5;
template <typename T, bool PACKED, int RK>
__global__ __launch_bounds__(EV_THREADS, EV_WAVES_PER_SIMD) void k_l2_events(L2Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  stage_stamp(a.stamp);
  uint32_t *Q = (uint32_t *)lds;                                     // [s + EV_PROBE], staged once per fragment, with sentinels
  constexpr int QT_BITS = RK ? EV_OCC_BITS + EV_SUB_BITS : EV_QT_BITS;   // resolution of the table over the hash range, in bits
  constexpr int OCC_N = (1 << EV_OCC_BITS) + 1;                      // (+ the bucket behind the range: rank s, nothing occupied)
  constexpr uint32_t SUB_MASK = (1u << EV_SUB_BITS) - 1u;
  static_assert(EV_SUB_BITS == 5, "one 32-bit occupancy word per bucket");
  __shared__ __align__(8) unsigned char rank_table[RK ? OCC_N * 4 + (OCC_N + 1) * 2 : ((1 << EV_QT_BITS) + 2) * 2];
  uint16_t *const QT = (uint16_t *)rank_table;                       // RK == 0
  uint32_t *const OCC = (uint32_t *)rank_table;                      // RK == 1
  uint16_t *const R0 = (uint16_t *)(rank_table + OCC_N * 4);
  const int f = a.frag_order ? a.frag_order[blockIdx.x] : (int)blockIdx.x;
  if (f < 0) return;
  const uint32_t l_lo = a.f_loci_lo[f], l_n = a.f_loci_n[f];
  if (l_n == 0) return;
  const int s = a.q_size[f];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (a.counters[2] || s > a.cnt_slots - 1) return;                  // loci overflowed / sketch larger than speculated: void pass
  for (int i = threadIdx.x; i < s + EV_PROBE; i += EV_THREADS) Q[i] = i < s ? a.q_hash[(size_t)f * a.qcap + i] : 0xFFFFFFFFu;
  if constexpr (RK) { for (int b = threadIdx.x; b < OCC_N; b += EV_THREADS) { OCC[b] = 0u; R0[b] = (uint16_t)s; } }
  else { for (int b = threadIdx.x; b <= (1 << QT_BITS) + 1; b += EV_THREADS) QT[b] = (uint16_t)s; }
  // ---- record range of every locus (the three searchIndex calls of computeL2MappedRegions) and its event count ----
  __shared__ uint32_t sh_wave[EV_THREADS / 64];
  __shared__ uint32_t sh_run, sh_base, sh_ok;
  __shared__ unsigned long long sh_records;
  __shared__ uint32_t sh_class[SCAN_CLASSES];                          // loci of this fragment per length class (scan_order)
  if (threadIdx.x == 0) { sh_run = 0; sh_records = 0; }
  if (threadIdx.x < SCAN_CLASSES) sh_class[threadIdx.x] = 0;
  __syncthreads();
  const int32_t *wpos = a.ix.rec_wpos;
  for (uint32_t c0 = 0; c0 < l_n; c0 += EV_THREADS) {
    const uint32_t l = l_lo + c0 + threadIdx.x;
    uint32_t nev = 0, records = 0;
    if (c0 + threadIdx.x < l_n) {
      const int lo = a.ix.contig_rec[a.l_seq[l]];
      // searchIndex(seqId, rangeStartPos): rangeStartPos <= wpos of the first seed and wpos is strictly increasing, so
      // the answer lies within fragment_length records before that seed
      const int rfirst = a.l_rfirst[l], target = a.l_start[l], rpart = a.l_rpart[l];
      const int last = a.ix.rec_fwd[a.l_rlast[l]];         // searchIndex(seqId, rangeEndPos + countMinimizerWindows)
      int x = max(lo, rfirst - a.frag_len), y = rfirst;
      if (rpart >= 0) {
        // ... and far closer: the range starts fragment_length - 1 bases before the partner seed p that k_l1 paired with
        // the first one, and rec_bwd[p] is the last record at or before wpos[p] - cmw + 1, which is only
        // fragment_length - cmw bases further on -- that many records at most, wpos being strictly increasing.  Six
        // probes inside two or three cache lines instead of twelve spread over the contig (scattered probes were a tenth
        // of this kernel's HBM traffic).
        y = min(rfirst, a.ix.rec_bwd[rpart] + 1);
        x = max(x, y - (a.frag_len - a.cmw + 1));
      }
      while (x < y) { int mid = x + ((y - x) >> 1); if (wpos[mid] < target) x = mid + 1; else y = mid; }
      const int beg = x;
      const int end0 = a.ix.rec_fwd[beg];                  // searchIndex(seqId, first wpos + countMinimizerWindows)
      // the slide stops at the window position where the last record is admitted; the records dropped by then are
      // the ones before the record active at that position
      const int ndrop = last > end0 ? a.ix.rec_bwd[last - 1] - beg : 0;
      a.l_beg[l] = beg; a.l_end0[l] = end0; a.l_last[l] = last; a.l_ndrop[l] = ndrop;
      // the first super-window is padded to whole 8-event groups: k_l2_scan applies it in bulk, without the pivot logic
      nev = (uint32_t)((((end0 - beg + 7) & ~7) + (last - end0) + ndrop + 7) & ~7);
      records = (uint32_t)(last - beg);
      a.l_nev[l] = nev;
      if (a.scan_class_div) atomicAdd(&sh_class[scan_class(nev, a.scan_class_div)], 1u);
    }
    // exclusive scan of the event counts inside the workgroup
    uint32_t incl = nev;
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) sh_wave[wv] = incl;
    uint32_t rsum = records;
    for (int d = 32; d > 0; d >>= 1) rsum += __shfl_down(rsum, d);
    if (lane == 0) atomicAdd(&sh_records, (unsigned long long)rsum);
    __syncthreads();
    uint32_t off = sh_run + incl - nev;
    for (int q = 0; q < wv; q++) off += sh_wave[q];
    if (c0 + threadIdx.x < l_n) a.l_ioff[l] = off;           // relative to the fragment for now
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int q = 0; q < EV_THREADS / 64; q++) tot += sh_wave[q]; sh_run += tot; }
    __syncthreads();
  }
  if (a.scan_class_div && threadIdx.x < SCAN_CLASSES && sh_class[threadIdx.x])      // (the loci of a fragment sit in ONE region of the numbering)
    atomicAdd(&a.scan_hist[((uint32_t)f & (a.loci.n - 1u)) * SCAN_CLASSES + threadIdx.x], sh_class[threadIdx.x]);
  if (threadIdx.x == 0) {
    const uint32_t region = (uint32_t)f & (a.n_regions - 1);
    const unsigned long long base = atomicAdd(&a.ev_region[region], (unsigned long long)sh_run);   // order of fragments is irrelevant
    sh_ok = base + sh_run <= a.region_cap;
    if (!sh_ok) atomicOr(&a.pinfo[1], (unsigned long long)SPEC_EVENTS);
    sh_base = (uint32_t)(region * a.region_cap + base);
    atomicAdd(&a.rec_region[region], sh_records);
  }
  // Bucket table over the hash range the query sketch actually spans: minimizer hashes are window minima, i.e. heavily
  // skewed towards 0, so the buckets divide [0, 2^bits) with 2^bits > the largest query hash rather than the full 32-bit
  // range.  QT[b] = first query rank whose hash is >= b << qshift (rank i opens the buckets after the one of rank i-1,
  // up to its own); a reference hash beyond the range ranks after all.
  const uint32_t hmax = s > 0 ? Q[s - 1] : 0u;
  const int qshift = max(0, (32 - __clz((int)(hmax | 1u))) - QT_BITS);   // hmax < 2^(qshift + QT_BITS)
  for (int i = threadIdx.x; i < s; i += EV_THREADS) {
    if constexpr (RK) {
      const uint32_t sb = Q[i] >> qshift;                            // sub-bucket of sketch entry i (< 2^QT_BITS: hmax fits)
      atomicOr(&OCC[sb >> EV_SUB_BITS], 1u << (sb & SUB_MASK));
      const int bi = (int)(sb >> EV_SUB_BITS), bp = i ? (int)(Q[i - 1] >> qshift >> EV_SUB_BITS) : -1;
      for (int b = bp + 1; b <= bi; b++) R0[b] = (uint16_t)i;         // first rank at or behind the start of bucket b
    } else {
      const int bi = (int)(Q[i] >> qshift), bp = i ? (int)(Q[i - 1] >> qshift) : -1;
      for (int b = bp + 1; b <= bi; b++) QT[b] = (uint16_t)i;
    }
  }
  __syncthreads();
  if (!sh_ok) {                                                      // the event buffer is too small: void pass
    for (uint32_t i = threadIdx.x; i < l_n; i += EV_THREADS) a.l_nev[l_lo + i] = 0;
    return;
  }
  // The waves of the workgroup take the loci of the fragment round-robin; a wave goes through a locus in *trips* of
  // 64 x EV_RPL records (EV_RPL per lane): first the records of the first super-window, then the later ones, then the
  // stream is copied out.  (Measured and left out: issuing the reads of the next trip before working on the current one,
  // two register sets swapping roles, and fetching the ranges of the next locus a locus ahead -- 187 -> 187..195 us; the
  // kernel was held to be HBM-bound then; it is bound by its vector issue slots, profiles/EXPERIMENTS.md.)
  const uint32_t l_end = l_lo + l_n;
  uint32_t l = l_lo + wv;
  if (l >= l_end) return;                                            // (no workgroup barrier below this line)
  struct Locus { int beg, end0, last, ndrop; uint32_t ioff; };
  auto fetch_locus = [&](uint32_t lx) __attribute__((always_inline)) {        // unconditional loads from a clamped index
    const uint32_t lc = min(lx, l_end - 1);
    return Locus{a.l_beg[lc], a.l_end0[lc], a.l_last[lc], a.l_ndrop[lc], a.l_ioff[lc]};
  };
  auto uniform = [](const Locus &p) __attribute__((always_inline)) {          // the same in all lanes: keep it in scalar registers
    return Locus{__builtin_amdgcn_readfirstlane(p.beg), __builtin_amdgcn_readfirstlane(p.end0), __builtin_amdgcn_readfirstlane(p.last),
                 __builtin_amdgcn_readfirstlane(p.ndrop), (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ioff)};
  };
  struct Trip { uint32_t h[EV_RPL], geo[EV_RPL], pd[EV_RPL]; int32_t pv[EV_RPL], bw[EV_RPL], fw[EV_RPL]; uint8_t rf[EV_RPL]; };
  // the HBM reads of a trip, issued together (unconditionally, from a clamped index: a predicated load would wait for
  // the one before it)
  // (R = records per lane of this trip: EV_RPL, or fewer in the last trip of a phase -- its instructions are paid per
  // record slot, filled or not)
  auto issue = [&](auto rpl_tag, Trip &t, bool first, int t0, int hi) __attribute__((always_inline)) {
    constexpr int R = decltype(rpl_tag)::value;
#pragma unroll
    for (int u = 0; u < R; u++) {
      const int ic = min(t0 + lane + 64 * u, hi - 1);
      if (PACKED) {
        // (a uniform base -- the first record of the trip -- plus a 32-bit byte offset: the load takes both as they are,
        // where indexing with a signed record number cost a sign extension and a 64-bit shift-add per record)
        const uint32_t off = min((uint32_t)(lane + 64 * u), (uint32_t)(hi - 1 - t0)) * (uint32_t)sizeof(uint2);
        const uint2 hg = *(const uint2 *)((const char *)(a.ix.rec_hg + t0) + off);
        t.h[u] = hg.x; t.geo[u] = hg.y;
      } else { t.h[u] = a.ix.rec_hash[ic]; t.rf[u] = a.ix.rec_flags[ic]; t.bw[u] = a.ix.rec_bwd[ic]; t.fw[u] = a.ix.rec_fwd[ic + 1]; }
    }
    if (first) {                                                     // (uniform)
#pragma unroll
      for (int u = 0; u < R; u++) {
        const int ic = min(t0 + lane + 64 * u, hi - 1);
        if (PACKED) t.pd[u] = (uint32_t)a.ix.rec_prev16[ic]; else t.pv[u] = a.ix.rec_prev[ic];
      }
    }
  };
  typedef __attribute__((address_space(3))) T *lds_out_t;
  T *const out_lds = (T *)(lds + ev_sketch_bytes(a.cnt_slots)) + (size_t)wv * a.ev_stage;
  const uint32_t obase = (uint32_t)(uintptr_t)(lds_out_t)out_lds;
  // the events of the records [t0, hi) of one trip; FIRST = records of the first super-window
  auto work = [&](auto rpl_tag, auto first_tag, auto store, const Trip &t, const Locus &p, int t0, int hi) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int R = decltype(rpl_tag)::value;
    const int beg = p.beg, end0 = p.end0, ndrop = p.ndrop;
    const int n_init_pad = (end0 - beg + 7) & ~7;
    // rank of a reference hash among the query hashes and whether it is one of them: the bucket's first rank, then
    // EV_PROBE sketch entries at once; the few hashes that rank behind all of them walk on (the sentinels stop the
    // walk).  Both the sketch and the reference hashes crowd towards 0, so two entries were not enough: nine trips out
    // of ten still took the walk.
    // (membership is read off the entry AT the final rank -- the sketch is sorted and distinct, so that entry is the first
    // one >= the hash: one more LDS read instead of four equality tests whose results, kept as booleans across the walk,
    // the compiler packed into bytes at ten instructions per record)
    int x[R]; bool found[R];
    if constexpr (RK) {
      uint32_t occ[R], r0[R], q0[R], q1[R]; bool more = false;
#pragma unroll
      for (int u = 0; u < R; u++) {
        const uint32_t b = min(t.h[u] >> (qshift + EV_SUB_BITS), (uint32_t)(1 << EV_OCC_BITS));   // the bucket behind the range: rank s
        occ[u] = OCC[b]; r0[u] = R0[b];
      }
#pragma unroll
      for (int u = 0; u < R; u++) {
        // sketch entries of the bucket in sub-buckets BELOW the hash's own: shift its own bit to the top, count, take it off
        // (the shift distance 31 - sub is the complement of the sub-bucket number: the shifter reads five bits)
        const uint32_t top = occ[u] << (~(t.h[u] >> qshift) & SUB_MASK);
        x[u] = (int)((uint32_t)__builtin_popcount(top) + r0[u] - (top >> 31));
      }
#pragma unroll
      for (int u = 0; u < R; u++) { q0[u] = Q[x[u]]; q1[u] = Q[x[u] + 1]; }   // (one ds_read2_b32)
#pragma unroll
      for (int u = 0; u < R; u++) {
        // the sketch is sorted and distinct: the hash ranks behind q0 if q0 is smaller, and is a sketch hash iff it equals the
        // entry at its rank -- q0, or q1 when q0 is smaller
        found[u] = q0[u] == t.h[u] || q1[u] == t.h[u];
        x[u] += q0[u] < t.h[u] ? 1 : 0;
        more = more || q1[u] < t.h[u];
      }
      if (__builtin_amdgcn_ballot_w64(more)) {                       // two sketch entries of its own sub-bucket below it: walk on
#pragma unroll
        for (int u = 0; u < R; u++) if (q1[u] < t.h[u]) { x[u]++; while (Q[x[u]] < t.h[u]) x[u]++; found[u] = Q[x[u]] == t.h[u]; }
      }
    } else {
    uint32_t q[R][EV_PROBE], qx[R]; bool more = false;
#pragma unroll
    for (int u = 0; u < R; u++) x[u] = QT[min(t.h[u] >> qshift, (uint32_t)(1 << QT_BITS))];   // the last bucket is [2^bits, inf): rank s
#pragma unroll
    for (int u = 0; u < R; u++) {
#pragma unroll
      for (int j = 0; j < EV_PROBE; j++) q[u][j] = Q[x[u] + j];
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
#pragma unroll
      for (int j = 0; j < EV_PROBE; j++) x[u] += q[u][j] < t.h[u] ? 1 : 0;     // (a compare and an add-with-carry per entry)
      more = more || q[u][EV_PROBE - 1] < t.h[u];
    }
    if (__builtin_amdgcn_ballot_w64(more)) {
#pragma unroll
      for (int u = 0; u < R; u++) if (q[u][EV_PROBE - 1] < t.h[u]) { while (Q[x[u]] < t.h[u]) x[u]++; }
    }
#pragma unroll
    for (int u = 0; u < R; u++) qx[u] = Q[x[u]];
#pragma unroll
    for (int u = 0; u < R; u++) found[u] = qx[u] == t.h[u];
    }
    if constexpr (PACKED) {
      // Bit arithmetic instead of compare + select: this kernel runs seven waves per SIMD and IS its vector issue slots
      // (profiles/r03_valu_model.json), where v_and / v_or / v_add / v_sub / v_lshrrev cost 2.3 cycles and v_cmp, v_cndmask,
      // v_bfe and the three-operand forms 4.2 (profiles/r03_valu_rates.txt).  With c = n_init_pad - end0 - beg (uniform),
      // Bd / Fd the packed distances and the flags as single bits of the same word:
      //   admit at  c + 2i - Bd,        value  base | (not linked ? unit : 0)
      //   drop  at  c + 2i + 1 + Fd - same,    base | (not linked ? 3 unit : 0) | DROP | (same ? no-eval : 0)
      // unit = 1 << EV_DM for a hash of the query sketch, 1 << EV_DW otherwise (the two-bit field then holds +1 resp. -1).
      const int c2 = n_init_pad - end0 - beg + 2 * (t0 + lane);
#pragma unroll
      for (int u = 0; u < R; u++) {
        const int i = t0 + lane + 64 * u;
        if (i >= hi) continue;
        const uint32_t geo = t.geo[u], ng = ~geo;
        const uint32_t base = (uint32_t)(x[u] + 1) << EV_SLOT;
        const uint32_t unit = (found[u] && x[u] < s) ? (1u << EV_DM) : (1u << EV_DW);
        if (FIRST) {
          const bool prev_in = t.pd[u] <= (uint32_t)(i - beg);
          store((uint32_t)(i - beg), base | (prev_in ? 0u : unit) | (i == end0 - 1 ? 0u : ev_noeval<T>()));
        } else {
          const uint32_t m_ins = 0u - ((ng >> (2 * GEO_BITS)) & 1u);                   // FLAG_INS_LINKED clear
          store((uint32_t)(c2 + 128 * u) - ((geo >> GEO_BITS) & ((1u << GEO_BITS) - 1u)), base | (unit & m_ins));
        }
        if (i - beg < ndrop) {
          const uint32_t same = (geo >> (2 * GEO_BITS + 2)) & 1u;                      // FLAG_SAME_STEP
          const uint32_t m_del = 0u - ((ng >> (2 * GEO_BITS + 1)) & 1u);               // FLAG_DEL_LINKED clear
          store((uint32_t)(c2 + 128 * u + 1) + (geo & ((1u << GEO_BITS) - 1u)) - same,
                base | ((unit + unit + unit) & m_del) | (1u << EV_DROP) | ((0u - same) & ev_noeval<T>()));
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
      const int i = t0 + lane + 64 * u;
      if (i >= hi) continue;
      uint32_t flags; bool prev_in; int32_t bwd, fwd1;
      if (PACKED) {
        flags = t.geo[u] >> (2 * GEO_BITS); prev_in = t.pd[u] <= (uint32_t)(i - beg);
        bwd = i - (int32_t)((t.geo[u] >> GEO_BITS) & ((1u << GEO_BITS) - 1u)); fwd1 = i + 1 + (int32_t)(t.geo[u] & ((1u << GEO_BITS) - 1u));
      } else { flags = t.rf[u]; prev_in = t.pv[u] >= beg; bwd = t.bw[u]; fwd1 = t.fw[u]; }
      // both events carry slot = rank + 1 and ONE signed delta: of the matched bit when the hash is in the query
      // sketch (a sentinel is not a match), of the window-only count otherwise
      const uint32_t base = (uint32_t)(x[u] + 1) << EV_SLOT;
      const int dsh = (found[u] && x[u] < s) ? EV_DM : EV_DW;
      // admit.  First super-window: inserted in record order, compared once after the last one, a no-op when the hash
      // is already in the window (prev_in).  Later: after the drops of all records before the one active at its window
      // position (rec_bwd), a no-op when linked to the previous record of the same hash.
      if (FIRST) store((uint32_t)(i - beg), base | ((prev_in ? 0u : 1u) << dsh) | (i == end0 - 1 ? 0u : ev_noeval<T>()));
      else store((uint32_t)(n_init_pad + (i - end0) + (bwd - beg)), base | (((flags & FLAG_INS_LINKED) ? 0u : 1u) << dsh));
      if (i - beg < ndrop) {
        // dropped at window position wpos[i+1], after the admits of earlier positions and before the admit of
        // that same position (FLAG_SAME_STEP), which then carries the comparison
        const uint32_t same = (flags & FLAG_SAME_STEP) ? 1u : 0u;
        const uint32_t off = (flags & FLAG_DEL_LINKED) ? 0u : 3u;         // -1 in the two-bit field
        store((uint32_t)(n_init_pad + (i - beg) + (fwd1 - (int)same - end0)),
              base | (off << dsh) | (1u << EV_DROP) | (same ? ev_noeval<T>() : 0u));
      }
    }
  };
  // events land at scattered 2-byte positions: the stream of a locus is built in LDS and streamed out in 16-byte pieces
  // (direct 2-byte stores doubled the HBM write traffic); very long streams fall back to direct stores
  struct Stream { uint32_t total, padded; bool staged; T *gout; };
  auto begin_locus = [&](const Locus &p, uint32_t lx) __attribute__((always_inline)) {
    const int n_init = p.end0 - p.beg, n_init_pad = (n_init + 7) & ~7;
    Stream st;
    st.total = (uint32_t)(n_init_pad + (p.last - p.end0) + p.ndrop);
    st.padded = (st.total + 7u) & ~7u;
    st.staged = st.padded <= (uint32_t)a.ev_stage;
    const uint32_t ioff = sh_base + p.ioff;
    if (lane == 0) a.l_ioff[lx] = ioff;                              // absolute, for k_l2_scan
    st.gout = (T *)a.items + ioff;
    const T pad = (T)ev_noeval<T>();                                 // padding: slot 0, no comparison
    T *o = st.staged ? out_lds : st.gout;
    for (uint32_t i = st.total + lane; i < st.padded; i += 64) o[i] = pad;
    if (lane < n_init_pad - n_init) o[n_init + lane] = pad;
    return st;
  };
  auto end_locus = [&](const Stream &st) __attribute__((always_inline)) {
    if (st.staged) {
      __builtin_amdgcn_wave_barrier();
      const uint4 *src = (const uint4 *)out_lds;
      uint4 *dst = (uint4 *)st.gout;
      const uint32_t n16 = st.padded * (uint32_t)sizeof(T) / 16u;
      for (uint32_t i = lane; i < n16; i += 64) dst[i] = src[i];
      __builtin_amdgcn_wave_barrier();
    }
  };

  for (; l < l_end; l += EV_THREADS / 64) {
    const Locus cur = uniform(fetch_locus(l));
    __builtin_amdgcn_wave_barrier();                                 // (the relative offset has been read: begin_locus overwrites it)
    const Stream st = begin_locus(cur, l);
    auto run = [&](auto store) __attribute__((always_inline)) {
      Trip t;
      // whole trips of EV_RPL records per lane, then ONE trip sized for what is left of the phase (a 630-record locus is 240
      // + 390: trips of 256 filled 82 % of their record slots, this fills 94 %)
      auto phase = [&](auto first_tag, int lo, int hi) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int t0 = lo;
        for (; t0 + 64 * EV_RPL <= hi; t0 += 64 * EV_RPL) {
          issue(std::integral_constant<int, EV_RPL>(), t, FIRST, t0, hi); work(std::integral_constant<int, EV_RPL>(), first_tag, store, t, cur, t0, hi);
        }
        const int rem = hi - t0;                                       // (uniform)
        auto tail = [&](auto rpl_tag) __attribute__((always_inline)) { issue(rpl_tag, t, FIRST, t0, hi); work(rpl_tag, first_tag, store, t, cur, t0, hi); };
        if (rem > 64 * 3) tail(std::integral_constant<int, (EV_RPL >= 4 ? 4 : EV_RPL)>());
        else if (rem > 64 * 2) tail(std::integral_constant<int, (EV_RPL >= 3 ? 3 : EV_RPL)>());
        else if (rem > 64) tail(std::integral_constant<int, (EV_RPL >= 2 ? 2 : EV_RPL)>());
        else if (rem > 0) tail(std::integral_constant<int, 1>());
      };
      phase(std::true_type(), cur.beg, cur.end0);
      phase(std::false_type(), cur.end0, cur.last);
    };
    if (st.staged) run([&](uint32_t pos, uint32_t v) __attribute__((always_inline)) { *(lds_out_t)(uintptr_t)(obase + pos * (uint32_t)sizeof(T)) = (T)v; });
    else { T *gout = st.gout; run([&](uint32_t pos, uint32_t v) __attribute__((always_inline)) { gout[pos] = (T)v; }); }
    end_locus(st);
  }
}

// State of one lane: ST[r] = (1 + number of distinct window-only hashes of insertion rank r) << 1 | (query rank r
// currently matched).  ST is uint8 on the fast path (one byte per rank keeps 8 waves per CU resident); a count that
// leaves the byte carries out of it (seen in the 32-bit value before it is stored), the lane flags its locus and the
// uint16 instantiation redoes it (l_redo).  The array is shifted by one slot so that the boundary rank r*-1 needs no
// clamp when r* = 0.
//
// The slide is bound by instruction issue (one lane per locus, ~1000 dependent events; the SIMDs that carry two waves
// set the kernel time), so the state is kept in the form that needs the fewest instructions: the pivot as the LDS
// address of its slot (rl = r* x LNB + lane: comparisons against the event's slot address need no conversion),
// g = r* + #window-only hashes below r* - s (the quantity both pivot tests use, against zero) instead of the terms,
// pivot moves as one signed step `delta` applied with multiply-adds, the count stored with a bias of one because a
// pivot move changes g by count + 1, and the comparison with the best window so far done on a copy of the shared
// count that the event's no-eval bit makes negative.
//
// BASE0: the state starts at LDS address 0 (a kernel without static LDS; the kernel checks it), so that with 64
// one-byte lanes per row the masked event word OR the lane IS the address of the event's slot.
template <typename T, typename ST, int LNT, bool BASE0>
struct Slide {
  typedef __attribute__((address_space(3))) ST *lds_ptr;
  static constexpr int STB = (int)sizeof(ST);
  static constexpr int SBITS = 8 * STB;
  static constexpr int RB = EvBits<T>::RANK;
  int lbase, LNB, s;
  int rl, g, shared, best, beg, opt_s, opt_e;
  uint32_t overflow;

  // LDS byte addresses as plain integers: `extern __shared__` has a link-time base the compiler would add on every access
  static __device__ __forceinline__ uint32_t ld(int addr) { return (uint32_t)*(lds_ptr)(uintptr_t)(uint32_t)addr; }
  static __device__ __forceinline__ void sto(int addr, uint32_t v) { *(lds_ptr)(uintptr_t)(uint32_t)addr = (ST)v; }

  __device__ __forceinline__ void init(ST *st, int lane, int lanes, int sketch, int beg0) {
    lbase = (int)(uint32_t)(uintptr_t)(lds_ptr)st + lane * STB;
    LNB = (LNT ? LNT : lanes) * STB;                              // bytes between consecutive slots of one lane
    s = sketch;
    rl = lbase; g = 0; shared = 0; best = -1; beg = opt_s = opt_e = beg0; overflow = 0;
  }
  static constexpr uint32_t EMPTY = 2u;                           // count 0 (biased), not matched
  __device__ __forceinline__ void clear() { for (int i = 0; i <= s + 1; i++) sto(i * LNB + lbase, EMPTY); }
  // SH: position of the event inside its 32-bit word (a constant after unrolling)
  template <int SH> __device__ __forceinline__ int slot_addr(uint32_t word) const {
    if constexpr (LNT == 64) {
      const uint32_t row = (SH ? word >> SH : word) & ev_slot_mask<T>();          // slot x 64
      if constexpr (BASE0) return (int)((STB == 1 ? row : row * STB) | (uint32_t)lbase);
      else return (int)(STB == 1 ? row : row * STB) + lbase;
    } else {
      return (int)__builtin_amdgcn_ubfe(word, SH + EV_SLOT, RB) * LNB + lbase;
    }
  }
  // an event of the first super-window: its admit only changes the per-rank state
  template <int SH> __device__ __forceinline__ void fill(uint32_t word) {
    const int dM = __builtin_amdgcn_sbfe(word, SH + EV_DM, 2), dW = __builtin_amdgcn_sbfe(word, SH + EV_DW, 2);
    const int addr = slot_addr<SH>(word);
    const uint32_t nv = (ld(addr) + (uint32_t)(dW << 1)) ^ ((uint32_t)dM & 1u);
    overflow |= nv;
    sto(addr, nv);
  }
  // pivot of the filled window: r* = min{r : r + sum_{c <= r} cnt[c] >= s} (s if there is none); the comparison after
  // the last admit of the first window (best was -1: it always wins)
  __device__ __forceinline__ void read_pivot() {
    int rstar = s, F = s, acc = 0, m_acc = 0;
    bool hit = false;
    for (int r = 0; r < s; r++) {
      const uint32_t v = ld((r + 1) * LNB + lbase);
      const int c = (int)(v >> 1) - 1, mt = (int)(v & 1u);
      const bool now = !hit && r + acc + c >= s;
      rstar = now ? r : rstar;
      F = now ? r + acc : F;
      shared = now ? m_acc : shared;
      hit = hit || now;
      acc += c; m_acc += mt;
    }
    F = hit ? F : s + acc;
    shared = hit ? shared : m_acc;
    g = F - s;
    rl = rstar * LNB + lbase;                                     // address of slot r* (= state of rank r*-1)
    best = shared;
  }
  // straight-line selects only: the 64 lanes of a wave follow 64 different loci, any branch would serialise them
  template <int SH> __device__ __forceinline__ void step(uint32_t word) {
    const int dM = __builtin_amdgcn_sbfe(word, SH + EV_DM, 2), dW = __builtin_amdgcn_sbfe(word, SH + EV_DW, 2);
    const int drp = (int)__builtin_amdgcn_ubfe(word, SH + EV_DROP, 1);
    // both LDS reads are issued together: the touched rank, and the rank at the boundary the pivot may move across
    const int addr = slot_addr<SH>(word);
    const int addrb = rl + drp * LNB;                             // slot of rank r* (drop) or r*-1 (admit)
    const uint32_t v = ld(addr), vb0 = ld(addrb);
    // admits of a matched rank set bit 0, drops clear it; window-only hashes count in the bits above
    const uint32_t nv = (v + (uint32_t)(dW << 1)) ^ ((uint32_t)dM & 1u);
    overflow |= nv;                                               // bit SBITS set = the count left its field
    sto(addr, nv);
    const bool below = addr <= rl;                                // rank < r*
    const int dWb = below ? dW : 0;
    shared += below ? dM : 0;
    g += dWb;
    const uint32_t vb = (addrb == addr) ? nv : vb0;
    const int c1 = (int)(vb >> 1), mbm = __builtin_amdgcn_sbfe(vb, 0, 1);   // count + 1 of the boundary rank, -(its matched bit)
    // a window-only hash left and query rank r* re-enters the s smallest of the union (g + count < 0 implies r* < s) ...
    const bool up = ((dW & (g + c1 - 1)) < 0);                    // dW < 0 and g + count < 0: both sign bits set
    // ... or one arrived below r* and f(r*-1) reached s: the largest query rank falls out
    const bool down = dWb > 0 && g > 0;
    const int delta = up ? 1 : (down ? -1 : 0);
    g += __mul24(delta, c1);                                      // r* moves by delta, the count of the crossed rank changes sides
    shared += delta & mbm;
    rl += __mul24(delta, LNB);
    beg += drp;
    // the comparison of this window position, if the event carries one: otherwise its shared count reads as negative
    const int sh_e = (int)((uint32_t)shared | ((word << ((32 - SH - 8 * (int)sizeof(T)) & 31)) & 0x80000000u));
    const bool gt = sh_e > best, ge = sh_e >= best;
    best = max(best, sh_e);
    opt_s = gt ? beg : opt_s;
    opt_e = ge ? beg : opt_e;
  }
};

// The loci of every region into `scan_order`, longest event stream first (second half of the counting sort k_l2_events began):
// workgroup b takes 256 consecutive locus numbers of region b mod n; a locus is placed behind the loci of the longer classes
// (prefix sums of the region's class counts) at the next free place of its class -- one returning atomic per class and workgroup.
// The order inside a class is whatever the atomics make it: every locus is slid on its own, and the group maximum is an atomicMax.
__global__ __launch_bounds__(256) void k_l2_order(L2Args a) {
  __shared__ uint32_t sh_cnt[SCAN_CLASSES], sh_at[SCAN_CLASSES];
  const uint32_t r = blockIdx.x & (a.loci.n - 1u), off = (blockIdx.x / a.loci.n) * 256u + threadIdx.x;
  // (a pass that raised a speculation flag is void and will be repeated: its class counts may not match its stream lengths --
  // k_l2_events zeroes the lengths of a fragment whose events found no room -- so nothing is ordered, and k_l2_scan does nothing)
  if (a.counters[2] || a.pinfo[1]) return;
  const uint32_t live = min(a.loci.count[r], 1u << a.loci.shift);
  if ((blockIdx.x / a.loci.n) * 256u >= live) return;                  // (uniform: nothing of this chunk is live)
  if (threadIdx.x < SCAN_CLASSES) sh_cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t l = (r << a.loci.shift) + off;
  uint32_t c = 0, rank = 0;
  if (off < live) { c = scan_class(a.l_nev[l], a.scan_class_div); rank = atomicAdd(&sh_cnt[c], 1u); }
  __syncthreads();
  if (threadIdx.x < SCAN_CLASSES) {
    uint32_t before = 0;
    for (uint32_t q = 0; q < threadIdx.x; q++) before += a.scan_hist[r * SCAN_CLASSES + q];
    sh_at[threadIdx.x] = before + (sh_cnt[threadIdx.x] ? atomicAdd(&a.scan_cursor[r * SCAN_CLASSES + threadIdx.x], sh_cnt[threadIdx.x]) : 0u);
  }
  __syncthreads();
  if (off < live) a.scan_order[(r << a.loci.shift) + min(sh_at[c] + rank, (1u << a.loci.shift) - 1u)] = l;
}

// LNT = lanes per workgroup at compile time (64) or 0 for the run-time value used when a huge sketch forces fewer lanes
// per workgroup.
template <typename T, typename ST, int LNT>
__global__ __launch_bounds__(L2_THREADS) void k_l2_scan(L2Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr bool REDO = sizeof(ST) > 1;
  const int LN = LNT ? LNT : a.lanes;
  ST *st = (ST *)lds;                                              // [cnt_slots + 1][LN], lane-interleaved
  const int lane = threadIdx.x;
  if (lane >= LN) return;
  // workgroup b takes chunk b / n of region b mod n (regions are filled from their start: the workgroups that have loci come
  // first in dispatch order, as they did with one dense numbering, and the empty ones behind them exit at once)
  const uint32_t region = blockIdx.x & (a.loci.n - 1u), off = (blockIdx.x / a.loci.n) * (uint32_t)LN + (uint32_t)lane;
  if (a.counters[2] || off >= (1u << a.loci.shift) || off >= a.loci.count[region]) return;    // (locus_live of the identity order)
  if (a.scan_order && a.pinfo[1]) return;                             // void pass: no order was made (k_l2_order)
  // place `off` of the region: the locus of that number, or -- sorted by stream length -- the one k_l2_order put there
  const uint32_t l = a.scan_order ? a.scan_order[(region << a.loci.shift) + off] : (region << a.loci.shift) + off;
  if (REDO) { if (!a.l_redo[l]) return; }
  else a.l_redo[l] = 0;
  const int s = a.q_size[a.l_frag[l]];
  if (s + 1 > a.cnt_slots) return;                                  // SPEC_SMAX was raised: void pass
  const int beg0 = a.l_beg[l];
  const uint32_t nev = a.l_nev[l];                                  // multiple of 8
  constexpr int PER = 16 / sizeof(T);                               // events per 16-byte load
  const uint4 *ev = (const uint4 *)((const T *)a.items + a.l_ioff[l]);
  Slide<T, ST, LNT, true> sl;
  sl.init(st, lane, LN, s, beg0);
  sl.clear();
  if (sl.lbase != lane * (int)sizeof(ST)) __builtin_trap();         // BASE0: this kernel has no static LDS

  // ---- the first super-window: its admits only change the per-rank state, so they are applied without the pivot logic
  //      (the pivot, g and the shared count are functions of the state and are read off it afterwards) ----
  const uint32_t ngroups = nev / PER;
  const uint32_t fill_groups = min(ngroups, (uint32_t)((((a.l_end0[l] - beg0) + 7) & ~7) / PER));
  auto word_of = [](const uint4 &g, int q) __attribute__((always_inline)) {
    constexpr int WPE = (int)sizeof(T);
    return (q * WPE / 4 == 0) ? g.x : (q * WPE / 4 == 1) ? g.y : (q * WPE / 4 == 2) ? g.z : g.w;
  };
  uint4 cur = ngroups ? ev[0] : make_uint4(0, 0, 0, 0);
  for (uint32_t g = 0; g < fill_groups; g++) {
    const uint4 nxt = (g + 1 < ngroups) ? ev[g + 1] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      if (sizeof(T) == 2 && (q & 1)) sl.template fill<16>(word_of(cur, q)); else sl.template fill<0>(word_of(cur, q));
    }
    cur = nxt;
  }
  sl.read_pivot();
  if (!fill_groups) sl.best = -1;
  for (uint32_t g = fill_groups; g < ngroups; g++) {
    const uint4 nxt = (g + 1 < ngroups) ? ev[g + 1] : make_uint4(0, 0, 0, 0);   // prefetch: the load is off the chain
#pragma unroll
    for (int q = 0; q < PER; q++) {
      if (sizeof(T) == 2 && (q & 1)) sl.template step<16>(word_of(cur, q)); else sl.template step<0>(word_of(cur, q));
    }
    cur = nxt;
  }
  if (sl.overflow >> Slide<T, ST, LNT, true>::SBITS) {
    if (!REDO) { a.l_redo[l] = 1; atomicAdd(a.redo_count, 1u); return; }
  }
  a.l_shared[l] = sl.best < 0 ? 0 : sl.best;
  a.l_pos[l] = (a.ix.rec_wpos[sl.opt_s] + a.ix.rec_wpos[sl.opt_e]) / 2;
  if (sl.best >= a.pass_lut[s]) {
    unsigned long long key = ((unsigned long long)(uint32_t)sl.best << 32) | (unsigned long long)(0xFFFFFFFFu - l);
    atomicMax(&a.group_best[a.l_group[l]], key);
  }
}

#ifdef FA_EXPERIMENTS
}  // namespace fa
#include "../../scripts/experiments/fa_l2_fused.hip.h"   // k_l2_fused, k_event_bits, k_pack_hf (FA_L2_FUSED=1)
namespace fa {
#endif

// ----------------------------------------------------------------------------------------------------------
// computeCGI.  Step 1 (best mapping per reference genome and query fragment) is the group maximum taken by
// k_l2.  Step 2 (best mapping per reference bin) is an atomicMax into a dense table of bins; step 3 walks the
// bins of every (query genome, reference genome) pair in order and averages in float32, as the reference does.
// Tie-breaks are the canonical ones of DESIGN.md: equal identity -> smaller (refSeqId, refStartPos) in step 1,
// smaller querySeqId in step 2.
// ----------------------------------------------------------------------------------------------------------
struct CgiArgs {
  IndexView ix;
  const unsigned long long *group_best;
  const uint32_t *counters;     // [2] loci overflow flag, [3] wide-state loci
  const int32_t *l_frag, *l_seq, *l_pos;
  const int32_t *q_size;
  const float *ident_lut;       // triangular
  const int32_t *frag_query;    // [F] query genome (batch-local) of a fragment
  const int32_t *frag_qseq;     // [F] fragment number inside its query genome (querySeqId)
  unsigned long long *bins;     // [NQ * total_bins]
  int32_t bin_len;              // fragment_length - 20
  int32_t query_base;           // first query genome of this pass (frag_query is batch-wide)
  uint32_t group_bound;         // locus numbers (= group numbers) lie below this: regions x their capacity
  int32_t wide_launched;        // the wide-state scan ran in this part (k_l2_scan<., uint16_t>): loci that left the byte state are settled
  unsigned long long *stamp;    // stage_stamp: start of the CGI stage
};

__global__ void k_cgi_bins(CgiArgs a) {
  stage_stamp(a.stamp);
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.counters[2] || g >= a.group_bound) return;                    // (a group carries the number of its first locus; the table of
                                                                      //  group maxima is cleared per pass: no maximum, no group)
  // A void part must leave no trace in the bin table, which later parts and the repeat of this one accumulate into: when a
  // locus overflowed the one-byte slide state and the wide pass was not launched, the group maxima lack that locus, and a
  // lesser locus of its group could land in a bin the true best never touches (the host repeats the part, fa_engine.hip)
  if (a.counters[3] && !a.wide_launched) return;
  unsigned long long best = a.group_best[g];
  if (best == 0) return;
  uint32_t l = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu);
  int shared = (int)(best >> 32);
  int f = a.l_frag[l];
  int s = a.q_size[f];
  float ident = a.ident_lut[(size_t)s * (size_t)(s + 1) / 2 + shared];
  int seq = a.l_seq[l];
  int bin = a.ix.contig_bin[seq] + a.l_pos[l] / a.bin_len;
  unsigned long long key = ((unsigned long long)__float_as_uint(ident) << 32) |
                           (unsigned long long)(0xFFFFFFFFu - (uint32_t)a.frag_qseq[f]);
  atomicMax(&a.bins[(size_t)(a.frag_query[f] - a.query_base) * a.ix.total_bins + bin], key);
}

// One wave per (query genome, reference genome) pair: coalesced 64-bin reads, then the non-empty bins are added one
// by one in bin order (readlane + add), which reproduces the reference's sequential float32 sum bit for bit.  The bins
// of eight 64-bin chunks are fetched together so that the sequential chain waits for HBM once per 512 bins, and
// chunks without any mapping are skipped (x + 0.0f == x).  When `emit` is set (small passes) the last workgroup to
// finish also forms the rows -- flag the non-empty pairs, scan, write in (query, genome) order -- which saves a launch.
// Hand-over of a pass to the host (fa_engine.hip, k_publish_status): the status block and, for a one-query call, the hit
// rows are copied into pinned host memory, then the pass number is released for the host that polls it.  Run by one
// workgroup: the last one of k_cgi_rows when that kernel also forms the rows, else a kernel of its own.
constexpr int ROWS_INLINE_MAX = 256;    // rows the publishing workgroup writes into pinned host memory itself
struct PublishArgs {
  uint32_t *status_dev;              // the device status block, as words
  uint32_t *status_host;             // its pinned mirror
  int32_t words;                     // words to copy (everything before the pass number)
  int32_t seq_word;                  // index of the pass number in the host block
  uint32_t seq;                      // 0 = nothing to publish here
  unsigned long long *stamp;         // the stamps of the pass inside the device block ([3] CGI start, [4] end)
  const int32_t *total_rows;         // in the device block
  const fa_cgi_row *rows_dev;
  fa_cgi_row *rows_host;             // nullptr: the rows stay on the device
  int64_t cap;
};
__device__ __forceinline__ void publish_pass(const PublishArgs &p) {
  if (threadIdx.x == 0) {
    p.stamp[4] = __builtin_amdgcn_s_memrealtime();
    if (p.stamp[3] == 0) p.stamp[3] = p.stamp[4];                  // (a pass without pairs has no CGI stage)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.words; i += blockDim.x) p.status_host[i] = p.status_dev[i];
  if (p.rows_host) {
    // every store to the mapped host buffer is a transaction over PCIe (~60 ns a piece, whatever its width): 16-byte pieces,
    // and only for the few rows of a one-query call -- the host fetches more than ROWS_INLINE_MAX rows with one DMA copy
    // (1 450 rows of a 29-genome chunk took 0.44 ms word by word)
    const int64_t n = min((int64_t)*p.total_rows, p.cap);
    static_assert(sizeof(fa_cgi_row) % 4 == 0, "rows are copied in words");
    if (n <= ROWS_INLINE_MAX) {
      const int64_t words = n * (int64_t)(sizeof(fa_cgi_row) / 4);
      const bool aligned = ((((uintptr_t)p.rows_dev) | ((uintptr_t)p.rows_host)) & 15) == 0;
      const int64_t quads = aligned ? words / 4 : 0;
      const uint4 *qs = (const uint4 *)p.rows_dev;
      uint4 *qd = (uint4 *)p.rows_host;
      for (int64_t i = threadIdx.x; i < quads; i += blockDim.x) qd[i] = qs[i];
      const uint32_t *rs = (const uint32_t *)p.rows_dev;
      uint32_t *rd = (uint32_t *)p.rows_host;
      for (int64_t i = quads * 4 + threadIdx.x; i < words; i += blockDim.x) rd[i] = rs[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&p.status_host[p.seq_word], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_publish_status(PublishArgs p) { publish_pass(p); }

struct RowsArgs {
  const unsigned long long *bins;
  const int32_t *genome_bin;
  int32_t total_bins, G, NQ;
  int32_t *row_count;
  float *row_ident;
  int32_t emit;                    // fused row emission (npairs small)
  uint32_t *done;                  // workgroups finished (zeroed by k_clear)
  const int32_t *query_total_frag;
  int32_t query_id_base;
  fa_cgi_row *rows;
  int64_t cap;
  int32_t *total_rows;
  PublishArgs pub;                 // emit only: the last workgroup also hands the pass over to the host (seq != 0)
};

__global__ __launch_bounds__(256) void k_cgi_rows(RowsArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t npairs = (int64_t)a.NQ * a.G;
  if (i < npairs) {
    const int q = (int)(i / a.G), g = (int)(i % a.G);
    const unsigned long long *b = a.bins + (size_t)q * a.total_bins;
    const int x0 = a.genome_bin[g], x1 = a.genome_bin[g + 1];
    int cnt = 0;
    float sum = 0.0f;
    // the loads of up to 2048 bins are issued together (one round trip to HBM for a 5 Mb genome); only the upper word
    // of a bin -- the identity as float bits, never zero for a mapped bin -- is kept
    constexpr int CGI_BATCH = 32;
    for (int x = x0; x < x1; x += 64 * CGI_BATCH) {
      uint32_t bits[CGI_BATCH];
#pragma unroll
      for (int u = 0; u < CGI_BATCH; u++) bits[u] = (x + 64 * u + lane < x1) ? (uint32_t)(b[x + 64 * u + lane] >> 32) : 0u;
#pragma unroll
      for (int u = 0; u < CGI_BATCH; u++) {
        const uint64_t any = __ballot(bits[u] != 0u);
        if (any == 0) continue;                                    // wave-uniform
        cnt += __popcll(any);
        // empty bins hold +0.0f and x + 0.0f == x exactly, so adding all 64 lanes in lane order IS the reference's
        // sequential sum over the non-empty bins; constant lane indices keep the chain at one v_readlane + v_add each
        // (sixteen lanes are read into scalar registers before their sixteen dependent additions: back to back, a
        // v_readlane and the v_add that consumes it wait on each other's scalar operand)
#pragma unroll
        for (int base = 0; base < 64; base += 16) {
          float part[16];
#pragma unroll
          for (int k = 0; k < 16; k++) part[k] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)bits[u], base + k));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 16; k++) sum += part[k];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (lane == 0) {
      a.row_count[i] = cnt;
      a.row_ident[i] = cnt ? sum / (float)cnt : 0.0f;
    }
  }
  if (!a.emit) return;
  // ---- last workgroup done: rows of the whole pass ----
  __shared__ int sh_last, sh_run, sh_wave[4];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) sh_last = atomicAdd(a.done, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!sh_last) return;
  __threadfence();
  // Ordered compaction of the non-empty pairs by this one workgroup: thread t owns `per` consecutive pairs (emit is only set
  // for npairs <= 16384: per <= 64; 512 by default: per <= 2), counts its non-empty ones -- the loads of a thread do not depend
  // on one another: one round trip to L2 --, one scan over the 256 counts gives its first row, then it writes its rows in order.
  const volatile int32_t *rc = a.row_count;
  const volatile float *ri = a.row_ident;
  const int n = (int)npairs;
  const int per = (n + 255) / 256;
  const int p0 = min(n, (int)threadIdx.x * per), p1 = min(n, p0 + per);
  int mine = 0;
  for (int p = p0; p < p1; p += 8) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = p + u < p1 ? rc[p + u] : 0;
#pragma unroll
    for (int u = 0; u < 8; u++) mine += c[u] != 0;
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
  if (lane == 63) sh_wave[wv] = incl;
  __syncthreads();
  int off = incl - mine;
  for (int w = 0; w < wv; w++) off += sh_wave[w];
  if (threadIdx.x == 0) sh_run = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
  for (int p = p0; p < p1 && mine; p++) {
    const int c = rc[p];
    if (c == 0) continue;
    if (off < a.cap) {
      fa_cgi_row r;
      r.query_id = a.query_id_base + p / a.G;
      r.ref_genome_id = p % a.G;
      r.count_seq = c;
      r.total_query_fragments = a.query_total_frag[p / a.G];
      r.identity = ri[p];
      a.rows[off] = r;
    }
    off++; mine--;
  }
  __syncthreads();
  if (threadIdx.x == 0) *a.total_rows = sh_run;
  if (a.pub.seq) {
    __threadfence();
    __syncthreads();                                                 // the rows and their count are written
    publish_pass(a.pub);
  }
}

// ordered compaction of the non-empty (query, genome) pairs into fa_cgi_row records
__global__ void k_emit_rows(const int32_t *row_count, const float *row_ident, const int32_t *row_off, int G, int64_t n,
                            const int32_t *query_total_frag, int32_t query_id_base, fa_cgi_row *rows, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || row_count[i] == 0) return;
  int64_t o = row_off[i];
  if (o >= cap) return;
  int q = (int)(i / G);
  fa_cgi_row r;
  r.query_id = query_id_base + q;
  r.ref_genome_id = (int32_t)(i % G);
  r.count_seq = row_count[i];
  r.total_query_fragments = query_total_frag[q];
  r.identity = row_ident[i];
  rows[o] = r;
}

__global__ void k_flag_nonzero(const int32_t *row_count, int64_t n, int32_t *flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = row_count[i] != 0;
}


// One launch instead of a dozen hipMemsetAsync calls: zero up to 8 device ranges (sizes in 16-byte units).
}  // namespace fa
