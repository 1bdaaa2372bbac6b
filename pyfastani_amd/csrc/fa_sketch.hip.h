// fa_sketch.hip.h -- K1: winnowed-minimizer extraction on gfx950, plus the host packer that feeds it.
//
// What it replaces: skch::CommonFunc::addMinimizers as adapted by pyfastani
// (src/pyfastani/_fastani.pyx:156-222 nucleotide, :252-309 protein) together with getHash
// (include/fastani/map/common_func.pxd:12) and the upper-case / reverse-complement helpers
// (src/pyfastani/_sequtils/sequtils.cpp:22-90).
//
// HBM layout of a sequence store
//   packed[]   2-bit codes (A=0 C=1 G=2 T=3), 16 bases per uint32, little-end first; every sequence starts on a
//              64-base (16-byte) boundary.  Bytes that are not ACGT (after upper-casing) are stored as code 0 and
//              listed in exc_pos[] / exc_val[] (sorted store offsets + the upper-cased byte).
//   bytes[]    protein mode only: the upper-cased residues, 1 byte each.
//   A *tile* is up to TILE consecutive k-mer positions of one sequence (a reference contig, or one query
//   fragment); one workgroup sketches one tile.
//
// This file holds the host packer, the hash functions, and k_sketch_tiles -- the kernel for what the hot form
// (k_sketch_fast, fa_sketch_fast.hip.h: plain-ACGT tiles, windows 4..64) does not serve: protein tiles and nucleotide tiles
// that touch a byte outside ACGT (BYTES = true, hashed from a byte image), windows below 4 or above 64, FA_K1_GENERAL=1.
//
// Algorithm of k_sketch_tiles per tile (all in LDS, no atomics, no MFMA -- this is integer hashing):
//   1. stage the packed words of the tile (+ halo of 2w-2 k-mer positions + k-1 bases) with coalesced loads (BYTES: expand
//      to bytes and patch the exceptions back in);
//   2. hash both strands with MurmurHash3_x64_128(seed 42) -- k = 16 from the 2-bit codes through 256-entry premix tables
//      (the ASCII bytes are never formed), other k from ASCII bytes rebuilt in registers, BYTES from the byte image --,
//      take the canonical minimum and flag strand-symmetric k-mers as invalid (_fastani.pyx:202);
//   3. window minimum: tiles whose positions are all valid take 32-bit passes over the hashes (minima over spans growing
//      x4, then x2); the others rebuild (hash<<32 | ~position) keys and run floor(log2 w) 64-bit doubling passes -- the
//      argmin is the right-most minimum, as the reference's deque keeps (_fastani.pyx:211-212 pops on >=);
//   4. a record is emitted where the window argmin differs from the argmin at the previous valid position
//      (= the deque front changed, _fastani.pyx:219-222); wave ballots + popcount prefix give ordered compaction.
// The one sequential quirk of the reference -- a new front whose hash equals the first emitted record is
// suppressed while that record's wpos is 0 (_fastani.pyx:216,220) -- is resolved per sequence afterwards
// (k_suppress_runs) because it only ever affects the leading run of records of a sequence.
#pragma once

#include "fa_common.h"
#include "fa_host.h"

namespace fa {

constexpr int TILE = 1024;         // k-mer positions per workgroup
constexpr int SK_THREADS = 256;
constexpr int SK_ITERS = TILE / SK_THREADS;

struct Tile {
  int64_t base;     // store offset (in bases / residues) of the first base of the sequence
  int32_t seq_len;  // sequence length
  int32_t pos0;     // first k-mer position of the tile, sequence-local
  int32_t npos;     // k-mer positions in the tile (1..TILE)
  int32_t seq;      // sequence number (contig id, or fragment number)
  int32_t exc_lo;   // first exception inside the tile's base span
  int32_t exc_n;    // number of exceptions inside the span (0 => fast 2-bit path)
};
static_assert(sizeof(Tile) == 32, "tile descriptor layout");

// Where a kernel finds a sequence store in HBM (the buffers may belong to a DevStore or sit inside one upload blob).
struct StoreView {
  const uint32_t *packed = nullptr;
  const uint8_t *bytes = nullptr;
  const int64_t *exc_pos = nullptr;
  const uint8_t *exc_val = nullptr;
  int64_t n_exc = 0;
};

// Device image of a sequence store.
struct DevStore {
  bool protein = false;
  DevBuf<uint32_t> packed;
  DevBuf<uint8_t> bytes;
  DevBuf<int64_t> exc_pos;
  DevBuf<uint8_t> exc_val;
  int64_t n_exc = 0;
  int64_t total = 0;
  StoreView view() const {
    StoreView v;
    v.packed = packed.p; v.bytes = bytes.p; v.exc_pos = exc_pos.p; v.exc_val = exc_val.p; v.n_exc = n_exc;
    return v;
  }
  // upload in pieces (Sketch flush: the copy of chunk c + 1 runs while chunk c is hashed): `begin` sizes the buffers and sends
  // the exception lists, `upload_bases` the packed words (or residue bytes) of the store offsets [lo, hi) -- sequences start on
  // 64-base boundaries, so a range of whole sequences is a range of whole words
  void begin(const HostStore &h, hipStream_t st) {
    protein = h.protein;
    total = h.total;
    n_exc = (int64_t)h.exc_pos.size();
    if (protein) bytes.ensure(h.bytes.size() + 64);
    else {
      packed.ensure(h.packed.size() + 16);
      if (n_exc) { exc_pos.upload(h.exc_pos, st); exc_val.upload(h.exc_val, st); }
    }
  }
  void upload_bases(const HostStore &h, int64_t lo, int64_t hi, hipStream_t st) {
    if (hi <= lo) return;
    if (protein) {
      FA_HIP(hipMemcpyAsync(bytes.p + lo, h.bytes.data() + lo, (size_t)(std::min<int64_t>(hi, (int64_t)h.bytes.size()) - lo), hipMemcpyHostToDevice, st));
    } else {
      const int64_t w0 = lo / 16, w1 = std::min<int64_t>((hi + 15) / 16, (int64_t)h.packed.size());
      FA_HIP(hipMemcpyAsync(packed.p + w0, h.packed.data() + w0, (size_t)(w1 - w0) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
  }
  void upload(const HostStore &h, hipStream_t st) {
    protein = h.protein;
    total = h.total;
    n_exc = (int64_t)h.exc_pos.size();
    if (protein) {
      bytes.ensure(h.bytes.size() + 64);
      bytes.upload(h.bytes.data(), h.bytes.size(), st);
    } else {
      packed.ensure(h.packed.size() + 16);
      packed.upload(h.packed.data(), h.packed.size(), st);
      if (n_exc) { exc_pos.upload(h.exc_pos, st); exc_val.upload(h.exc_val, st); }
    }
  }
};

// Appends the tiles of one sequence slice [off, off+len) (a whole contig, or one query fragment).
// `tile_len` <= TILE k-mer positions per tile (a multiple of 4: a thread owns four consecutive positions): the kernels take any
// tile up to TILE (k1_tile_len below: the length reference sketching uses).
inline void make_tiles(std::vector<Tile> &tiles, const HostStore &hs, int64_t off, int64_t len, int seq, int k, int w, int tile_len = TILE) {
  int64_t npos_total = len - k + 1;
  if (npos_total <= 0) return;
  for (int64_t p0 = 0; p0 < npos_total; p0 += tile_len) {
    Tile t;
    t.base = off; t.seq_len = (int32_t)len; t.pos0 = (int32_t)p0;
    t.npos = (int32_t)std::min<int64_t>(tile_len, npos_total - p0);
    t.seq = seq; t.exc_lo = 0; t.exc_n = 0;
    if (!hs.exc_pos.empty()) {
      int64_t hb = std::min<int64_t>(p0, 2 * (int64_t)w - 2);
      int64_t lo = off + p0 - hb, hi = off + p0 + t.npos + k - 1;  // base span [lo, hi)
      auto a = std::lower_bound(hs.exc_pos.begin(), hs.exc_pos.end(), lo);
      auto b = std::lower_bound(a, hs.exc_pos.end(), hi);
      t.exc_lo = (int32_t)(a - hs.exc_pos.begin());
      t.exc_n = (int32_t)(b - a);
    }
    tiles.push_back(t);
  }
}

// Positions per tile of REFERENCE sketching (round 5).  A workgroup hashes its tile's positions PLUS a halo of 2w - 2 in
// front of it in trips of SK_THREADS: a full tile of 1 024 positions behind a 46-position halo (w = 24) is 1 070 hashes = FIVE
// trips, the last one a single wave's worth.  TILE - (2w - 2) positions per tile make it four full trips at 4.7 % more tiles.
// Measured (profiles/r05_k1_tiles.txt): 142.4 -> 144.9 G bases/s on one 5 Mb genome, 179.1 -> 180.4 on forty -- the fifth trip
// only ever ran on one of the four waves, so this is a per cent, not the fifth of the hashing loop its trip count suggests.
// (Half tiles, asked for to even out the last resident round of a one-genome launch, buy nothing on top: rounds x trips is
// 3 x 4 for 976-position tiles, 4 x 3 for 722, 6 x 2 for 466.)  Query fragments keep whole tiles: 2 985 positions do not fit
// three four-trip tiles (2 980), and a fourth tile costs more than the fifth trip.  FA_K1_TILE = 1024 restores the full tile (A/B).
inline int k1_tile_len(int w) {
  static const int forced = [] { const char *e = getenv("FA_K1_TILE"); const int x = e ? atoi(e) : 0; return (x >= 256 && x <= TILE && x % 4 == 0) ? x : 0; }();
  if (forced) return forced;
  return std::max(256, (TILE - (2 * w - 2)) & ~3);
}

// the same into place: `dst` holds tile_count(len, k, tile_len) tiles (the tile lists of many sequences are filled concurrently)
inline int64_t tile_count(int64_t len, int k, int tile_len = TILE) {
  const int64_t npos_total = len - k + 1;
  return npos_total <= 0 ? 0 : (npos_total + tile_len - 1) / tile_len;
}
inline void make_tiles_at(Tile *dst, const HostStore &hs, int64_t off, int64_t len, int seq, int k, int w, int tile_len = TILE) {
  const int64_t npos_total = len - k + 1;
  if (npos_total <= 0) return;
  for (int64_t p0 = 0; p0 < npos_total; p0 += tile_len) {
    Tile t;
    t.base = off; t.seq_len = (int32_t)len; t.pos0 = (int32_t)p0;
    t.npos = (int32_t)std::min<int64_t>(tile_len, npos_total - p0);
    t.seq = seq; t.exc_lo = 0; t.exc_n = 0;
    if (!hs.exc_pos.empty()) {
      int64_t hb = std::min<int64_t>(p0, 2 * (int64_t)w - 2);
      int64_t lo = off + p0 - hb, hi = off + p0 + t.npos + k - 1;
      auto a = std::lower_bound(hs.exc_pos.begin(), hs.exc_pos.end(), lo);
      auto b = std::lower_bound(a, hs.exc_pos.end(), hi);
      t.exc_lo = (int32_t)(a - hs.exc_pos.begin());
      t.exc_n = (int32_t)(b - a);
    }
    *dst++ = t;
  }
}

// ----------------------------------------------------------------------------------------------------------
// device: MurmurHash3_x64_128, seed 42, low 32 bits (getHash)
// ----------------------------------------------------------------------------------------------------------
// h * 5 + c as a shift-and-add (v_lshl_add_u64) plus an add: the compiler's own choice for a 64-bit multiply by five is
// two v_mad_u64_u32, which issue at a quarter of the rate (the hashing loop is bound by exactly those)
__device__ __forceinline__ uint64_t mul5_add(uint64_t h, uint32_t c) {
  uint64_t t = h;
  asm("" : "+v"(t));                                   // (an opaque copy, or (h << 2) + h is folded back into h * 5)
  return (h << 2) + t + c;
}

struct Murmur {
  uint64_t h1, h2;
  // a 64-bit rotation by a constant is two v_alignbit_b32 on the halves (the generic form compiles to 64-bit shifts and
  // ORs, twice the instructions; there are eight rotations per k-mer position)
  __device__ __forceinline__ static uint64_t rotl(uint64_t x, int r) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (r >= 32) { const uint32_t t = lo; lo = hi; hi = t; r -= 32; }
    if (r == 0) return ((uint64_t)hi << 32) | lo;
    const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - r), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - r);
    return ((uint64_t)nh << 32) | nl;
  }
  __device__ __forceinline__ static uint64_t fmix(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
  }
  __device__ __forceinline__ void init() { h1 = 42; h2 = 42; }
  __device__ __forceinline__ void mix1(uint64_t k1) {
    k1 *= 0x87c37b91114253d5ULL; k1 = rotl(k1, 31); k1 *= 0x4cf5ad432745937fULL; h1 ^= k1;
  }
  __device__ __forceinline__ void mix2(uint64_t k2) {
    k2 *= 0x4cf5ad432745937fULL; k2 = rotl(k2, 33); k2 *= 0x87c37b91114253d5ULL; h2 ^= k2;
  }
  __device__ __forceinline__ void block(uint64_t k1, uint64_t k2) {
    mix1(k1);
    h1 = rotl(h1, 27); h1 += h2; h1 = mul5_add(h1, 0x52dce729u);
    mix2(k2);
    h2 = rotl(h2, 31); h2 += h1; h2 = mul5_add(h2, 0x38495ab5u);
  }
  __device__ __forceinline__ void tail(uint64_t k1, uint64_t k2, int rem) {
    if (rem > 8) mix2(k2);
    if (rem > 0) mix1(k1);
  }
  __device__ __forceinline__ uint32_t finish(int len) {
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix(h1); h2 = fmix(h2);
    return (uint32_t)(h1 + h2);
  }
};

// 4 two-bit codes (8 bits) -> 4 ASCII bytes "ACGT"[code], first base in the low byte
__device__ __forceinline__ uint32_t expand4(uint32_t x) {
  uint32_t t = (x | (x << 12)) & 0x000F000Fu;
  t = (t | (t << 6)) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x54474341u, t);  // selector bytes 0..3 pick 'A','C','G','T'
}
__device__ __forceinline__ uint64_t expand8(uint32_t x16) {
  return (uint64_t)expand4(x16 & 0xFFu) | ((uint64_t)expand4((x16 >> 8) & 0xFFu) << 32);
}
// reverse complement of 16 packed codes: code j of the result = 3 - code (15-j) of the input
__device__ __forceinline__ uint32_t revcomp16(uint32_t f) {
  uint32_t x = __brev(~f);
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
// 16 codes starting at base offset b of the LDS image
__device__ __forceinline__ uint32_t get16(const uint32_t *codes, int b) {
  int idx = b >> 4, sh = (b & 15) * 2;
  return __funnelshift_r(codes[idx], codes[idx + 1], sh);
}
__device__ __forceinline__ uint64_t bytemask(int n) {  // low n bytes set, n in 0..8
  return n >= 8 ? ~0ULL : ((1ULL << (8 * n)) - 1ULL);
}

// canonical hash of the k-mer at base offset b of a 2-bit LDS image; returns false for strand-symmetric k-mers
template <int KT>
__device__ __forceinline__ bool hash_codes(const uint32_t *codes, int b, int k_rt, uint32_t &out) {
  const int k = KT ? KT : k_rt;
  const int nblocks = k >> 4, rem = k & 15;
  Murmur f, r;
  f.init(); r.init();
  for (int bi = 0; bi < nblocks; bi++) {
    uint32_t cf = get16(codes, b + 16 * bi);
    f.block(expand8(cf & 0xFFFFu), expand8(cf >> 16));
    uint32_t cr = revcomp16(get16(codes, b + k - 16 * (bi + 1)));
    r.block(expand8(cr & 0xFFFFu), expand8(cr >> 16));
  }
  if (rem) {
    uint32_t cf = get16(codes, b + 16 * nblocks);
    uint64_t m1 = bytemask(rem < 8 ? rem : 8), m2 = bytemask(rem > 8 ? rem - 8 : 0);
    f.tail(expand8(cf & 0xFFFFu) & m1, expand8(cf >> 16) & m2, rem);
    uint32_t cr = revcomp16(get16(codes, b)) >> (2 * (16 - rem));
    r.tail(expand8(cr & 0xFFFFu) & m1, expand8(cr >> 16) & m2, rem);
  }
  uint32_t hf = f.finish(k), hb = r.finish(k);
  out = hf < hb ? hf : hb;
  return hf != hb;
}

// k = 16 fast path.  The first multiply of each 8-byte half of the MurmurHash3 block is linear in the key bytes:
// with A(g) the four ASCII bytes of the 8-bit code group g,  key*c = A(g0)*c + (A(g1)*c mod 2^32) << 32.
// TC1[g] = A(g) * c1 and TC2[g] = A(g) * c2 (mod 2^64) sit in LDS, so the ASCII bytes are never materialised and 4 of the
// 16 64-bit multiplies per base become two LDS reads and an add each.  The tables are constants of the code object,
// copied into LDS by every workgroup (each used to compute its copy: 31 vector instructions per thread and tile, two
// 64-bit multiplies among them).
struct PremixTables {
  uint64_t v[1024];                                 // TC1 | TC2 | TR1 | TR2, with TRx[g] = TCx[reverse complement of group g]
  constexpr PremixTables() : v() {
    for (uint32_t g = 0; g < 256; g++) {
      uint64_t A = 0;
      for (int j = 0; j < 4; j++) A |= (uint64_t)("ACGT"[(g >> (2 * j)) & 3u]) << (8 * j);
      v[g] = A * 0x87c37b91114253d5ULL;
      v[256 + g] = A * 0x4cf5ad432745937fULL;
    }
    for (uint32_t g = 0; g < 256; g++) {
      uint32_t rc = 0;
      for (int j = 0; j < 4; j++) rc |= (3u - ((g >> (2 * (3 - j))) & 3u)) << (2 * j);   // code j of the reverse complement = 3 - code (3-j)
      v[512 + g] = v[rc];
      v[768 + g] = v[256 + rc];
    }
  }
};
__device__ const PremixTables d_premix = PremixTables();

__device__ __forceinline__ uint32_t murmur16_premixed(uint64_t k1c1, uint64_t k2c2) {
  Murmur m;
  m.init();
  uint64_t k1 = Murmur::rotl(k1c1, 31) * 0x4cf5ad432745937fULL;
  m.h1 ^= k1;
  m.h1 = Murmur::rotl(m.h1, 27); m.h1 += m.h2; m.h1 = mul5_add(m.h1, 0x52dce729u);
  uint64_t k2 = Murmur::rotl(k2c2, 33) * 0x87c37b91114253d5ULL;
  m.h2 ^= k2;
  m.h2 = Murmur::rotl(m.h2, 31); m.h2 += m.h1; m.h2 = mul5_add(m.h2, 0x38495ab5u);
  return m.finish(16);
}
// Both strands from the four 8-bit groups g0..g3 of the forward k-mer: the reverse complement's groups are rc(g3), rc(g2),
// rc(g1), rc(g0), and TR1 / TR2 are the tables pre-composed with rc -- the reverse-complement word is never formed, and a
// group's two table reads share one address.
__device__ __forceinline__ bool hash_codes16(const uint32_t *codes, int b, const uint64_t *tc, uint32_t &out) {
  const uint32_t cf = get16(codes, b);
  const uint32_t g0 = cf & 0xFFu, g1 = (cf >> 8) & 0xFFu, g2 = (cf >> 16) & 0xFFu, g3 = cf >> 24;
  const uint64_t *tc1 = tc, *tc2 = tc + 256, *tr1 = tc + 512, *tr2 = tc + 768;
  const uint64_t f1 = tc1[g0] + ((uint64_t)(uint32_t)tc1[g1] << 32), f2 = tc2[g2] + ((uint64_t)(uint32_t)tc2[g3] << 32);
  const uint64_t r1 = tr1[g3] + ((uint64_t)(uint32_t)tr1[g2] << 32), r2 = tr2[g1] + ((uint64_t)(uint32_t)tr2[g0] << 32);
  const uint32_t hf = murmur16_premixed(f1, f2);
  const uint32_t hb = murmur16_premixed(r1, r2);
  out = hf < hb ? hf : hb;
  return hf != hb;
}

// scalar complement of an upper-cased byte: A<->T, C<->G, IUPAC pairs, everything else unchanged
// (semantics of the reference's scalar table, src/pyfastani/_sequtils/complement.h)
__device__ __forceinline__ uint32_t complement_byte(uint32_t c) {
  switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
    case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
    default: return c;
  }
}

// byte-image variant (protein residues, or nucleotide tiles that contain non-ACGT bytes)
__device__ __forceinline__ bool hash_bytes(const uint8_t *bytes, int b, int k, bool protein, uint32_t &out) {
  const int nblocks = k >> 4, rem = k & 15;
  Murmur f, r;
  f.init(); r.init();
  for (int bi = 0; bi <= nblocks; bi++) {
    int n = bi < nblocks ? 16 : rem;
    if (n == 0) break;
    uint64_t k1 = 0, k2 = 0, q1 = 0, q2 = 0;
    for (int j = 0; j < n; j++) {
      uint64_t c = bytes[b + 16 * bi + j];
      if (j < 8) k1 |= c << (8 * j); else k2 |= c << (8 * (j - 8));
      if (!protein) {
        uint64_t d = complement_byte(bytes[b + k - 1 - (16 * bi + j)]);
        if (j < 8) q1 |= d << (8 * j); else q2 |= d << (8 * (j - 8));
      }
    }
    if (bi < nblocks) { f.block(k1, k2); if (!protein) r.block(q1, q2); }
    else { f.tail(k1, k2, rem); if (!protein) r.tail(q1, q2, rem); }
  }
  uint32_t hf = f.finish(k);
  if (protein) { out = hf; return true; }
  uint32_t hb = r.finish(k);
  out = hf < hb ? hf : hb;
  return hf != hb;
}

// Stage timings of a query pass without host-side events: the first thread of the first kernel of every stage leaves
// the chip-wide 100 MHz counter in the pass's status block (the kernels of a pass run one after the other on a stream,
// so the difference of two stamps is the time of everything in between).
__device__ __forceinline__ void stage_stamp(unsigned long long *stamp) {
  if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = __builtin_amdgcn_s_memrealtime();
}

// Zeroing several device ranges in one go (every counter and table of a query pass).  As a kernel of its own (k_clear) or
// as extra workgroups behind the tiles of k_sketch_tiles, where the memory-bound zeroing runs beside the hashing.
struct ClearArgs {
  uint4 *ptr[8];
  uint64_t n16[8];
  int count;
  unsigned long long *stamp;       // the stamps of the pass: [0] = its start (see stage_stamp)
};
__device__ __forceinline__ void clear_ranges(const ClearArgs &a, uint32_t block, uint32_t blocks) {
  const uint64_t stride = (uint64_t)blocks * blockDim.x;
  for (int r = 0; r < a.count; r++)
    for (uint64_t i = (uint64_t)block * blockDim.x + threadIdx.x; i < a.n16[r]; i += stride) a.ptr[r][i] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_clear(ClearArgs a) {
  if (a.stamp && blockIdx.x == 0 && threadIdx.x == 0) { a.stamp[0] = __builtin_amdgcn_s_memrealtime(); a.stamp[3] = 0; }   // [3]: CGI stage, if any
  clear_ranges(a, blockIdx.x, gridDim.x);
}

struct SketchArgs {
  const Tile *tiles;
  const uint32_t *packed;
  const uint8_t *bytes;
  const int64_t *exc_pos;
  const uint8_t *exc_val;
  uint32_t *stage_hash;   // [ntiles * TILE]
  int32_t *stage_wpos;    // [ntiles * TILE]
  int32_t *tile_count;    // [ntiles]
  int32_t k, w, levels;   // levels = floor(log2(w))
  int32_t protein;
  int32_t code_words;     // LDS words reserved for the 2-bit image / byte image
  int32_t npos_cap;       // LDS key slots: TILE + 2w - 2
  int32_t ntiles;         // workgroups beyond the tiles zero the ranges of `clear` (a query pass)
  int32_t fast;           // the 32-bit window minimum may be used (3 <= w <= 1000 and not switched off)
  ClearArgs clear;
};

// LDS carve-up (dynamic): [image: code_words*4 B][keyA: npos_cap*8][keyB: npos_cap*8][valid: (npos_cap/64+1)*8]
//                         [emit: (TILE/64)*8][prefix: (TILE/64+1)*4]
inline size_t sketch_lds_bytes(int k, int w) {
  size_t npos_cap = (size_t)TILE + 2 * (size_t)w - 2;
  size_t span = npos_cap + (size_t)k - 1 + 64;            // bases (or bytes) staged, with slack for get16 over-read
  size_t image = ((span + 3) / 4 + 4) * 4;                // byte image is the larger of the two
  image = (image + 15) / 16 * 16;
  return image + npos_cap * 16 + (npos_cap / 64 + 1) * 8 + (TILE / 64) * 8 + (TILE / 64 + 1) * 4 + 16 + 4 * 256 * 8;
}

// BYTES = false: tiles of plain ACGT, hashed from the 2-bit image (the hot kernel, kept free of the byte path so that it
// fits 4+ waves per SIMD); BYTES = true: protein tiles and nucleotide tiles that contain other bytes.  Both are launched
// over the same tile range and each skips the tiles of the other kind.
template <int KT, bool BYTES>
__global__ __launch_bounds__(SK_THREADS, BYTES ? 2 : 4) void k_sketch_tiles(SketchArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= a.ntiles) {                                // not a tile: one of the zeroing workgroups of a query pass
    if (a.clear.stamp && blockIdx.x == (uint32_t)a.ntiles && tid == 0) a.clear.stamp[3] = 0;   // [3]: CGI stage, if any
    clear_ranges(a.clear, blockIdx.x - (uint32_t)a.ntiles, gridDim.x - (uint32_t)a.ntiles);
    return;
  }
  if (a.clear.stamp && blockIdx.x == 0 && tid == 0) a.clear.stamp[0] = __builtin_amdgcn_s_memrealtime();   // start of the pass
  const Tile t = a.tiles[blockIdx.x];
  if (BYTES ? (!a.protein && t.exc_n == 0) : (t.exc_n > 0)) return;
  const int k = KT ? KT : a.k, w = a.w;

  uint32_t *codes = (uint32_t *)lds;
  uint8_t *img = (uint8_t *)lds;
  uint64_t *keyA = (uint64_t *)(lds + (size_t)a.code_words * 4);
  uint64_t *keyB = keyA + a.npos_cap;
  uint64_t *valid = keyB + a.npos_cap;
  uint64_t *emit = valid + (a.npos_cap / 64 + 1);
  uint32_t *prefix = (uint32_t *)(emit + TILE / 64);
  // (an offset from `lds`, not a rounded-up pointer value: a pointer that went through an integer is a generic one to
  // the compiler, and the eight table reads per position became flat loads instead of LDS reads)
  const size_t tc_off = ((size_t)((unsigned char *)(prefix + TILE / 64 + 1) - lds) + 15) & ~(size_t)15;
  uint64_t *tc1 = (uint64_t *)(lds + tc_off);                        // TC1 | TC2 | TR1 | TR2 (PremixTables)

  const int hb = min(t.pos0, 2 * w - 2);          // halo of k-mer positions in front of the tile
  const int jlo = t.pos0 - hb;                    // first k-mer position computed (sequence-local)
  const int npt = hb + t.npos;                    // k-mer positions computed
  const int nb = npt + k - 1;                     // bases staged
  const int64_t base0 = t.base + jlo;             // store offset of the first staged base
  const bool byte_mode = BYTES;
  int shift = 0;
  if (!BYTES && KT == 16) for (int i = tid; i < 1024; i += SK_THREADS) tc1[i] = d_premix.v[i];

  // ---- 1. stage the sequence image ----
  if (BYTES && a.protein) {
    for (int i = tid; i < nb; i += SK_THREADS) img[i] = a.bytes[base0 + i];
  } else {
    const int64_t w0 = base0 >> 4;
    shift = (int)(base0 & 15);
    const int nwords = (shift + nb + 15) / 16 + 1;
    if (!byte_mode) {
      for (int i = tid; i < nwords; i += SK_THREADS) codes[i] = a.packed[w0 + i];
    } else {
      // expand to bytes, then patch the non-ACGT bytes back in
      for (int i = tid; i < nb; i += SK_THREADS) {
        int64_t g = base0 + i;
        uint32_t c = (a.packed[g >> 4] >> ((g & 15) * 2)) & 3u;
        img[i] = (uint8_t)((0x54474341u >> (8 * c)) & 0xFFu);
      }
      __syncthreads();
      for (int e = tid; e < t.exc_n; e += SK_THREADS) {
        int64_t p = a.exc_pos[t.exc_lo + e] - base0;
        if (p >= 0 && p < nb) img[p] = a.exc_val[t.exc_lo + e];
      }
      shift = 0;
    }
  }
  __syncthreads();

  // ---- 2. hash both strands, canonical minimum, validity ----
  // The 16 B per position behind the image are one pool of 32-bit words.  Hs[j] = canonical hash of position j.
  uint32_t *const pool = (uint32_t *)keyA;
  uint32_t *const Hs = pool;
  __shared__ int tile_plain;                        // every position valid and no hash equal to the "none" value
  if (tid == 0) tile_plain = 1;
  __syncthreads();
  for (int j0 = 0; j0 < npt; j0 += SK_THREADS) {
    int j = j0 + tid;
    bool ok = false;
    uint32_t h = 0xFFFFFFFFu;
    if (j < npt) {
      if (BYTES) ok = hash_bytes(img, j, k, a.protein != 0, h);
      else if (KT == 16) ok = hash_codes16(codes, shift + j, tc1, h);
      else ok = hash_codes<KT>(codes, shift + j, k, h);
    }
    uint64_t bal = __ballot(ok);
    if (j < a.npos_cap) Hs[j] = ok ? h : 0xFFFFFFFFu;
    if (__ballot(j < npt && (!ok || h == 0xFFFFFFFFu)) && lane == 0) tile_plain = 0;
    if (lane == 0 && (j0 / 64 + wave) <= a.npos_cap / 64) valid[j0 / 64 + wave] = bal;
  }
  __syncthreads();
  const int first_check = (w - 1) - jlo;          // tile-local index of sequence position w-1
  uint32_t rec_hash[SK_ITERS];
  bool rec_emit[SK_ITERS];

  if (a.fast && tile_plain) {
    // ---- 3a / 4a. all positions valid (all but one tile in 10^4: a k-mer equal to its reverse complement is rare):
    // the window minimum and the emission test need the hashes only.  With m(i), p(i) the minimum of the window that
    // ends at i and its rightmost position: p(i) != p(i-1) iff the newcomer is a minimum of the old window as well
    // (rightmost wins: H[i] <= m(i-1)) or the old one stood at the position that leaves, alone (H[i-w] < M2, the minimum
    // of the w-1 positions in between).  M2 comes from a table of minima over sp = 2^floor(log2(w-1)) positions, built
    // in 32-bit passes that multiply the span by four (two passes for w = 24), then by two.
    // Hs | P0 | P1, each with room for the reads past the tile's end, 16-byte aligned: a thread takes FOUR consecutive
    // positions per trip, reads whole 16-byte words and shares the loads and the partial minima between its outputs
    // (the one-position-per-thread form spent 28 instructions per position on the two passes of w = 24, this one 10)
    const int stride = ((a.npos_cap + w + 3) & ~3) + 4;
    int levels32 = 0;
    while ((2 << levels32) <= w - 1) levels32++;
    const int sp = 1 << levels32;
    int src = 0, dst = stride, cur = 1;             // Hs is never written: the passes alternate between P0 and P1
    if (sp >= 4) {
      // span 1 -> 4: out[i] = min(h[i .. i+3]) for i = 0..3 from h[0 .. 6], with the minima of (h2,h3) and (h4,h5) shared
      for (int b = 4 * tid; b < npt; b += 4 * SK_THREADS) {
        const uint4 lo = *(const uint4 *)(pool + src + b), hi = *(const uint4 *)(pool + src + b + 4);
        const uint32_t m23 = min(lo.z, lo.w), m45 = min(hi.x, hi.y);
        uint4 o;
        o.x = min(min(lo.x, lo.y), m23); o.y = min(min(lo.y, m23), hi.x); o.z = min(m23, m45); o.w = min(min(lo.w, m45), hi.z);
        *(uint4 *)(pool + dst + b) = o;
      }
      __syncthreads();
      src = dst; dst = src == stride ? 2 * stride : stride; cur = 4;
    }
    while (cur >= 4 && cur * 4 <= sp) {
      for (int b = 4 * tid; b < npt; b += 4 * SK_THREADS) {
        const uint4 q0 = *(const uint4 *)(pool + src + b), q1 = *(const uint4 *)(pool + src + b + cur);
        const uint4 q2 = *(const uint4 *)(pool + src + b + 2 * cur), q3 = *(const uint4 *)(pool + src + b + 3 * cur);
        uint4 o;
        o.x = min(min(q0.x, q1.x), min(q2.x, q3.x)); o.y = min(min(q0.y, q1.y), min(q2.y, q3.y));
        o.z = min(min(q0.z, q1.z), min(q2.z, q3.z)); o.w = min(min(q0.w, q1.w), min(q2.w, q3.w));
        *(uint4 *)(pool + dst + b) = o;
      }
      __syncthreads();
      src = dst; dst = src == stride ? 2 * stride : stride; cur *= 4;
    }
    if (cur >= 4 && cur * 2 <= sp) {
      for (int b = 4 * tid; b < npt; b += 4 * SK_THREADS) {
        const uint4 q0 = *(const uint4 *)(pool + src + b), q1 = *(const uint4 *)(pool + src + b + cur);
        uint4 o;
        o.x = min(q0.x, q1.x); o.y = min(q0.y, q1.y); o.z = min(q0.z, q1.z); o.w = min(q0.w, q1.w);
        *(uint4 *)(pool + dst + b) = o;
      }
      __syncthreads();
      src = dst; dst = src == stride ? 2 * stride : stride; cur *= 2;
    }
    if (cur * 2 <= sp) {                             // (sp = 2: w = 3 or 4)
      for (int j = tid; j < npt; j += SK_THREADS) pool[dst + j] = min(pool[src + j], pool[src + j + cur]);
      __syncthreads();
      src = dst; dst = src == stride ? 2 * stride : stride; cur *= 2;
    }
    const uint32_t *X = pool + src;
#pragma unroll
    for (int it = 0; it < SK_ITERS; it++) {
      const int tt = it * SK_THREADS + tid;         // tile position
      const int il = hb + tt;                       // tile-local index incl. halo
      bool em = false;
      uint32_t hsh = 0;
      if (tt < t.npos && il >= first_check) {
        const uint32_t H = Hs[il], Hw = Hs[max(il - w, 0)];
        const uint32_t M2 = min(X[il - w + 1], X[il - sp]);
        hsh = min(M2, H);
        em = il == first_check || H <= min(Hw, M2) || Hw < M2;
      }
      rec_hash[it] = hsh;
      rec_emit[it] = em;
      uint64_t bal = __ballot(em);
      if (lane == 0) emit[it * (SK_THREADS / 64) + wave] = bal;
    }
  } else {
    // ---- 3b. general form: 64-bit keys (hash, ~position; all ones for a position without a k-mer), sparse table: after
    //      `levels` doubling passes X[j] = min key over [j, j + 2^levels) ----
    {
      uint32_t hreg[(TILE + 2048) / SK_THREADS + 1];
      const int nreg = (int)(sizeof(hreg) / sizeof(hreg[0]));
      // (npos_cap can exceed what the registers hold only for w > 1024: go through in rounds, back to front, so that
      // no key overwrites a hash that is still to be read -- key j occupies the words 2j and 2j+1 >= j)
      for (int r0 = (a.npos_cap - 1) / (SK_THREADS * nreg) * (SK_THREADS * nreg); r0 >= 0; r0 -= SK_THREADS * nreg) {
#pragma unroll
        for (int q = 0; q < nreg; q++) { const int j = r0 + q * SK_THREADS + tid; hreg[q] = j < a.npos_cap ? Hs[j] : 0u; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < nreg; q++) {
          const int j = r0 + q * SK_THREADS + tid;
          if (j < a.npos_cap) {
            const bool ok = j < npt && ((valid[j >> 6] >> (j & 63)) & 1ULL);
            keyA[j] = ok ? (((uint64_t)hreg[q] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)j)) : ~0ULL;
          }
        }
        __syncthreads();
      }
    }
    uint64_t *X = keyA, *Y = keyB;
    for (int lv = 0; lv < a.levels; lv++) {
      const int step = 1 << lv;
      for (int j = tid; j < npt; j += SK_THREADS) {
        uint64_t v = X[j];
        if (j + step < npt) { uint64_t u = X[j + step]; v = u < v ? u : v; }
        Y[j] = v;
      }
      __syncthreads();
      uint64_t *tmp = X; X = Y; Y = tmp;
    }
    const int span = 1 << a.levels;                 // span <= w < 2*span

    // ---- 4b. emission: the window argmin changed since the previous valid position ----
#pragma unroll
    for (int it = 0; it < SK_ITERS; it++) {
      const int tt = it * SK_THREADS + tid;         // tile position
      const int il = hb + tt;                       // tile-local index incl. halo
      bool em = false;
      uint32_t hsh = 0;
      if (tt < t.npos && il >= first_check && ((valid[il >> 6] >> (il & 63)) & 1ULL)) {
        uint64_t a1 = X[il - w + 1], a2 = X[il - span + 1];
        uint64_t cur = a1 < a2 ? a1 : a2;
        hsh = (uint32_t)(cur >> 32);
        // previous valid position that was itself checked, within the last w-1 positions
        int lo = max(first_check, il - w + 1), prev = -1;
        for (int q = il - 1; q >= lo; q--) {
          if ((valid[q >> 6] >> (q & 63)) & 1ULL) { prev = q; break; }
        }
        if (prev < 0) em = true;
        else {
          uint64_t b1 = X[prev - w + 1], b2 = X[prev - span + 1];
          uint64_t old = b1 < b2 ? b1 : b2;
          em = (uint32_t)old != (uint32_t)cur;      // low words hold ~position of the argmin
        }
      }
      rec_hash[it] = hsh;
      rec_emit[it] = em;
      uint64_t bal = __ballot(em);
      if (lane == 0) emit[it * (SK_THREADS / 64) + wave] = bal;
    }
  }
  __syncthreads();

  // ---- 5. ordered compaction ----
  if (wave == 0) {
    int c = lane < TILE / 64 ? __popcll(emit[lane]) : 0;
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane < TILE / 64) prefix[lane] = incl - c;
    if (lane == TILE / 64 - 1) { prefix[TILE / 64] = incl; a.tile_count[blockIdx.x] = incl; }
  }
  __syncthreads();
  const size_t out0 = (size_t)blockIdx.x * TILE;
#pragma unroll
  for (int it = 0; it < SK_ITERS; it++) {
    if (rec_emit[it]) {
      int word = it * (SK_THREADS / 64) + wave;
      uint64_t below = emit[word] & ((1ULL << lane) - 1ULL);
      size_t o = out0 + prefix[word] + __popcll(below);
      a.stage_hash[o] = rec_hash[it];
      a.stage_wpos[o] = t.pos0 + it * SK_THREADS + tid - w + 1;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// the leading-run quirk: if the first record of a sequence has wpos 0, the records that directly follow it with
// the same hash were never emitted by the reference (their wpos field is still 0, so they compare equal to
// minimizerIndex.back(), _fastani.pyx:216,220).  drop[s] = length of that run.
// ----------------------------------------------------------------------------------------------------------
__global__ void k_suppress_runs(const int32_t *seq_tile_lo, int nseq, const int32_t *tile_count, const uint32_t *stage_hash,
                                const int32_t *stage_wpos, int32_t *drop) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseq) return;
  int t0 = seq_tile_lo[s], t1 = seq_tile_lo[s + 1];
  int d = 0;
  bool have_first = false, done = false;
  uint32_t h0 = 0;
  for (int t = t0; t < t1 && !done; t++) {
    int c = tile_count[t];
    size_t o = (size_t)t * TILE;
    for (int i = 0; i < c; i++) {
      if (!have_first) {
        if (stage_wpos[o + i] != 0) { done = true; break; }
        h0 = stage_hash[o + i];
        have_first = true;
      } else if (stage_hash[o + i] == h0) d++;
      else { done = true; break; }
    }
  }
  drop[s] = d;
}

// copies the staged records of every tile to their final, position-ordered place (reference-side sketching)
__global__ __launch_bounds__(256) void k_compact_records(const Tile *tiles, const int32_t *tile_count, const int32_t *tile_off,
                                                         const int32_t *seq_tile_lo, const int32_t *drop,
                                                         const int32_t *drop_off, const uint32_t *stage_hash,
                                                         const int32_t *stage_wpos, const int32_t *seq_ids, int64_t out_base,
                                                         uint32_t *rec_hash, int32_t *rec_seq, int32_t *rec_wpos) {
  const int t = blockIdx.x;
  const int s = tiles[t].seq;
  const int c = tile_count[t];
  const int first = tile_off[t] - tile_off[seq_tile_lo[s]];   // records of this sequence before the tile
  const int d = drop[s];
  const int64_t seq_out = out_base + tile_off[seq_tile_lo[s]] - drop_off[s];
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    int r = first + i;
    if (r >= 1 && r <= d) continue;
    int64_t o = seq_out + (r > d ? r - d : r);
    rec_hash[o] = stage_hash[(size_t)t * TILE + i];
    rec_seq[o] = seq_ids[s];
    rec_wpos[o] = stage_wpos[(size_t)t * TILE + i];
  }
}

}  // namespace fa
