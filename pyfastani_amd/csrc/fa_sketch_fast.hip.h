// fa_sketch_fast.hip.h -- K1 for plain-ACGT tiles and windows 3 <= w <= 64: the hot form of winnowed-minimizer extraction.
//
// Same contract as k_sketch_tiles (fa_sketch.hip.h; reference: src/pyfastani/_fastani.pyx:156-222 of pyfastani, getHash =
// include/fastani/map/common_func.pxd:12): per tile the staged (hash, wpos) records in position order + their count.
// What differs is everything behind the hashing loop, which had grown to 44 % of the kernel's time
// (profiles/r03_valu_rates.txt prices the instructions):
//
//   * No 64-bit (hash, position) keys and no doubling passes at all.  With H[j] the canonical hash of position j and
//     +inf where a position holds no k-mer (a k-mer equal to its reverse complement, _fastani.pyx:202), let G(j) be the
//     right-most minimum of H over (j - w, j].  G only ever moves forward, and it moves at j iff
//         c(j) :=  H[j] <= min(H[j-w], M2)   (the newcomer is a minimum of the old window as well: right-most wins,
//                                             the deque's `>=` pop rule, _fastani.pyx:211-212)
//               or H[j-w] <  M2              (the old minimum stood alone at the position that leaves),
//     M2 = min H over the w - 1 positions in between.  The reference emits at a VALID position i iff its deque front
//     differs from the front at the previous valid position (= the last comparison it made, _fastani.pyx:219-222), i.e.
//     iff G moved anywhere in (i_prev, i]:   emit(i) = c(i) or c(t) for some t of the run of positions without a k-mer
//     right before i.  Tiles without such positions (nearly all) need c alone; the others add a byte map of c and a short
//     backward walk.  The first comparison of a sequence (position w - 1) always emits.  Checked against the oracle on
//     CPU before it was written for the GPU (palindromic repeats, (AT)n, ACGT repeats); tests/test_gpu_parity.py.
//   * A thread owns FOUR consecutive positions and reads its w + 4 hashes as whole 16-byte LDS words: the minimum of the
//     w - 4 positions common to its four windows is formed once (v_min3), 4 slow instructions per position instead of
//     two passes over LDS with barriers; the hashes are the only array, so a workgroup needs ~13 KB of LDS and eight
//     fit a CU (the 64-bit fallback had sized it at 27 KB: six).
//   * Ordered compaction by v_mbcnt over the four emission ballots of a wave -- no emit words in LDS, no second pass.
//   * k = 14 and k = 21 hash from the same 256-entry premix tables as k = 16 (MurmurHash3's first multiply of every
//     8-byte lane is linear in the key bytes): a short lane is a table entry of a group padded with code 0 minus a constant.
#pragma once

#include "fa_sketch.hip.h"

namespace fa {

constexpr int SKF_MIN_W = 4, SKF_MAX_W = 64;      // windows served by the brute-force minimum (the others: k_sketch_tiles)
constexpr uint64_t MM_C1 = 0x87c37b91114253d5ULL, MM_C2 = 0x4cf5ad432745937fULL;

// LDS carve-up (dynamic): [image: 2-bit words][H: front pad + hashes + tail pad][valid: ballot words][wave totals]
//                         [premix tables 8 KB (k = 14, 16, 21); reused as the byte map of c by tiles with invalid positions]
struct SkfLayout { uint32_t image_b, h_words, valid_words, table_off, total; };
__host__ __device__ inline SkfLayout skf_layout(int k, int w) {
  SkfLayout L;
  const uint32_t npos_cap = (uint32_t)(TILE + 2 * w - 2);
  L.image_b = (((npos_cap + (uint32_t)k - 1 + 15 + 16) / 16 + 2) * 4 + 15) / 16 * 16;
  L.h_words = ((uint32_t)(w + 3) + npos_cap + 8 + 3) / 4 * 4;
  L.valid_words = (npos_cap / 64 + 2 + 1) / 2 * 2;   // (even: the wave totals behind it are read as one 16-byte word)
  const uint32_t fixed = L.image_b + L.h_words * 4 + L.valid_words * 8 + 16;
  L.table_off = (fixed + 15) / 16 * 16;
  L.total = L.table_off + 4 * 256 * 8;
  return L;
}

// ---- MurmurHash3_x64_128 (seed 42, low 32 bits) from premixed lanes: k = 14 (tail only), 16 (one block), 21 (block + 5) ----
__device__ __forceinline__ uint32_t murmur14_premixed(uint64_t k1c1, uint64_t k2c2) {
  Murmur m;
  m.init();
  m.h2 ^= Murmur::rotl(k2c2, 33) * MM_C1;           // tail, rem = 14 > 8
  m.h1 ^= Murmur::rotl(k1c1, 31) * MM_C2;
  return m.finish(14);
}
__device__ __forceinline__ uint32_t murmur21_premixed(uint64_t k1c1, uint64_t k2c2, uint64_t t1c1) {
  Murmur m;
  m.init();
  m.h1 ^= Murmur::rotl(k1c1, 31) * MM_C2;
  m.h1 = Murmur::rotl(m.h1, 27); m.h1 += m.h2; m.h1 = mul5_add(m.h1, 0x52dce729u);
  m.h2 ^= Murmur::rotl(k2c2, 33) * MM_C1;
  m.h2 = Murmur::rotl(m.h2, 31); m.h2 += m.h1; m.h2 = mul5_add(m.h2, 0x38495ab5u);
  m.h1 ^= Murmur::rotl(t1c1, 31) * MM_C2;           // tail, rem = 5
  return m.finish(21);
}
__device__ __forceinline__ uint64_t lanes(uint64_t lo_entry, uint32_t hi_low32) { return lo_entry + ((uint64_t)hi_low32 << 32); }

// Canonical hash of the k-mer at base offset b of the 2-bit image; false for a k-mer equal to its reverse complement.
// tc = TC1 | TC2 | TR1 | TR2 (PremixTables): TCx[g] = ASCII(group g) * cx, TRx[g] = TCx[reverse complement of group g].
template <int KT>
__device__ __forceinline__ bool skf_hash(const uint32_t *codes, int b, int k_rt, const uint64_t *tc, uint32_t &out) {
  const uint64_t *tc1 = tc, *tc2 = tc + 256, *tr1 = tc + 512, *tr2 = tc + 768;
  if constexpr (KT == 16) {
    return hash_codes16(codes, b, tc, out);
  } else if constexpr (KT == 14) {
    // forward bytes 0-7 = groups at codes 0, 4; bytes 8-13 = group at 8 + the two codes 12, 13 (a group padded with code 0 =
    // 'A' in its upper two bytes: subtract 0x41410000 * c).  Reverse strand: rc(c13) ... rc(c0), i.e. the groups at codes
    // 10, 6, 2 through the rc-composed tables, then rc(c1), rc(c0): the group (0, 0, c0, c1), whose rc ends in 'T', 'T'.
    const uint32_t cf = get16(codes, b);
    const uint32_t fa_ = (uint32_t)(0x41410000ULL * MM_C2), fr_ = (uint32_t)(0x54540000ULL * MM_C2);
    const uint64_t f1 = lanes(tc1[cf & 0xFFu], (uint32_t)tc1[(cf >> 8) & 0xFFu]);
    const uint64_t f2 = lanes(tc2[(cf >> 16) & 0xFFu], (uint32_t)tc2[(cf >> 24) & 0xFu] - fa_);
    const uint64_t r1 = lanes(tr1[(cf >> 20) & 0xFFu], (uint32_t)tr1[(cf >> 12) & 0xFFu]);
    const uint64_t r2 = lanes(tr2[(cf >> 4) & 0xFFu], (uint32_t)tr2[(cf & 0xFu) << 4] - fr_);
    const uint32_t hf = murmur14_premixed(f1, f2), hb = murmur14_premixed(r1, r2);
    out = hf < hb ? hf : hb;
    return hf != hb;
  } else if constexpr (KT == 21) {
    // one block (codes 0-15) + a tail of five bytes (the group at 16 + the single code 20, padded: subtract 0x41414100 * c1);
    // reverse strand: groups at codes 17, 13 | 9, 5 | tail 1, then rc(c0) alone: the group (0, 0, 0, c0), rc ends in 'T' x 3
    const uint32_t lo = get16(codes, b), hi = get16(codes, b + 16) & 0x3FFu;
    const uint32_t fa_ = (uint32_t)(0x41414100ULL * MM_C1), fr_ = (uint32_t)(0x54545400ULL * MM_C1);
    const uint64_t f1 = lanes(tc1[lo & 0xFFu], (uint32_t)tc1[(lo >> 8) & 0xFFu]);
    const uint64_t f2 = lanes(tc2[(lo >> 16) & 0xFFu], (uint32_t)tc2[lo >> 24]);
    const uint64_t ft = lanes(tc1[hi & 0xFFu], (uint32_t)tc1[hi >> 8] - fa_);
    const uint64_t r1 = lanes(tr1[(hi >> 2) & 0xFFu], (uint32_t)tr1[((lo >> 26) | (hi << 6)) & 0xFFu]);
    const uint64_t r2 = lanes(tr2[(lo >> 18) & 0xFFu], (uint32_t)tr2[(lo >> 10) & 0xFFu]);
    const uint64_t rt = lanes(tr1[(lo >> 2) & 0xFFu], (uint32_t)tr1[(lo & 3u) << 6] - fr_);
    const uint32_t hf = murmur21_premixed(f1, f2, ft), hb = murmur21_premixed(r1, r2, rt);
    out = hf < hb ? hf : hb;
    return hf != hb;
  } else {
    return hash_codes<0>(codes, b, k_rt, out);
  }
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }   // v_min3_u32

// One tile through the hot form.  `sink.emit(i, hash, wpos)` receives the i-th record of the tile (position order, any
// thread), `sink.count(n)` their number (thread 0).  The premix tables are written to LDS from tv0 / tv1 when
// `install` is set; a tile with positions that hold no k-mer reuses their bytes (returns false: install them again
// before the next tile).
// WT = window size at compile time (0: a.w); KT = k at compile time (0: a.k, hashed without tables)
template <int KT, int WT, typename Sink>
__device__ __forceinline__ bool skf_tile(const SketchArgs &a, const Tile &t, unsigned char *lds, bool install, const uint4 &tv0, const uint4 &tv1,
                                         Sink sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr bool TABLES = KT == 14 || KT == 16 || KT == 21;
  const int k = KT ? KT : a.k, w = WT ? WT : a.w;
  const SkfLayout L = skf_layout(k, w);
  uint32_t *codes = (uint32_t *)lds;
  uint32_t *Hst = (uint32_t *)(lds + L.image_b);                    // hashes behind a front pad, see FRONT
  uint64_t *valid = (uint64_t *)(lds + L.image_b + (size_t)L.h_words * 4);
  uint32_t *wtot = (uint32_t *)(valid + L.valid_words);
  uint64_t *tc = (uint64_t *)(lds + L.table_off);
  uint8_t *cmap = (uint8_t *)tc;                                    // (tiles with invalid positions, after the hashing)

  const int hb = min(t.pos0, 2 * w - 2);          // halo of k-mer positions in front of the tile
  const int jlo = t.pos0 - hb;                    // first k-mer position computed (sequence-local)
  const int npt = hb + t.npos;                    // k-mer positions computed
  const int nb = npt + k - 1;                     // bases staged
  const int64_t base0 = t.base + jlo;             // store offset of the first staged base
  // H of tile-local position il lives at Hst[FRONT + il]; FRONT >= w entries of +inf stand for "before the sequence / the
  // halo", and FRONT is chosen so that a thread's first read (position hb + 4 tid - w) is 16-byte aligned
  const int FRONT = w + ((4 - (hb & 3)) & 3);
  uint32_t *const Hs = Hst + FRONT;
  if (TABLES && install) { uint4 *dst = (uint4 *)tc; dst[tid] = tv0; dst[tid + SK_THREADS] = tv1; }
  // ---- 1. stage the 2-bit image ----
  const int64_t w0 = base0 >> 4;
  const int shift = (int)(base0 & 15);
  const int nwords = (shift + nb + 15) / 16 + 1;
  for (int i = tid; i < nwords; i += SK_THREADS) codes[i] = a.packed[w0 + i];
  for (int i = tid; i < FRONT; i += SK_THREADS) Hst[i] = 0xFFFFFFFFu;
  if (tid < 8) Hs[npt + tid] = 0xFFFFFFFFu;
  __shared__ int tile_plain;                      // every position holds a k-mer
  if (tid == 0) tile_plain = 1;
  __syncthreads();
  // ---- 2. hash both strands, canonical minimum, validity ----
  for (int j0 = 0; j0 < npt; j0 += SK_THREADS) {
    const int j = j0 + tid;
    bool ok = false;
    uint32_t h = 0xFFFFFFFFu;
    if (j < npt) ok = skf_hash<KT>(codes, shift + j, k, tc, h);
    const uint64_t bal = __ballot(ok);
    if (j < npt) Hs[j] = ok ? h : 0xFFFFFFFFu;
    if (__ballot(j < npt && !ok) && lane == 0) tile_plain = 0;
    if (lane == 0) valid[j0 / 64 + wave] = bal;
  }
  __syncthreads();
  const bool plain = tile_plain != 0;
  const int first_check = (w - 1) - jlo;          // tile-local index of sequence position w - 1: the first comparison

  // ---- 3. the four windows of a thread: c, the emitted hash ----
  // With base = il0 - w (16-byte aligned in Hst) and w = 4 q + r: chunk 0 = H[il0-w .. il0-w+3] (the elements that leave),
  // chunks 1 .. q-1 and the first r elements of chunk q are common to the four windows, H[il0 .. il0+3] = elements
  // r .. r+3 of the pair (chunk q, chunk q+1).
  auto quad = [&](int il0, bool (&cb)[4], uint32_t (&hsh)[4]) __attribute__((always_inline)) {
    const uint4 *p = (const uint4 *)(Hs + (il0 - w));
    const int q = w >> 2, r = w & 3;
    const uint4 A = p[0];
    uint32_t Cm = 0xFFFFFFFFu;
#pragma unroll 4
    for (int c = 1; c < q; c++) { const uint4 v = p[c]; Cm = min3u(Cm, v.x, v.y); Cm = min3u(Cm, v.z, v.w); }
    const uint4 E = p[q], F = p[q + 1];
    uint32_t h0, h1, h2, h3;
    if (r == 0) { h0 = E.x; h1 = E.y; h2 = E.z; h3 = E.w; }
    else if (r == 1) { Cm = min(Cm, E.x); h0 = E.y; h1 = E.z; h2 = E.w; h3 = F.x; }
    else if (r == 2) { Cm = min3u(Cm, E.x, E.y); h0 = E.z; h1 = E.w; h2 = F.x; h3 = F.y; }
    else { Cm = min3u(Cm, E.x, E.y); Cm = min(Cm, E.z); h0 = E.w; h1 = F.x; h2 = F.y; h3 = F.z; }
    // M2 of window j = min over the positions between the one that leaves (A.[j]) and the newcomer (h[j])
    const uint32_t a23 = min(A.z, A.w), h01 = min(h0, h1);
    const uint32_t m0 = min3u(Cm, A.y, a23), m1 = min3u(Cm, a23, h0), m2 = min3u(Cm, A.w, h01), m3 = min3u(Cm, h01, h2);
    cb[0] = h0 <= min(A.x, m0) || A.x < m0; hsh[0] = min(m0, h0);
    cb[1] = h1 <= min(A.y, m1) || A.y < m1; hsh[1] = min(m1, h1);
    cb[2] = h2 <= min(A.z, m2) || A.z < m2; hsh[2] = min(m2, h2);
    cb[3] = h3 <= min(A.w, m3) || A.w < m3; hsh[3] = min(m3, h3);
  };
  const int tt0 = 4 * tid, il0 = hb + tt0;
  bool em[4];
  uint32_t hsh[4];
  quad(il0, em, hsh);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int il = il0 + j;
    em[j] = (em[j] && il > first_check) || il == first_check;       // (nothing is compared before position w - 1)
  }
  if (!plain) {
    // positions without a k-mer: c of every position (the halo's last w - 2 included) as a byte map, then
    // emit(i) = valid(i) and (c(i) or c(t) for a t of the invalid run right before i)
    __syncthreads();                                                // (the tables are dead: every wave is past the hashing)
#pragma unroll
    for (int j = 0; j < 4; j++) cmap[il0 + j] = em[j] ? 1 : 0;
    const int nhalo = min(hb, w - 2);                               // positions in front of the tile a walk can reach
    if (tid < (nhalo + 3) / 4) {
      const int hl0 = hb - 4 * (tid + 1);                           // (hl0 - w >= -FRONT: hb = 2w - 2 whenever there is a halo)
      bool hc[4]; uint32_t hh[4];
      if (hl0 >= 0) {
        quad(hl0, hc, hh);
#pragma unroll
        for (int j = 0; j < 4; j++) { const int il = hl0 + j; cmap[il] = ((hc[j] && il > first_check) || il == first_check) ? 1 : 0; }
      } else {
        for (int il = max(hl0, 0); il < hl0 + 4; il++) cmap[il] = il == first_check ? 1 : 0;   // (only when hb < 4: no window is complete there)
      }
    }
    __syncthreads();
    auto is_valid = [&](int il) __attribute__((always_inline)) { return (valid[il >> 6] >> (il & 63)) & 1ULL; };
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int il = il0 + j;
      bool e = em[j];
      if (il < npt && !is_valid(il)) e = false;
      else if (!e && il > first_check) {
        for (int q = il - 1; q >= max(first_check, 0) && !is_valid(q); q--) if (cmap[q]) { e = true; break; }
      }
      em[j] = e;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) em[j] = em[j] && tt0 + j < t.npos && il0 + j >= first_check;

  // ---- 4. ordered compaction: positions run lane-major inside a wave (a lane owns four consecutive ones) ----
  uint32_t below = 0, mine = 0;
  uint32_t wave_total = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint64_t bal = __ballot(em[j]);
    below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, below));
    wave_total += (uint32_t)__popcll(bal);
  }
  if (lane == 0) wtot[wave] = wave_total;
  __syncthreads();
  const uint4 tot = *(const uint4 *)wtot;
  const uint32_t wbase = (wave > 0 ? tot.x : 0u) + (wave > 1 ? tot.y : 0u) + (wave > 2 ? tot.z : 0u);
  if (tid == 0) sink.count((int32_t)(tot.x + tot.y + tot.z + tot.w));
  const uint32_t out0 = wbase + below;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (em[j]) {
      sink.emit(out0 + mine, hsh[j], t.pos0 + tt0 + j - w + 1);
      mine++;
    }
  }
  return plain;
}

template <int KT, int WT>
__global__ __launch_bounds__(SK_THREADS, 8) void k_sketch_fast(SketchArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.ntiles) {                                // not a tile: one of the zeroing workgroups of a query pass
    if (a.clear.stamp && blockIdx.x == (uint32_t)a.ntiles && tid == 0) a.clear.stamp[3] = 0;   // [3]: CGI stage, if any
    clear_ranges(a.clear, blockIdx.x - (uint32_t)a.ntiles, gridDim.x - (uint32_t)a.ntiles);
    return;
  }
  if (a.clear.stamp && blockIdx.x == 0 && tid == 0) a.clear.stamp[0] = __builtin_amdgcn_s_memrealtime();   // start of the pass
  // the premix tables are on their way (constants of the code object, 16 KB away in L2) while the tile descriptor is fetched
  constexpr bool TABLES = KT == 14 || KT == 16 || KT == 21;
  uint4 tv0 = make_uint4(0, 0, 0, 0), tv1 = tv0;
  if (TABLES) { const uint4 *src = (const uint4 *)d_premix.v; tv0 = src[tid]; tv1 = src[tid + SK_THREADS]; }
  const Tile t = a.tiles[blockIdx.x];
  if (t.exc_n > 0) return;                                          // a tile with other bytes: k_sketch_tiles<0, true> takes it
  struct Staged {                                                    // the records to the staging arrays of the tile
    const SketchArgs &a;
    __device__ __forceinline__ void count(int32_t n) const { a.tile_count[blockIdx.x] = n; }
    __device__ __forceinline__ void emit(uint32_t i, uint32_t hash, int32_t wpos) const {
      const size_t o = (size_t)blockIdx.x * TILE + i;
      a.stage_hash[o] = hash; a.stage_wpos[o] = wpos;
    }
  };
  skf_tile<KT, WT>(a, t, lds, true, tv0, tv1, Staged{a});
}

}  // namespace fa
