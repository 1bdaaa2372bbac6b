// fa_l2_fused.hip.h -- EXPERIMENT (round 2), not part of the product build.
//
// The fused form of the L2 stage: event generation and the sequential slide in one launch, the event stream of a locus
// never leaves the compute unit.  Bit-exact, measured 2x slower than k_l2_events + k_l2_scan on the bench step (0.81 ms
// against 0.42 ms; profiles/EXPERIMENTS.md), so it is compiled only with -DFA_EXPERIMENTS (fa_map.hip.h includes this
// file then, and FA_L2_FUSED=1 selects it at run time).  The default libfastani_hip.so holds none of these symbols.
#pragma once

namespace fa {

// The time order of the slide's events is a property of the contig, not of the query: record i is admitted at window
// position wpos[i] - cmw + 1 and dropped at wpos[i+1]; at one position the drop comes first.  Numbering the events of a
// contig in that order, the admit of record i is event  i + rec_bwd[i]  (i admits and rec_bwd[i] drops precede it, in
// absolute record numbers) and the drop of record j is event  j + rec_fwd[j+1] - same_step(j).  ev_bits holds bit 1 at
// the admit positions: a slide that is at event E with A admits behind it finds, in the next 64 bits, which of its
// next events are admits (of records A, A+1, ...) and which are drops (of records E-A, E-A+1, ...) -- by popcount, without
// a search, and any stretch of the stream can be generated on its own.  Records of the first super-window of a contig
// are never admitted by a slide (they are its initial content), so their bits stay clear.
__global__ void k_event_bits(const int32_t *rec_seq, const int32_t *rec_bwd, const int32_t *contig_rec, int64_t N, uint32_t *ev_bits) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t b = rec_bwd[i];
  if (b < contig_rec[rec_seq[i]]) return;
  const uint64_t pos = (uint64_t)i + (uint64_t)b;
  atomicOr(&ev_bits[pos >> 5], 1u << (pos & 31));
}


// ----------------------------------------------------------------------------------------------------------
// L2, fused: the event stream of a locus never leaves the compute unit (k_l2_events + k_l2_scan in one launch, no
// round trip of the events through HBM).
//
// One workgroup per query fragment, two waves.  Wave 0 is the *slider*: one lane per candidate locus of the fragment
// (64 at a time), the same branch-free sequential slide as k_l2_scan, reading its events from a ring in LDS.  Wave 1
// is the *producer*: for every locus it generates the next FU_C events of its time-ordered stream straight into the
// ring, one lane per EVENT -- which record an event belongs to comes from the merged admit/drop order kept as one bit
// per event in the index (k_event_bits): with A admits behind a stream that stands at event E, the set bits of the
// next FU_C positions are the admits of records A, A+1, ... and the clear bits the drops of records E-A, E-A+1, ...
// So the reads of the records (rec_hf: hash + flags + distance to the previous record of the same hash, 8 bytes) are two
// short coalesced runs per locus, the rank of the hash in the query sketch is one bucket probe + a short search in LDS
// (done once for the admit and once for the drop of a record: cheaper than carrying 240 ranks per locus in LDS), and
// the event lands at a fixed ring slot -- no scatter, no event arena.  The ring is double buffered by row (FU_C events
// per locus): the producer fills row r+1 while the slider consumes row r, one workgroup barrier per row, and the
// producer's loads run one (records) and two (order bits) rows ahead of the row it composes.  The first rows carry the
// initial super-window (admits in record order, applied without the pivot logic), padded so that every lane reads the
// pivot off its state at the same row.
//
// LDS per workgroup at sketches <= 256: 16.1 KB of slide state + 1 KB sketch + 4 KB ring + 1.3 KB tables = 22.4 KB, i.e.
// seven workgroups per CU -- the 1666 fragments of a 5 Mb query are resident at once (the slide is a latency-bound
// chain per locus: a second round of workgroups would double the time).  blockIdx is mapped to fragments so that
// neighbouring fragments -- whose loci overlap on the reference -- share an XCD and its L2.
// ----------------------------------------------------------------------------------------------------------
constexpr int FU_QT_BITS = 8;               // bucket table resolution
constexpr int FU_PROBE = 4;                 // sketch entries compared at once per rank lookup

// FU_C = events per locus and ring row (16, or 8 when the longer ring would cost a workgroup per CU)
template <typename ST>
__host__ __device__ inline size_t fused_lds_bytes(int cnt_slots, int ev_bytes, int fu_c) {
  size_t state = ((size_t)(cnt_slots + 1) * 64 * sizeof(ST) + 15) / 16 * 16;
  size_t q = ((size_t)(cnt_slots + FU_PROBE) * 4 + 15) / 16 * 16;
  return state + q + (size_t)2 * fu_c * 64 * ev_bytes;
}
constexpr size_t FU_STATIC_LDS = 528;       // QT + the three scalars (checked against the compiler's figure at launch)

// hash, flags and the (saturated) distance to the previous record of the same hash in one 8-byte record for the fused
// kernel: y = flags | min(i - rec_prev[i], 65535) << 8   (65535 also for "no earlier record of this hash in the contig")
__global__ void k_pack_hf(const uint32_t *rec_hash, const uint8_t *rec_flags, const int32_t *rec_prev, int64_t N, uint2 *rec_hf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t pv = rec_prev[i];
  const uint32_t d = pv < 0 ? 65535u : (uint32_t)min((int64_t)65535, i - (int64_t)pv);
  rec_hf[i] = make_uint2(rec_hash[i], (uint32_t)rec_flags[i] | (d << 8));
}

// Workgroup barrier that orders LDS accesses only: __syncthreads() also drains every outstanding global load
// (s_waitcnt vmcnt(0)), which would expose the latency of the producer's prefetches at every ring row.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// NPROD producer waves per workgroup (each serves 64 / NPROD loci); FU_C events per locus and ring row
template <typename T, typename ST, bool REDO, int FU_C, int NPROD>
__global__ __launch_bounds__(64 * (1 + NPROD), (NPROD == 2 && FU_C == 8) ? 6 : 4) void k_l2_fused(L2Args a, int64_t n_frag) {
  extern __shared__ __align__(16) unsigned char lds[];
  if (REDO == false) stage_stamp(a.stamp);
  constexpr int FU_THREADS = 64 * (1 + NPROD);
  constexpr int FU_UPI = 64 / FU_C;                    // loci per producer iteration (of one wave)
  constexpr int FU_ITERS = 64 / NPROD / FU_UPI;        // producer iterations per row (of one wave)
  static_assert(FU_ITERS >= 2 && FU_ITERS <= FU_C, "every locus of a producer wave needs an owner lane");
  constexpr int RB = EvBits<T>::RANK;
  constexpr int SBITS = 8 * (int)sizeof(ST);
  __shared__ uint16_t QT[(1 << FU_QT_BITS) + 2];
  __shared__ int sh_fill_rows, sh_rows, sh_steps;
  // fragments of one XCD (blockIdx % 8, the dispatch order of workgroups) are consecutive
  const int64_t per_xcd = (n_frag + 7) / 8;
  const int64_t f64 = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int64_t)(blockIdx.x >> 3) >= per_xcd || f64 >= n_frag) return;
  const int f = (int)f64;
  const uint32_t l_lo = a.f_loci_lo[f], l_n = a.f_loci_n[f];
  if (l_n == 0) return;
  const int s = a.q_size[f];
  if (a.counters[2] || s > a.cnt_slots - 1) return;                  // loci overflowed / sketch larger than speculated: void pass
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // FA_FUSED_DEBUG & 8: per-workgroup time stamps {start, first row, end, hardware id, rows, loci} into the (otherwise
  // unused) event arena, read back with fa_mapper_debug_items
  unsigned long long *stamp = (a.dbg & 8) ? (unsigned long long *)a.items + (size_t)blockIdx.x * 8 : nullptr;
  if (stamp && tid == 0) { stamp[0] = __builtin_amdgcn_s_memrealtime(); stamp[3] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); stamp[5] = l_n; stamp[6] = 0; stamp[7] = 0;
    // workgroups alive when this one starts (a running count kept behind the stamps)
    stamp[5] |= (unsigned long long)atomicAdd((unsigned int *)((unsigned long long *)a.items + (size_t)gridDim.x * 8), 1u) << 32; }
  ST *st = (ST *)lds;                                                // [cnt_slots + 1][64], lane-interleaved
  uint32_t *Q = (uint32_t *)(lds + ((size_t)(a.cnt_slots + 1) * 64 * sizeof(ST) + 15) / 16 * 16);
  T *ring = (T *)((unsigned char *)Q + ((size_t)(a.cnt_slots + FU_PROBE) * 4 + 15) / 16 * 16);   // [2][FU_C][64]
  // between groups the ring doubles as the hand-over of the locus ranges from the slider's lanes to the producer's
  int32_t *u_beg = (int32_t *)ring;                                   // first record of the locus range (-1: lane unused)
  uint32_t *u_e0 = (uint32_t *)ring + 64, *u_nmain = (uint32_t *)ring + 128;   // events behind the first window; events of the slide
  for (int i = tid; i < s + FU_PROBE; i += FU_THREADS) Q[i] = i < s ? a.q_hash[(size_t)f * a.qcap + i] : 0xFFFFFFFFu;   // + sentinels
  __syncthreads();
  // Bucket table for the rank lookups.  Minimizer hashes are window minima: their density falls off exponentially from 0,
  // so the buckets are laid over 1 - 2^(-c h) (monotone, one v_exp_f32) rather than over h itself -- about one sketch
  // entry per bucket everywhere, instead of a dozen in the first buckets.  QT[b] = first rank whose bucket is >= b.
  const uint32_t hmax = s > 0 ? Q[s - 1] : 0u;
  const float bscale = -8.0f / (float)max(hmax, 1u);                 // 2^-8 of the mass beyond the largest hash
  auto bucket_of = [&](uint32_t h) __attribute__((always_inline)) {
    const float t = __builtin_amdgcn_exp2f((float)h * bscale);      // 1 .. 2^-8 .. 0
    return (uint32_t)min(255.0f, 256.0f - 256.0f * t);              // 0 .. 255 (monotone in h)
  };
  if (tid == 0) sh_steps = 0;
  for (int b = tid; b <= (1 << FU_QT_BITS) + 1; b += FU_THREADS) QT[b] = (uint16_t)s;
  __syncthreads();
  for (int i = tid; i < s; i += FU_THREADS) {
    const uint32_t bi = bucket_of(Q[i]), bp = i > 0 ? bucket_of(Q[i - 1]) : 0xFFFFFFFFu;
    // rank i opens every bucket in (bucket of rank i-1, bucket of rank i]
    if (i == 0) { for (uint32_t b = 0; b <= bi; b++) QT[b] = 0; }
    else for (uint32_t b = bp + 1; b <= bi; b++) QT[b] = (uint16_t)i;
  }
  __syncthreads();
  {
    // sh_steps: further FU_PROBE-wide probes that the fullest bucket needs after the first one
    int width = 0;
    for (int b = tid; b < (1 << FU_QT_BITS); b += FU_THREADS) width = max(width, (int)QT[b + 1] - (int)QT[b]);
    for (int d = 32; d > 0; d >>= 1) width = max(width, __shfl_xor(width, d));
    if (lane == 0 && width > FU_PROBE) atomicMax(&sh_steps, (width - 1) / FU_PROBE);
  }
  typedef __attribute__((address_space(3))) ST *lds_ptr;
  constexpr int STB = (int)sizeof(ST);
  constexpr int LNB = 64 * STB;                                      // bytes between consecutive slots of one lane
  const int lbase = (int)(uint32_t)(uintptr_t)(lds_ptr)st + lane * STB;
  const int32_t *wpos = a.ix.rec_wpos;

  for (uint32_t g0 = 0; g0 < l_n; g0 += 64) {
    // ---- the loci of this group: record ranges (the three searchIndex calls of computeL2MappedRegions) ----
    __syncthreads();
    int32_t my_locus = -1;                                             // slider lanes: the locus of this lane
    int my_beg = 0;
    if (wv == 0) {
      const uint32_t l = l_lo + g0 + lane;
      bool active = g0 + lane < l_n;
      if (active) { if (REDO) active = a.l_redo[l] != 0; else a.l_redo[l] = 0; }
      int beg = -1, fill_rows = 0, main_rows = 0;
      uint32_t e0 = 0, nmain = 0, records = 0, nev = 0;
      if (active) {
        const int lo = a.ix.contig_rec[a.l_seq[l]];
        const int rfirst = a.l_rfirst[l], target = a.l_start[l];
        int x = max(lo, rfirst - a.frag_len), y = rfirst;
        while (x < y) { int mid = (x + y) >> 1; if (wpos[mid] < target) x = mid + 1; else y = mid; }
        beg = x;
        const int end0 = a.ix.rec_fwd[beg];
        const int last = a.ix.rec_fwd[a.l_rlast[l]];
        const int ndrop = last > end0 ? a.ix.rec_bwd[last - 1] - beg : 0;
        e0 = (uint32_t)beg + (uint32_t)end0;                           // events behind the first super-window
        nmain = last > end0 ? (uint32_t)(last - end0) + (uint32_t)ndrop : 0u;
        fill_rows = (end0 - beg + FU_C - 1) / FU_C;
        main_rows = (int)((nmain + FU_C - 1) / FU_C);
        records = (uint32_t)(last - beg);
        nev = (uint32_t)(end0 - beg) + nmain;
        my_locus = (int32_t)l;
        my_beg = beg;
      }
      u_beg[lane] = beg; u_e0[lane] = e0; u_nmain[lane] = nmain;
      int fr = fill_rows, mr = main_rows;
      uint32_t rsum = records, esum = nev;
      for (int d = 32; d > 0; d >>= 1) {
        fr = max(fr, __shfl_xor(fr, d)); mr = max(mr, __shfl_xor(mr, d));
        rsum += __shfl_xor(rsum, d); esum += __shfl_xor(esum, d);
      }
      if (lane == 0) {
        sh_fill_rows = fr; sh_rows = fr ? fr + mr : 0;
        if (!REDO && rsum) { atomicAdd(a.rec_total, (unsigned long long)rsum); atomicAdd(&a.pinfo[0], (unsigned long long)esum); }
      }
    }
    for (int i = tid; i < (s + 2) * 64; i += FU_THREADS) st[i] = (ST)Slide<T, ST, 64, false>::EMPTY;
    __syncthreads();
    const int R_fill = sh_fill_rows, R_all = sh_rows, n_steps = sh_steps;
    if (R_all == 0) continue;
    if (stamp && tid == 0 && g0 == 0) { stamp[1] = __builtin_amdgcn_s_memrealtime(); stamp[4] = (unsigned long long)R_all | ((unsigned long long)R_fill << 32); }

    // ---- producer: a three-stage pipeline over the rows, so that no stage waits for the loads it issued itself ----
    //   A(row)  main rows: load the merged-order bits of the row                         (global, one row-time ahead of B)
    //   B(row)  which record every event of the row belongs to; load that record (rec_hf) (global, one row-time ahead of C)
    //   C(row)  rank of the hash in the query sketch, compose the event, store it in ring buffer row & 1    (LDS only)
    // One producer step runs C(t), B(t+1), A(t+2); the slider consumes row t-1 meanwhile.  Everything is straight-line,
    // select-based code with the sixteen iterations of a row independent of each other, so that their LDS round trips
    // overlap: one wave produces for 64 loci and cannot afford a chain of dependent probes per iteration.
    const int pe = lane & (FU_C - 1);                                  // event of the row this producer lane composes
    const int pu = lane / FU_C;                                        // its locus inside an iteration
    // the cursors of a locus live in ONE lane (lane (pu, pe) owns locus pe * FU_UPI + pu) and reach the 16 lanes that
    // compose the locus's events by cross-lane reads
    const int ubase = (wv > 0 ? wv - 1 : 0) * (64 / NPROD);           // first locus this producer wave serves
    const int own = ubase + (pe < FU_ITERS ? pe : 0) * FU_UPI + pu;    // (lanes with pe >= FU_ITERS own nothing)
    const bool owner = pe < FU_ITERS;
    int own_beg = -1;
    uint32_t own_e0 = 0u, own_nfill = 0u, own_nmain = 0u, own_ia = 0u;
    uint32_t own_w0 = 0u, own_w1 = 0u;                                 // A -> B: order bits of the owned locus's row (raw words)
    uint2 phf[FU_ITERS];                                               // B -> C: the raw records
    uint32_t pok = 0u, padmit = 0u;                                    // B -> C: bit it = event of iteration it exists / is an admit
    auto stage_a = [&](int row) __attribute__((always_inline)) {
      if (row < R_fill || row >= R_all) return;
      const uint32_t off = (uint32_t)(row - R_fill) * FU_C;
      const uint32_t wi = (own_beg >= 0 && off < own_nmain) ? ((own_e0 + off) >> 5) : 0u;   // unconditional loads (see stage B)
      own_w0 = a.ix.ev_bits[wi];
      own_w1 = a.ix.ev_bits[wi + 1];
    };
    auto stage_b = [&](int row) __attribute__((always_inline)) {
      pok = 0u; padmit = 0u;
      if (row >= R_all) return;
      uint32_t prec[FU_ITERS];
      if (row < R_fill) {
        const uint32_t k0 = (uint32_t)row * FU_C;
        const uint32_t my_first = (uint32_t)max(own_beg, 0) + k0;       // first record of the owned locus's row
        const uint32_t my_n = own_nfill > k0 ? min(own_nfill - k0, (uint32_t)FU_C) : 0u;
#pragma unroll
        for (int it = 0; it < FU_ITERS; it++) {
          const int src = pu * FU_C + it;                              // the lane that owns locus it * FU_UPI + pu
          const uint32_t first = (uint32_t)__shfl((int)my_first, src), n = (uint32_t)__shfl((int)my_n, src);
          const bool ok = (uint32_t)pe < n;
          prec[it] = ok ? first + (uint32_t)pe : 0u;
          pok |= (ok ? 1u : 0u) << it;
        }
      } else {
        const uint32_t off = (uint32_t)(row - R_fill) * FU_C;
        const uint32_t my_n = (own_beg >= 0 && own_nmain > off) ? min(own_nmain - off, (uint32_t)FU_C) : 0u;
        const uint32_t my_p0 = own_e0 + off;
        const uint32_t my_bits = __funnelshift_r(own_w0, own_w1, my_p0 & 31u) & ((1u << FU_C) - 1u);
        const uint32_t my_ia = own_ia;
        const uint32_t my_bn = my_bits | (my_n << 16);
        const uint32_t my_jd = my_p0 - my_ia;                          // drops behind the stream = the next record to drop
        own_ia = my_ia + (uint32_t)__popc(my_bits);
        const uint32_t below = (1u << pe) - 1u;
#pragma unroll
        for (int it = 0; it < FU_ITERS; it++) {
          const int src = pu * FU_C + it;
          const uint32_t bn = (uint32_t)__shfl((int)my_bn, src), ia = (uint32_t)__shfl((int)my_ia, src), jd = (uint32_t)__shfl((int)my_jd, src);
          const bool ok = (uint32_t)pe < (bn >> 16);
          const uint32_t before = (uint32_t)__popc(bn & below);
          const uint32_t admit = (bn >> pe) & 1u;
          const uint32_t rec = admit ? ia + before : jd + ((uint32_t)pe - before);
          prec[it] = ok ? rec : 0u;
          pok |= (ok ? 1u : 0u) << it;
          padmit |= admit << it;
        }
      }
      // the loads of the row back to back and unconditional (lanes without an event read record 0): a predicated load
      // into a register that an earlier load may still be writing makes the compiler wait for that load first, which
      // would serialise the sixteen round trips
      if (a.dbg & 4) return;
#pragma unroll
      for (int it = 0; it < FU_ITERS; it++) phf[it] = a.ix.rec_hf[prec[it]];
    };
    auto stage_c = [&](int row) __attribute__((always_inline)) {
      if (row >= R_all || (a.dbg & 2)) return;
      T *out = ring + (size_t)(row & 1) * FU_C * 64 + (size_t)pe * 64 + ubase + pu;
      const bool fill = row < R_fill;
      const uint32_t k = (uint32_t)(row * FU_C + pe);
      constexpr int HALF = FU_ITERS > 8 ? FU_ITERS / 2 : FU_ITERS;     // lookups in flight together
#pragma unroll
      for (int hf = 0; hf < FU_ITERS / HALF; hf++) {
        // ranks of HALF hashes at once.  rank = first rank of the bucket + the entries of the bucket below the hash: the
        // bucket start, then FU_PROBE sketch entries read together (entries past the bucket are larger anyway, the
        // sketch being sorted; past the sketch sit sentinels) -- two dependent LDS round trips, HALF of them in flight.
        // Buckets wider than FU_PROBE (n_steps > 0, rare) get further probes, the same number in every lane.
        int x[HALF];
        bool eq[HALF];
#pragma unroll
        for (int q = 0; q < HALF; q++) x[q] = QT[bucket_of(phf[hf * HALF + q].x)];
#pragma unroll
        for (int q = 0; q < HALF; q++) {
          const uint32_t h = phf[hf * HALF + q].x;
          const uint32_t *e = Q + x[q];
          uint32_t v[FU_PROBE];
#pragma unroll
          for (int j = 0; j < FU_PROBE; j++) v[j] = e[j];
          int below = 0;
          bool hit = false;
#pragma unroll
          for (int j = 0; j < FU_PROBE; j++) { below += v[j] < h ? 1 : 0; hit = hit || v[j] == h; }
          x[q] += below; eq[q] = hit;
        }
        for (int stp = 0; stp < n_steps; stp++) {
#pragma unroll
          for (int q = 0; q < HALF; q++) {
            // continues only where all FU_PROBE entries were below the hash (x advanced by a full probe each time)
            const uint32_t h = phf[hf * HALF + q].x;
            const uint32_t *e = Q + x[q];
            // (where the previous probe stopped short, Q[x] >= h already and nothing changes)
            int below = 0;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < FU_PROBE; j++) { const uint32_t vj = e[j]; below += vj < h ? 1 : 0; hit = hit || vj == h; }
            x[q] += below; eq[q] = eq[q] || hit;
          }
        }
#pragma unroll
        for (int q = 0; q < HALF; q++) {
          const int it = hf * HALF + q;
          const uint32_t y_ = phf[it].y;
          const bool found = eq[q] && x[q] < s;                        // (an equal entry at rank >= s is a sentinel)
          const uint32_t base = (uint32_t)(x[q] + 1) << EV_SLOT;
          const int dsh = found ? EV_DM : EV_DW;
          const uint32_t dist = (y_ >> 8) & 0xFFFFu;
          // fill: a no-op admit when the hash is already in the window (its previous occurrence lies at or after `beg`)
          const uint32_t ev_fill = base | ((dist <= k ? 0u : 1u) << dsh) | ev_noeval<T>();
          // admit: after the drops of all records before the one active at its window position; carries the comparison;
          // a no-op when linked to the previous record of the same hash
          const uint32_t ev_admit = base | (((y_ & FLAG_INS_LINKED) ? 0u : 1u) << dsh);
          // drop at window position wpos[i+1], before the admit of that same position (FLAG_SAME_STEP), which then
          // carries the comparison
          const uint32_t same = (y_ & FLAG_SAME_STEP) ? 1u : 0u;
          const uint32_t ev_drop = base | (((y_ & FLAG_DEL_LINKED) ? 0u : 3u) << dsh) | (1u << EV_DROP) | (same ? ev_noeval<T>() : 0u);
          uint32_t ev = fill ? ev_fill : (((padmit >> it) & 1u) ? ev_admit : ev_drop);
          ev = ((pok >> it) & 1u) ? ev : ev_noeval<T>();
          out[it * FU_UPI] = (T)ev;
        }
      }
    };

    // slider state (wave 0 only)
    Slide<T, ST, 64, false> sl;
    sl.init(st, lane, 64, s, 0);
    if (wv == 0) {
      sl.beg = sl.opt_s = sl.opt_e = my_beg;
    } else {
      own_beg = owner ? u_beg[own] : -1; own_e0 = u_e0[own]; own_nmain = u_nmain[own];
      own_nfill = own_beg >= 0 ? own_e0 - 2u * (uint32_t)own_beg : 0u;  // e0 = beg + end0
      own_ia = own_e0 - (uint32_t)max(own_beg, 0);                     // = end0: the records of the first window count as admitted
    }
    __syncthreads();                                                   // the hand-over has been read: the ring is the ring again
    if (wv != 0) { stage_a(0); stage_b(0); stage_a(1); stage_c(0); stage_b(1); stage_a(2); }
    lds_barrier();
    unsigned long long t_work = 0;                                     // FA_FUSED_DEBUG & 8: cycles between the barriers
    for (int row = 0; row < R_all; row++) {
      const unsigned long long t_in = stamp ? __builtin_amdgcn_s_memtime() : 0ULL;
      if (wv != 0) {
        stage_c(row + 1); stage_b(row + 2); stage_a(row + 3);
      } else if (!(a.dbg & 1)) {
        const T *in = ring + (size_t)(row & 1) * FU_C * 64 + lane;
        uint32_t word[FU_C];
#pragma unroll
        for (int q = 0; q < FU_C; q++) word[q] = (uint32_t)in[q * 64];
        if (row < R_fill) {
          // the first super-window: its admits only change the per-rank state
#pragma unroll
          for (int q = 0; q < FU_C; q++) sl.template fill<0>(word[q]);
          if (row == R_fill - 1) sl.read_pivot();       // incl. the comparison after the last admit of the first window
        } else {
#pragma unroll
          for (int q = 0; q < FU_C; q++) sl.template step<0>(word[q]);
        }
      }
      if (stamp) t_work += __builtin_amdgcn_s_memtime() - t_in;
      lds_barrier();
    }
    if (stamp && lane == 0) atomicAdd(&stamp[6 + (wv ? 1 : 0)], t_work);
    if (wv == 0) {
      if (my_locus >= 0) {
        const int32_t l = my_locus;
        if ((sl.overflow >> SBITS) && !REDO) {
          a.l_redo[l] = 1; atomicAdd(a.redo_count, 1u);
        } else {
          a.l_shared[l] = sl.best < 0 ? 0 : sl.best;
          a.l_pos[l] = (wpos[sl.opt_s] + wpos[sl.opt_e]) / 2;
          if (sl.best >= a.pass_lut[s]) {
            unsigned long long key = ((unsigned long long)(uint32_t)sl.best << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)l);
            atomicMax(&a.group_best[a.l_group[l]], key);
          }
        }
      }
    }
  }
  if (stamp && tid == 0) { stamp[2] = __builtin_amdgcn_s_memrealtime(); atomicSub((unsigned int *)((unsigned long long *)a.items + (size_t)gridDim.x * 8), 1u); }
}


}  // namespace fa
