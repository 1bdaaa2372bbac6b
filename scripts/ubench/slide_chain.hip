// Micro-benchmark (gfx950): the sequential slide of k_l2_scan -- the PRODUCT kernel, not a model of it -- on synthetic
// event streams, at chosen numbers of waves per SIMD.  Answers what a per-lane dependent chain of the shape
// "two LDS reads -> ~40 dependent VALU -> one LDS write" per event costs when a wave has its SIMD to itself, when two
// share it, and at the bench step's 1.4 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o slide_chain slide_chain.hip && ./slide_chain >> profiles/r03_valu_rates.txt
//
// Every lane slides its own stream (1024 distinct streams, reused across waves): 240 first-window admits applied in
// bulk, then EV pivot events alternating admits and drops of a FIFO window, generated on the host so that the
// per-rank state stays valid (counts never negative, a matched rank admitted once) -- the branch-free loop executes the same
// instructions per event whatever the data, so what matters is only that the LDS addresses are spread as in the real
// thing.  The shader clock comes from a calibration kernel run right before (s_memtime per s_memrealtime tick).
#include <algorithm>
#include <cstdio>
#include <deque>
#include <random>
#include <vector>

#include "../../pyfastani_amd/csrc/fa_map.hip.h"

using namespace fa;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_clock(unsigned long long *out, int spin) {
  uint32_t a = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < spin; i++) asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0" : "+v"(a));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (a == 0x12345) out[0] = a;
}

int main(int argc, char **argv) {
  // argv[2]: sketch size (default 240).  A smaller sketch is a smaller LDS state per wave: what more waves per SIMD would buy the
  // UNCHANGED instruction chain (round 6: the bound on any scheme that shrinks the state, e.g. four bits per rank)
  const int S = argc > 2 ? atoi(argv[2]) : 240, FILL = 240, EV = argc > 1 ? atoi(argv[1]) : 976, NSTREAM = 1024;
  const int nev = FILL + EV;                                         // both multiples of 8
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, simds = cus * 4;
  // ---- synthetic streams ----
  std::mt19937 rng(12345);
  std::vector<uint16_t> items((size_t)NSTREAM * nev);
  for (int st = 0; st < NSTREAM; st++) {
    uint16_t *e = items.data() + (size_t)st * nev;
    std::vector<char> matched(S + 1, 0);
    struct Rec { int rank; bool m; bool noop; };
    std::deque<Rec> win;
    auto admit = [&](bool first, bool last_of_first) {
      Rec r; r.rank = (int)(rng() % (S + 1)); r.m = r.rank < S && (rng() % 100) < 60; r.noop = false;
      if (r.m && matched[r.rank]) r.noop = true;                      // the hash is in the window already: linked, a no-op
      if (r.m && !r.noop) matched[r.rank] = 1;
      win.push_back(r);
      const uint32_t base = (uint32_t)(r.rank + 1) << EV_SLOT;
      const int dsh = r.m ? EV_DM : EV_DW;
      uint32_t w = base | ((r.noop ? 0u : 1u) << dsh);
      if (first && !last_of_first) w |= ev_noeval<uint16_t>();
      if (!first && (rng() & 1)) w |= 0;                             // an admit carries the comparison of its window position
      return (uint16_t)w;
    };
    auto drop = [&]() {
      Rec r = win.front(); win.pop_front();
      if (r.m && !r.noop) matched[r.rank] = 0;
      const uint32_t base = (uint32_t)(r.rank + 1) << EV_SLOT;
      const int dsh = r.m ? EV_DM : EV_DW;
      const bool same = (rng() % 100) < 45;                          // followed by an admit at the same window position
      return (uint16_t)(base | ((r.noop ? 0u : 3u) << dsh) | (1u << EV_DROP) | (same ? ev_noeval<uint16_t>() : 0u));
    };
    for (int i = 0; i < FILL; i++) e[i] = admit(true, i == FILL - 1);
    for (int i = 0; i < EV; i++) e[FILL + i] = (i & 1) ? admit(false, false) : drop();
  }
  // ---- the arguments k_l2_scan reads ----
  const int maxw = 8 * simds;
  const uint32_t max_loci = (uint32_t)maxw * 64;
  uint16_t *d_items; CHECK(hipMalloc(&d_items, items.size() * 2)); CHECK(hipMemcpy(d_items, items.data(), items.size() * 2, hipMemcpyHostToDevice));
  std::vector<int32_t> h_zero(max_loci, 0), h_end0(max_loci, FILL), h_wpos(8192);
  std::vector<uint32_t> h_nev(max_loci, (uint32_t)nev), h_ioff(max_loci);
  for (uint32_t l = 0; l < max_loci; l++) h_ioff[l] = (uint32_t)((l % NSTREAM) * (size_t)nev);
  for (int i = 0; i < 8192; i++) h_wpos[i] = 13 * i;
  std::vector<int32_t> h_pass(S + 2, 1 << 30);
  auto dev_i32 = [&](const std::vector<int32_t> &v) { int32_t *p = nullptr; hipMalloc(&p, v.size() * 4); hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
  auto dev_u32 = [&](const std::vector<uint32_t> &v) { uint32_t *p = nullptr; hipMalloc(&p, v.size() * 4); hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
  L2Args a{};
  a.l_frag = dev_i32(h_zero); a.l_beg = dev_i32(h_zero); a.l_end0 = dev_i32(h_end0); a.l_group = dev_i32(h_zero);
  a.l_shared = dev_i32(h_zero); a.l_pos = dev_i32(h_zero);
  a.l_nev = dev_u32(h_nev); a.l_ioff = dev_u32(h_ioff);
  a.ix.rec_wpos = dev_i32(h_wpos);
  a.pass_lut = dev_i32(h_pass);
  int32_t qs = S; int32_t *d_qs; CHECK(hipMalloc(&d_qs, 4)); CHECK(hipMemcpy(d_qs, &qs, 4, hipMemcpyHostToDevice)); a.q_size = d_qs;
  uint32_t *d_counters; CHECK(hipMalloc(&d_counters, 32)); a.counters = d_counters; a.redo_count = d_counters + 3;
  // one region that holds every locus of a launch (LociRegions: workgroup b takes loci [64 b, 64 b + 64))
  uint32_t *d_live; CHECK(hipMalloc(&d_live, 4)); a.loci.count = d_live; a.loci.n = 1; a.loci.shift = 24;
  uint8_t *d_redo; CHECK(hipMalloc(&d_redo, max_loci)); a.l_redo = d_redo;
  unsigned long long *d_gb; CHECK(hipMalloc(&d_gb, 64)); CHECK(hipMemset(d_gb, 0, 64)); a.group_best = d_gb;
  a.items = d_items; a.cnt_slots = S + 17; a.lanes = 64; a.qcap = 0; a.cmw = 0;
  const size_t state = ((size_t)(a.cnt_slots + 1) * 64 + 15) / 16 * 16;
  unsigned long long *d_clk; CHECK(hipMalloc(&d_clk, (size_t)cus * 16));
  auto kernel = k_l2_scan<uint16_t, uint8_t, 64>;
  CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  printf("# k_l2_scan<uint16_t, uint8_t, 64> on synthetic streams: sketch %d, %d first-window events + %d pivot events per locus, 64 loci per wave; state %zu B of LDS per wave\n", S, FILL, EV, state);
  printf("# %-34s %9s %9s %12s %14s %16s\n", "launch", "waves", "us", "clock MHz", "ns per event", "cycles per event");
  struct Cfg { const char *name; int waves; int per_cu; };
  const Cfg cfgs[] = {{"1 wave per SIMD (4 per CU fit)", simds, 4}, {"2 waves per SIMD (8 per CU fit)", 2 * simds, 8}, {"bench-like 1428 waves (8 fit)", 1428, 8},
                      {"1024 waves, 8 per CU fit", simds, 8}, {"3 waves per SIMD (12 per CU fit... LDS 8)", 3 * simds, 8}, {"4 waves per SIMD in two rounds", 4 * simds, 8},
                      // (per_cu 0: the LDS request is the state alone -- with a small sketch as many waves per SIMD as the state allows)
                      {"state-limited: 3 waves per SIMD", 3 * simds, 0}, {"state-limited: 4 waves per SIMD", 4 * simds, 0}, {"state-limited: 6 waves per SIMD", 6 * simds, 0}};
  for (const Cfg &c : cfgs) {
    const size_t lds = c.per_cu == 0 ? state : std::max(state, (size_t)(c.per_cu == 4 ? 36 * 1024 : 19 * 1024));
    const uint32_t loci = (uint32_t)c.waves * 64;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CHECK(hipMemcpy(d_counters, cnt, 32, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_live, &loci, 4, hipMemcpyHostToDevice));
    // clock under a comparable load
    hipLaunchKernelGGL(k_clock, dim3(cus), dim3(256), 0, 0, d_clk, 20000);
    std::vector<unsigned long long> hc((size_t)cus * 2);
    CHECK(hipMemcpy(hc.data(), d_clk, hc.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz; for (int i = 0; i < cus; i++) if (hc[2 * i + 1]) mhz.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double f = mhz.empty() ? 0 : mhz[mhz.size() / 2];
    float best = 1e9f;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(c.waves), dim3(64), lds, 0, a);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    CHECK(hipGetLastError());
    const double us = best * 1e3, per_ev_ns = us * 1e3 / EV;
    printf("  %-34s %9d %9.1f %12.0f %14.1f %16.1f\n", c.name, c.waves, us, f, per_ev_ns, per_ev_ns * f * 1e-3);
  }
  {
    // (results of the last launch: the same checksum from every variant of the slide built into this binary's kernel)
    std::vector<int32_t> hs(4096), hp(4096);
    CHECK(hipMemcpy(hs.data(), a.l_shared, hs.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hp.data(), a.l_pos, hp.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (size_t i = 0; i < hs.size(); i++) sum = sum * 1000003ULL + (unsigned long long)(uint32_t)hs[i] * 31ULL + (unsigned long long)(uint32_t)hp[i];
    printf("# checksum of l_shared / l_pos over the first 4096 loci: %016llx\n", sum);
  }
  printf("# ns / cycles per event divide the launch time by the %d pivot events of a lane (the %d bulk admits and the pivot read-off ride along: ~15 %% of the time);\n", EV, FILL);
  printf("# at two waves per SIMD a SIMD executes two events in that time.\n");
  return 0;
}
