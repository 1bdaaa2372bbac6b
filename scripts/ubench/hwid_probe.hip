// Where do the waves of a launch of single-wave workgroups land?  (ScanOrder, fa_map.hip.h: k_l2_scan deals its batches
// by SIMD.)  hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip && ./hwid_probe [workgroups] [lds bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) void probe(unsigned *out, int spin) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
  }
  lds[threadIdx.x] = 1;
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
}
int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 2048, lds = argc > 2 ? atoi(argv[2]) : 18560;
  unsigned *d; hipMalloc(&d, n * 8);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(n), dim3(64), lds, 0, d, 30);
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  unsigned or_hw = 0, and_hw = ~0u, or_x = 0, and_x = ~0u;
  for (int i = 0; i < n; i++) { or_hw |= h[2 * i]; and_hw &= h[2 * i]; or_x |= h[2 * i + 1]; and_x &= h[2 * i + 1]; }
  printf("HW_ID bits that vary: %08x (always set %08x)  XCC_ID bits that vary: %08x (always set %08x)\n", or_hw & ~and_hw, and_hw, or_x & ~and_x, and_x);
  auto tally = [&](const char *name, unsigned mask_hw, unsigned mask_x) {
    std::map<unsigned long long, int> m;
    for (int i = 0; i < n; i++) m[((unsigned long long)(h[2 * i + 1] & mask_x) << 32) | (h[2 * i] & mask_hw)]++;
    std::map<int, int> hist; for (auto &kv : m) hist[kv.second]++;
    printf("%-28s distinct %5zu  waves-per-key histogram:", name, m.size());
    for (auto &kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
  };
  tally("simd|cu|sh|se + xcc[3:0]", 0xFF30, 0xF);
  tally("simd|pipe|cu|sh|se + xcc", 0xFFF0, 0xF);
  tally("cu|sh|se + xcc (per CU)", 0xFF00, 0xF);
  tally("simd|cu|sh|se, no xcc", 0xFF30, 0);
  tally("bits 4..15, xcc all", 0xFFF0, ~0u);
  for (int i = 0; i < 12; i++) printf("wg %4d hw %08x xcc %08x\n", i * 97 % n, h[2 * (i * 97 % n)], h[2 * (i * 97 % n) + 1]);
  // which workgroups share a SIMD
  std::map<unsigned long long, std::vector<int>> by;
  for (int i = 0; i < n; i++) by[((unsigned long long)(h[2 * i + 1] & 0xF) << 32) | (h[2 * i] & 0xFF30)].push_back(i);
  int shown = 0;
  for (auto &kv : by) { if (shown++ >= 6) break; printf("key %llx:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  return 0;
}
