// Micro-benchmark: issue rate of the integer VALU instructions the minimizer hash is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
    if (OP == 1) { a0 *= 0x9E3779B1u; a1 *= 0x85EBCA6Bu; a2 *= 0xC2B2AE35u; a3 *= 0x27D4EB2Fu; a4 *= 0x165667B1u; a5 *= 0x9E3779B1u; a6 *= 0x85EBCA6Bu; a7 *= 0xC2B2AE35u; }
    if (OP == 2) { a0 = __umulhi(a0, 0x9E3779B1u); a1 = __umulhi(a1, 0x85EBCA6Bu); a2 = __umulhi(a2, 0xC2B2AE35u); a3 = __umulhi(a3, 0x27D4EB2Fu); a4 = __umulhi(a4, 0x165667B1u); a5 = __umulhi(a5, 0x9E3779B1u); a6 = __umulhi(a6, 0x85EBCA6Bu); a7 = __umulhi(a7, 0xC2B2AE35u); }
    if (OP == 3) { b0 *= 0x87c37b91114253d5ULL; b1 *= 0x4cf5ad432745937fULL; b2 *= 0xff51afd7ed558ccdULL; b3 *= 0xc4ceb9fe1a85ec53ULL; }
    if (OP == 4) { a0 = __umul24(a0, 0x9E3779u); a1 = __umul24(a1, 0x85EBCAu); a2 = __umul24(a2, 0xC2B2AEu); a3 = __umul24(a3, 0x27D4EBu); a4 = __umul24(a4, 0x165667u); a5 = __umul24(a5, 0x9E3779u); a6 = __umul24(a6, 0x85EBCAu); a7 = __umul24(a7, 0xC2B2AEu); }
    if (OP == 5) { a0 = __builtin_amdgcn_perm(a0, a1, 0x03020100u ^ a2); a1 = __builtin_amdgcn_perm(a1, a2, a3); a2 = __builtin_amdgcn_perm(a2, a3, a4); a3 = __builtin_amdgcn_perm(a3, a4, a5); a4 = __builtin_amdgcn_perm(a4, a5, a6); a5 = __builtin_amdgcn_perm(a5, a6, a7); a6 = __builtin_amdgcn_perm(a6, a7, a0); a7 = __builtin_amdgcn_perm(a7, a0, a1); }
    if (OP == 6) { b0 = (b0 << 31) | (b0 >> 33); b1 = (b1 << 27) | (b1 >> 37); b2 = (b2 << 33) | (b2 >> 31); b3 = (b3 << 13) | (b3 >> 51); b0 ^= b1; b1 += b2; b2 ^= b3; b3 += b0; }
    if (OP == 7) { b0 ^= b0 >> 33; b1 ^= b1 >> 33; b2 ^= b2 >> 33; b3 ^= b3 >> 33; b0 += b1; b2 += b3; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3) ^ (uint32_t)((b0 ^ b1 ^ b2 ^ b3) >> 32);
}
template <int OP> int run(const char *name, int ops_per_iter, uint32_t *out) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_ops = (double)blocks * 4 * ITERS * ops_per_iter;           // wave-level source operations
  double per_simd = wave_ops / 1024.0;                                    // 256 CUs x 4 SIMDs
  printf("%-28s %8.3f ms  %.2f ns per wave-op per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  return 0;
}
int main() {
  uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
  run<0>("v_add_u32", 8, out);
  run<1>("v_mul_lo_u32", 8, out);
  run<2>("v_mul_hi_u32", 8, out);
  run<3>("u64 * const (64-bit low)", 4, out);
  run<4>("v_mul_u32_u24", 8, out);
  run<5>("v_perm_b32", 8, out);
  run<6>("rotl64 + xor/add u64", 8, out);
  run<7>("x ^= x >> 33; add u64", 6, out);
  return 0;
}
