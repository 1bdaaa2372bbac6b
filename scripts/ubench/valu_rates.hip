// Micro-benchmark (gfx950): what one wave64 VALU instruction costs its SIMD, per opcode, at 1 / 2 / 4 / 8 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates > profiles/r03_valu_rates.txt
//
// Every kernel runs ITERS x 8 copies of ONE opcode (inline asm, eight independent register chains, so a wave never waits
// for its own result) between two reads of the shader clock (s_memtime) and of the constant 100 MHz counter
// (s_memrealtime); a dependent variant (one chain) shows the issue-to-issue latency of a wave working alone.  A launch
// puts exactly n workgroups of 4 waves on every CU (the LDS request is sized so that n fit and n + 1 do not), i.e. n waves
// per SIMD.  Reported per opcode and n:
//   cadence  = shader cycles between two instructions of ONE wave          (what a wave sees)
//   slot     = cadence / n = SIMD cycles per wave-instruction              (what the SIMD pays: the roofline unit)
// MI355X_MICROARCH.md: a wave64 VALU instruction occupies the SIMD-32 for 2 cycles; one wave alone issues every 4.
// The shader clock is measured, not assumed: cycles per 100 MHz tick over the timed loop (printed per run).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;

enum Op { ADD, XOR, MINU, AND_OR, BFE_I, ALIGNBIT, PERM, CNDMASK, CMP_CND, MAD_I24, MUL_U24, MUL_LO, MUL_HI, MAD_U64, LSHL_ADD_U64, ADD64, DEP_ADD, DEP_MUL_LO, DEP_MAD_U64, NOPS };
static const char *kNames[NOPS] = {"v_add_u32", "v_xor_b32", "v_min_u32", "v_and_or_b32", "v_bfe_i32", "v_alignbit_b32", "v_perm_b32", "v_cndmask_b32",
                                   "v_cmp_lt_u32+v_cndmask (2)", "v_mad_i32_i24", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32",
                                   "v_lshl_add_u64", "v_add_co+v_addc_co (2)", "dependent v_add_u32", "dependent v_mul_lo_u32", "dependent v_mad_u64_u32"};
static const int kInstr[NOPS] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *ticks, uint32_t seed) {
  extern __shared__ unsigned char lds[];
  uint32_t a[8];
  uint64_t b[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * (2 * i + 3) + seed + i; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u + 1u); }
  uint32_t c = seed * 0x9E3779B1u | 1u, d = threadIdx.x | 0x01020304u;
  if (threadIdx.x == 0xFFFF) lds[0] = 1;                             // (keeps the LDS request alive)
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // ONE asm statement per trip: between separate statements the compiler pads with s_nop, which costs issue slots
#define OUT8A "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define OUT8B "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
#define R8(L) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
#define ASM_A(L) asm volatile(R8(L) : OUT8A : "v"(c), "v"(d) : "vcc")
#define ASM_B(L) asm volatile(R8(L) : OUT8B : "v"(c), "v"(d) : "vcc")
#define L_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define L_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define L_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n\t"
#define L_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define L_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 1, 30\n\t"
#define L_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n\t"
#define L_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define L_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define L_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define L_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n\t"
#define L_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define L_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define L_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define L_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
#define L_LSHLADD(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %" #i "\n\t"
#define L_DADD(i) "v_add_u32 %0, %0, %8\n\t"
#define L_DMULLO(i) "v_mul_lo_u32 %0, %0, %8\n\t"
#define L_DMAD64(i) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
#define L_ADD64(i) "v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %4, vcc, %4, %9, vcc\n\tv_add_co_u32 %1, vcc, %1, %8\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\t"
  for (int it = 0; it < ITERS; it++) {
    if (OP == ADD) ASM_A(L_ADD);
    else if (OP == XOR) ASM_A(L_XOR);
    else if (OP == MINU) ASM_A(L_MIN);
    else if (OP == AND_OR) ASM_A(L_ANDOR);
    else if (OP == BFE_I) ASM_A(L_BFE);
    else if (OP == ALIGNBIT) ASM_A(L_ALIGN);
    else if (OP == PERM) ASM_A(L_PERM);
    else if (OP == CNDMASK) ASM_A(L_CND);
    else if (OP == CMP_CND) ASM_A(L_CMPCND);
    else if (OP == MAD_I24) ASM_A(L_MAD24);
    else if (OP == MUL_U24) ASM_A(L_MUL24);
    else if (OP == MUL_LO) ASM_A(L_MULLO);
    else if (OP == MUL_HI) ASM_A(L_MULHI);
    else if (OP == MAD_U64) ASM_B(L_MAD64);
    else if (OP == LSHL_ADD_U64) ASM_B(L_LSHLADD);
    else if (OP == ADD64) asm volatile(L_ADD64(0) L_ADD64(0) L_ADD64(0) L_ADD64(0) : OUT8A : "v"(c), "v"(d) : "vcc");   // 8 pairs
    else if (OP == DEP_ADD) ASM_A(L_DADD);
    else if (OP == DEP_MUL_LO) ASM_A(L_DMULLO);
    else if (OP == DEP_MAD_U64) ASM_B(L_DMAD64);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  for (int i = 0; i < 8; i++) acc ^= a[i] ^ (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    ticks[2 * wv] = t1 - t0; ticks[2 * wv + 1] = r1 - r0;
  }
}

static uint32_t *g_out;
static unsigned long long *g_ticks;
static int g_cus = 256;

template <int OP> int run() {
  printf("%-28s", kNames[OP]);
  for (int n : {1, 2, 4, 8}) {
    const int lds = n == 1 ? 100 * 1024 : n == 2 ? 60 * 1024 : n == 4 ? 36 * 1024 : 19 * 1024;   // n workgroups fit a CU, n + 1 do not
    const int blocks = g_cus * n;
    CHECK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, g_out, g_ticks, 1u);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, g_out, g_ticks, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CHECK(hipMemcpy(h.data(), g_ticks, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, mhz;
    for (size_t w = 0; w < (size_t)blocks * 4; w++) { cyc.push_back((double)h[2 * w]); if (h[2 * w + 1]) mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double med = cyc[cyc.size() / 2], per = med / ((double)ITERS * 8 * kInstr[OP]);
    printf(" | n=%d cadence %6.2f slot %5.2f (%4.0f MHz, %.0f us)", n, per, per / n, mhz.empty() ? 0.0 : mhz[mhz.size() / 2], ms * 1e3);
  }
  printf("\n");
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("# %s, %d CUs, clockRate %d kHz, wall clock %d kHz; ITERS %d x 8 instructions per wave between the clock reads\n", prop.name, g_cus, prop.clockRate, wall_khz, ITERS);
  printf("# cadence = shader cycles (s_memtime) between two instructions of one wave; slot = cadence / n = SIMD cycles per wave-instruction;\n");
  printf("# MHz = s_memtime ticks per 100 MHz s_memrealtime tick over the loop (the clock the loop actually ran at); us = the whole launch by HIP events\n");
  CHECK(hipMalloc(&g_out, (size_t)g_cus * 8 * 256 * 4));
  CHECK(hipMalloc(&g_ticks, (size_t)g_cus * 8 * 4 * 16));
  run<ADD>(); run<XOR>(); run<MINU>(); run<AND_OR>(); run<BFE_I>(); run<ALIGNBIT>(); run<PERM>(); run<CNDMASK>(); run<CMP_CND>(); run<MAD_I24>();
  run<MUL_U24>(); run<MUL_LO>(); run<MUL_HI>(); run<MAD_U64>(); run<LSHL_ADD_U64>(); run<ADD64>(); run<DEP_ADD>(); run<DEP_MUL_LO>(); run<DEP_MAD_U64>();
  return 0;
}
