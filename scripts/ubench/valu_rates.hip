// Micro-benchmark (gfx950): what one wave64 VALU instruction costs its SIMD, per opcode, at 1 / 2 / 4 / 8 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates > profiles/r03_valu_rates.txt
//
// Every kernel runs ITERS trips of ONE asm statement holding eight copies of an opcode on eight independent register
// chains (a wave never waits for its own result; separate asm statements would be padded with s_nop by the compiler),
// between two reads of the shader clock (s_memtime) and of the chip-wide 100 MHz counter (s_memrealtime).  A launch asks
// for n workgroups of 4 waves per CU (grid = CUs x n, LDS request sized so that n fit a CU and n + 1 do not), i.e. n waves
// per SIMD.  Reported per opcode and n:
//   cad   = shader cycles between two instructions of ONE wave (median over the waves)
//   slot  = SIMD cycles per wave-instruction = (last wave's end - first wave's start) x clock / (n x instructions per wave):
//           what the SIMD pays, the unit of a VALU roofline.  (Not cad / n: the waves of a launch do not all overlap.)
//   MHz   = s_memtime ticks per s_memrealtime tick x 100: the clock the loop actually ran at (it drops under VALU load).
// The last three rows are one dependent chain per wave: the issue-to-issue latency a lone wave sees.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef ITERS
#define ITERS 4096
#endif

// id, name, register kind (A: 32-bit chains, B: 64-bit chains), asm line macro, wave-instructions per trip
#define OPS(X)                                                      \
  X(ADD, "v_add_u32", A, L_ADD, 8)                                  \
  X(ADD_E64, "v_add_u32_e64 (VOP3 form)", A, L_ADD64E, 8)           \
  X(SUB, "v_sub_u32", A, L_SUB, 8)                                  \
  X(XOR, "v_xor_b32", A, L_XOR, 8)                                  \
  X(AND, "v_and_b32", A, L_AND, 8)                                  \
  X(OR, "v_or_b32", A, L_OR, 8)                                     \
  X(MOV, "v_mov_b32", A, L_MOV, 8)                                  \
  X(LSHL, "v_lshlrev_b32", A, L_LSHL, 8)                            \
  X(LSHR, "v_lshrrev_b32", A, L_LSHR, 8)                            \
  X(ASHR, "v_ashrrev_i32", A, L_ASHR, 8)                            \
  X(MINU, "v_min_u32", A, L_MIN, 8)                                 \
  X(MAXI, "v_max_i32", A, L_MAX, 8)                                 \
  X(MUL_I24, "v_mul_i32_i24", A, L_MULI24, 8)                       \
  X(MUL_U24, "v_mul_u32_u24", A, L_MUL24, 8)                        \
  X(CMP, "v_cmp_lt_u32 (vcc)", A, L_CMP, 8)                         \
  X(CNDMASK, "v_cndmask_b32 (stale vcc)", A, L_CND, 8)              \
  X(CMP_CND, "v_cmp + v_cndmask", A, L_CMPCND, 16)                  \
  X(CMP1_CND8, "v_cmp + 8 v_cndmask (vcc)", A, L_CND, 9)            \
  X(CMPS_CND8, "v_cmp_e64 + 8 v_cndmask_e64 (sgpr)", A, L_CNDS, 9)  \
  X(CMP_X_CND, "v_cmp, v_add, v_cndmask (vcc)", A, L_ADD, 12)       \
  X(CMP_CND2, "v_cmp, 2 v_cndmask (vcc)", A, L_ADD, 12)             \
  X(CMP_CND2_E64, "v_cmp, 2 v_cndmask_e64 (vcc)", A, L_ADD, 12)     \
  X(CMPS_CND2, "v_cmp_e64, 2 v_cndmask_e64 (sgpr)", A, L_ADD, 12)   \
  X(CMPS_X_CND2, "v_cmp_e64, v_add, 2 v_cndmask_e64 (sgpr)", A, L_ADD, 16) \
  X(ARITH_SEL, "v_sub, v_ashrrev, v_and (select)", A, L_ADD, 12)    \
  X(ADD3, "v_add3_u32", A, L_ADD3, 8)                               \
  X(LSHL_ADD, "v_lshl_add_u32", A, L_LSHLADD32, 8)                  \
  X(LSHL_OR, "v_lshl_or_b32", A, L_LSHLOR, 8)                       \
  X(XAD, "v_xad_u32", A, L_XAD, 8)                                  \
  X(OR3, "v_or3_b32", A, L_OR3, 8)                                  \
  X(AND_OR, "v_and_or_b32", A, L_ANDOR, 8)                          \
  X(BFE_U, "v_bfe_u32", A, L_BFEU, 8)                               \
  X(BFE_I, "v_bfe_i32", A, L_BFE, 8)                                \
  X(ALIGNBIT, "v_alignbit_b32", A, L_ALIGN, 8)                      \
  X(PERM, "v_perm_b32", A, L_PERM, 8)                               \
  X(MAD_I24, "v_mad_i32_i24", A, L_MAD24, 8)                        \
  X(MUL_LO, "v_mul_lo_u32", A, L_MULLO, 8)                          \
  X(MUL_HI, "v_mul_hi_u32", A, L_MULHI, 8)                          \
  X(MAD_U64, "v_mad_u64_u32", B, L_MAD64, 8)                        \
  X(LSHL_ADD_U64, "v_lshl_add_u64", B, L_LSHLADD, 8)                \
  X(LSHLREV_B64, "v_lshlrev_b64", B, L_LSHL64, 8)                   \
  X(ADDC, "v_add_co_u32 + v_addc_co_u32", A, L_ADD, 16)             \
  X(DS_READ, "ds_read_b32 (8 in flight)", A, L_ADD, 8)              \
  X(DS_READ64, "ds_read_b64 (8 in flight)", B, L_MAD64, 8)          \
  X(DS_RW8, "ds_read_u8 + ds_write_b8", A, L_ADD, 16)               \
  X(DEP_ADD, "dependent v_add_u32", A, L_DADD, 8)                   \
  X(DEP_MUL_LO, "dependent v_mul_lo_u32", A, L_DMULLO, 8)           \
  X(DEP_MAD_U64, "dependent v_mad_u64_u32", B, L_DMAD64, 8)

enum Op {
#define X(id, name, kind, line, n) id,
  OPS(X)
#undef X
  NOPS
};

#define OUT8A "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define OUT8B "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
#define INS "v"(c), "v"(d), "v"(la)
#define CLOB "vcc", "s20", "s21", "memory"
#define R8(L) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
#define ASM_A(L) asm volatile(R8(L) : OUT8A : INS : CLOB)
#define ASM_B(L) asm volatile(R8(L) : OUT8B : INS : CLOB)
#define L_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define L_ADD64E(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n\t"
#define L_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n\t"
#define L_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define L_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define L_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n\t"
#define L_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define L_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n\t"
#define L_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n\t"
#define L_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n\t"
#define L_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n\t"
#define L_MAX(i) "v_max_i32 %" #i ", %" #i ", %8\n\t"
#define L_MULI24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n\t"
#define L_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define L_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\t"
#define L_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define L_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define L_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define L_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define L_LSHLADD32(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n\t"
#define L_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 2, %8\n\t"
#define L_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n\t"
#define L_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n\t"
#define L_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define L_BFEU(i) "v_bfe_u32 %" #i ", %" #i ", 1, 30\n\t"
#define L_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 1, 30\n\t"
#define L_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n\t"
#define L_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define L_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n\t"
#define L_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define L_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define L_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
#define L_LSHLADD(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %" #i "\n\t"
#define L_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n\t"
#define L_ADDC_(x, y) "v_add_co_u32 %" #x ", vcc, %" #x ", %8\n\tv_addc_co_u32 %" #y ", vcc, %" #y ", %9, vcc\n\t"
#define L_DADD(i) "v_add_u32 %0, %0, %8\n\t"
#define L_DMULLO(i) "v_mul_lo_u32 %0, %0, %8\n\t"
#define L_DMAD64(i) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *ticks, uint32_t seed) {
  extern __shared__ unsigned char lds[];
  uint32_t a[8];
  uint64_t b[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * (2 * i + 3) + seed + i; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u + 1u); }
  uint32_t c = seed * 0x9E3779B1u | 1u, d = threadIdx.x | 0x01020304u;
  const uint32_t la = (threadIdx.x * 8u) & 0x1FFu;                   // LDS address of the ds_ rows (one 8-byte column per lane of a wave)
  for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)lds)[i] = i * seed;
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_mov_b32 s20, 0x33333333\n\ts_mov_b32 s21, 0x33333333" ::: "vcc", "s20", "s21");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    if (OP == CMP1_CND8) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\t" R8(L_CND) : OUT8A : INS : CLOB);
    else if (OP == CMPS_CND8) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n\t" R8(L_CNDS) : OUT8A : INS : CLOB);
#define G4(L) L(0, 4) L(1, 5) L(2, 6) L(3, 7)
#define L_CXC(x, y) "v_cmp_lt_u32 vcc, %" #x ", %8\n\tv_add_u32 %" #y ", %" #y ", %9\n\tv_cndmask_b32 %" #x ", %" #x ", %9, vcc\n\t"
#define L_CC2(x, y) "v_cmp_lt_u32 vcc, %" #x ", %8\n\tv_cndmask_b32 %" #x ", %" #x ", %9, vcc\n\tv_cndmask_b32 %" #y ", %" #y ", %8, vcc\n\t"
#define L_CC2E(x, y) "v_cmp_lt_u32 vcc, %" #x ", %8\n\tv_cndmask_b32_e64 %" #x ", %" #x ", %9, vcc\n\tv_cndmask_b32_e64 %" #y ", %" #y ", %8, vcc\n\t"
#define L_CS2(x, y) "v_cmp_lt_u32_e64 s[20:21], %" #x ", %8\n\tv_cndmask_b32_e64 %" #x ", %" #x ", %9, s[20:21]\n\tv_cndmask_b32_e64 %" #y ", %" #y ", %8, s[20:21]\n\t"
#define L_CSX2(x, y) "v_cmp_lt_u32_e64 s[20:21], %" #x ", %8\n\tv_add_u32 %" #y ", %" #y ", %9\n\tv_cndmask_b32_e64 %" #x ", %" #x ", %9, s[20:21]\n\tv_cndmask_b32_e64 %" #y ", %" #y ", %8, s[20:21]\n\t"
#define L_ASEL(x, y) "v_sub_u32 %" #y ", %" #x ", %8\n\tv_ashrrev_i32 %" #y ", 31, %" #y "\n\tv_and_b32 %" #x ", %" #y ", %9\n\t"
    else if (OP == CMP_X_CND) asm volatile(G4(L_CXC) : OUT8A : INS : CLOB);
    else if (OP == CMP_CND2) asm volatile(G4(L_CC2) : OUT8A : INS : CLOB);
    else if (OP == CMP_CND2_E64) asm volatile(G4(L_CC2E) : OUT8A : INS : CLOB);
    else if (OP == CMPS_CND2) asm volatile(G4(L_CS2) : OUT8A : INS : CLOB);
    else if (OP == CMPS_X_CND2) asm volatile(G4(L_CSX2) : OUT8A : INS : CLOB);
    else if (OP == ARITH_SEL) asm volatile(G4(L_ASEL) : OUT8A : INS : CLOB);
    else if (OP == ADDC) asm volatile(L_ADDC_(0, 4) L_ADDC_(1, 5) L_ADDC_(2, 6) L_ADDC_(3, 7) L_ADDC_(0, 4) L_ADDC_(1, 5) L_ADDC_(2, 6) L_ADDC_(3, 7) : OUT8A : INS : CLOB);
    else if (OP == DS_READ) asm volatile("ds_read_b32 %0, %10\n\tds_read_b32 %1, %10 offset:512\n\tds_read_b32 %2, %10 offset:1024\n\tds_read_b32 %3, %10 offset:1536\n\t"
                                         "ds_read_b32 %4, %10 offset:2048\n\tds_read_b32 %5, %10 offset:2560\n\tds_read_b32 %6, %10 offset:3072\n\tds_read_b32 %7, %10 offset:3584\n\t"
                                         "s_waitcnt lgkmcnt(0)\n\t" : OUT8A : INS : CLOB);
    else if (OP == DS_READ64) asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %10 offset:512\n\tds_read_b64 %2, %10 offset:1024\n\tds_read_b64 %3, %10 offset:1536\n\t"
                                           "ds_read_b64 %4, %10 offset:2048\n\tds_read_b64 %5, %10 offset:2560\n\tds_read_b64 %6, %10 offset:3072\n\tds_read_b64 %7, %10 offset:3584\n\t"
                                           "s_waitcnt lgkmcnt(0)\n\t" : OUT8B : INS : CLOB);
    else if (OP == DS_RW8) asm volatile("ds_read_u8 %0, %10\n\tds_read_u8 %1, %10 offset:512\n\tds_read_u8 %2, %10 offset:1024\n\tds_read_u8 %3, %10 offset:1536\n\t"
                                        "ds_read_u8 %4, %10 offset:2048\n\tds_read_u8 %5, %10 offset:2560\n\tds_read_u8 %6, %10 offset:3072\n\tds_read_u8 %7, %10 offset:3584\n\t"
                                        "s_waitcnt lgkmcnt(0)\n\t"
                                        "ds_write_b8 %10, %0 offset:4096\n\tds_write_b8 %10, %1 offset:4608\n\tds_write_b8 %10, %2 offset:5120\n\tds_write_b8 %10, %3 offset:5632\n\t"
                                        "ds_write_b8 %10, %4 offset:6144\n\tds_write_b8 %10, %5 offset:6656\n\tds_write_b8 %10, %6 offset:7168\n\tds_write_b8 %10, %7 offset:7680\n\t"
                                        : OUT8A : INS : CLOB);
#define X(id, name, kind, line, n) else if (OP == id) ASM_##kind(line);
    OPS(X)
#undef X
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  for (int i = 0; i < 8; i++) acc ^= a[i] ^ (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    ticks[4 * wv] = t1 - t0; ticks[4 * wv + 1] = r1 - r0; ticks[4 * wv + 2] = r0; ticks[4 * wv + 3] = r1;
  }
}

static uint32_t *g_out;
static unsigned long long *g_ticks;
static int g_cus = 256;

template <int OP> int run(const char *name, int per_trip) {
  printf("%-36s", name);
  for (int n : {1, 2, 4, 8}) {
    const int lds = n == 1 ? 100 * 1024 : n == 2 ? 60 * 1024 : n == 4 ? 36 * 1024 : 19 * 1024;   // n workgroups fit a CU, n + 1 do not
    const int blocks = g_cus * n;
    CHECK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, g_out, g_ticks, 1u);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, g_out, g_ticks, 2u);
    CHECK(hipDeviceSynchronize());
    const size_t waves = (size_t)blocks * 4;
    std::vector<unsigned long long> h(waves * 4);
    CHECK(hipMemcpy(h.data(), g_ticks, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, mhz;
    unsigned long long first = ~0ULL, last = 0;
    for (size_t w = 0; w < waves; w++) {
      cyc.push_back((double)h[4 * w]);
      if (h[4 * w + 1]) mhz.push_back((double)h[4 * w] / (double)h[4 * w + 1] * 100.0);
      first = std::min(first, h[4 * w + 2]); last = std::max(last, h[4 * w + 3]);
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double f = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    const double instr = (double)ITERS * per_trip;
    const double cad = cyc[cyc.size() / 2] / instr, slot = (double)(last - first) * (f / 100.0) / (n * instr);
    printf(" | n=%d cad %5.2f slot %5.2f %4.0f MHz", n, cad, slot, f);
  }
  printf("\n");
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("# %s %s, %d CUs, clockRate %d kHz; %d trips of one asm statement per wave between the clock reads\n", prop.name, prop.gcnArchName, g_cus, prop.clockRate, (int)ITERS);
  printf("# n = waves per SIMD asked for; cad = shader cycles between two instructions of one wave (median); slot = SIMD cycles per wave-instruction\n");
  printf("# over the whole launch (first wave's start to last wave's end, 100 MHz chip-wide counter, at the measured clock)\n");
  CHECK(hipMalloc(&g_out, (size_t)g_cus * 8 * 256 * 4));
  CHECK(hipMalloc(&g_ticks, (size_t)g_cus * 8 * 4 * 32));
#define X(id, name, kind, line, n) if (run<id>(name, n)) return 1;
  OPS(X)
#undef X
  return 0;
}
