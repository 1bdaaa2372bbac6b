// Same question, closer to k_l2_fused: static + dynamic LDS, big kernarg, launch bounds, barriers, LDS traffic, s_setprio.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { unsigned long long pad[90]; };
template <int VARIANT>
__global__ __launch_bounds__(128, 4) void k(Big big, unsigned *alive, unsigned *peak, int spin) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ unsigned short QT[258];
  __shared__ int sh[3];
  asm volatile("v_mov_b32 v95, 0" ::: "v95");
  asm volatile("s_mov_b32 s99, 0" ::: "s99");
  if (threadIdx.x == 0) { unsigned a = atomicAdd(alive, 1u) + 1u; atomicMax(peak, a); sh[0] = (int)big.pad[3]; }
  for (int i = threadIdx.x; i < 258; i += 128) QT[i] = (unsigned short)i;
  for (int i = threadIdx.x; i < 21000; i += 128) lds[i] = (unsigned char)i;
  __syncthreads();
  if (VARIANT >= 1 && threadIdx.x < 64) __builtin_amdgcn_s_setprio(2);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {
    if (VARIANT >= 2) { for (int j = 0; j < 64; j++) acc += lds[(threadIdx.x * 67 + j * 64 + acc) % 21000]; __syncthreads(); }
    else __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(alive, 1u);
  if (acc == 0x12345 && QT[threadIdx.x] == 999 && sh[0] == 77) alive[1] = 1;
}
template <int V> void run(unsigned *d, const char *name, hipStream_t st) {
  Big big{};
  hipMemsetAsync(d, 0, 64, st);
  hipLaunchKernelGGL((k<V>), dim3(1672), dim3(128), 21792, st, big, d, d + 4, 400000);
  hipStreamSynchronize(st);
  unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("%s  static 528 + dynamic 21792  peak alive %5u  = %.2f per CU\n", name, h[4], h[4] / 256.0);
}
int main() {
  unsigned *d; hipMalloc(&d, 64);
  hipStream_t st; hipStreamCreate(&st);
  run<0>(d, "sleep          ", st);
  run<1>(d, "sleep + setprio", st);
  run<2>(d, "lds traffic    ", st);
  run<0>(d, "default stream ", 0);
  // other streams that have been used (each gets a hardware queue): does their existence change what one queue may hold?
  hipStream_t extra[6];
  for (auto &e : extra) { hipStreamCreate(&e); hipMemsetAsync(d + 8, 0, 4, e); hipLaunchKernelGGL((k<0>), dim3(1), dim3(128), 21792, e, Big{}, d + 8, d + 12, 10); hipStreamSynchronize(e); }
  run<0>(d, "after 6 streams", st);
  run<2>(d, "after 6 streams", extra[5]);
  return 0;
}
