// Workgroups per CU at 22 KB of LDS as a function of the VGPR / SGPR allocation (128 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V, int S>
__global__ void k(unsigned *alive, unsigned *peak, int spin) {
  extern __shared__ unsigned char lds[];
  if (V >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (V >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (V >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (S >= 100) asm volatile("s_mov_b32 s99, 0" ::: "s99");
  if (threadIdx.x == 0) { unsigned a = atomicAdd(alive, 1u) + 1u; atomicMax(peak, a); }
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(8); }
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(alive, 1u);
  if (lds[(threadIdx.x + 1) & 63] == 255 && spin < 0) alive[1] = 1;
}
template <int V, int S> void run(unsigned *d, const char *name) {
  for (int kb : {16, 22, 24}) {
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL((k<V, S>), dim3(8192), dim3(128), (size_t)kb * 1024, 0, d, d + 4, 200000);
    hipDeviceSynchronize();
    unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%s  lds %2d KB  peak alive %5u  = %.2f per CU\n", name, kb, h[4], h[4] / 256.0);
  }
}
int main() {
  unsigned *d; hipMalloc(&d, 64);
  run<0, 0>(d, "vgpr small sgpr small");
  run<64, 0>(d, "vgpr 64            ");
  run<96, 0>(d, "vgpr 96            ");
  run<128, 0>(d, "vgpr 128           ");
  run<0, 100>(d, "sgpr 100           ");
  run<96, 100>(d, "vgpr 96 sgpr 100   ");
  return 0;
}
