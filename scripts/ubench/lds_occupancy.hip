// How many workgroups of T threads with L bytes of dynamic LDS run at once on an MI355X?  (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned *alive, unsigned *peak, int spin) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) { unsigned a = atomicAdd(alive, 1u) + 1u; atomicMax(peak, a); }
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(8); }
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(alive, 1u);
  if (lds[(threadIdx.x + 1) & 63] == 255 && spin < 0) alive[1] = 1;
}
int main(int argc, char **argv) {
  unsigned *d; hipMalloc(&d, 64); 
  for (int threads : {64, 128, 256})
    for (int kb : {4, 8, 12, 16, 20, 22, 24, 28, 32, 40, 48, 64}) {
      hipMemset(d, 0, 64);
      size_t lds = (size_t)kb * 1024;
      hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(8192), dim3(threads), lds, 0, d, d + 4, 200000);
      hipDeviceSynchronize();
      unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      printf("threads %3d  lds %2d KB  peak alive %5u  = %.2f per CU\n", threads, kb, h[4], h[4] / 256.0);
    }
  return 0;
}
