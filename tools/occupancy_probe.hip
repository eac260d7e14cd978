// occupancy_probe.hip -- how many workgroups of a given dynamic-LDS size does a gfx950 CU hold at once?
// Every workgroup idles for ~20 us; the wall time of 256 x n workgroups over the time of 256 gives the number of rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(int* sink, long long ticks) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (s[threadIdx.x] == -1) sink[0] = 1;
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {192, 512})
    for (int kib10 : {400, 530, 640, 725, 800}) {
      const int bytes = kib10 * 1024 / 10;
      hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      printf("threads %3d  LDS %5.1f KiB:", threads, bytes / 1024.0);
      for (int n : {1, 2, 3, 4}) {
        hipLaunchKernelGGL(spin, dim3(256 * n), dim3(threads), bytes, 0, sink, 2000);  // 2000 ticks of 10 ns = 20 us
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin, dim3(256 * n), dim3(threads), bytes, 0, sink, 2000);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d x256 WGs %6.1f us", n, ms * 1e3);
      }
      printf("\n");
    }
  return 0;
}
