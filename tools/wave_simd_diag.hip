// prints the wave -> (SIMD, CU, XCC) placement of 512-thread workgroups (gfx950): HW_REG_HW_ID bits
// [3:0] wave_id, [5:4] simd_id, [7:6] pipe, [11:8] cu_id, [12] sh, [15:13] se ; XCC_ID register 20
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void diag(unsigned* out) {
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[(blockIdx.x * 8 + w) * 2] = hwid;
    out[(blockIdx.x * 8 + w) * 2 + 1] = xcc;
  }
}
int main() {
  unsigned* d;
  const int nb = 16;
  hipMalloc(&d, nb * 8 * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(diag, dim3(nb), dim3(512), 65536 * 2, 0, d);
  unsigned h[nb * 16];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < nb; ++b) {
    printf("block %2d:", b);
    for (int w = 0; w < 8; ++w) {
      unsigned id = h[(b * 8 + w) * 2];
      printf("  w%d simd%u cu%u", w, (id >> 4) & 3, (id >> 8) & 15);
    }
    printf("  xcc%u\n", h[b * 16 + 1] & 15);
  }
  return 0;
}
