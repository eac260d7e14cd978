// tr_probe.hip -- what ds_read_b64_tr_b16 returns on gfx950 (tools only).
//   hipcc --offload-arch=gfx950 -O2 -o tools/tr_probe tools/tr_probe.hip && tools/tr_probe
// LDS holds a [64 rows][32 cols] bf16 image with 64-byte rows, element value = row * 100 + col (as integer bits in a u16).
// Hypothesis (guide T10): per 16-lane group the instruction reads a [4 rows][16 cols] tile -- lane i of the group supplies the
// address of row (i >> 2), column piece 4 (i & 3) (8 bytes) -- and lane i receives column i: 4 values, rows 0..3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 32];
  for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (uint16_t)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  // group g: rows 8 * (g >> 1) + (i >> 2), column block 16 * (g & 1), piece 4 * (i & 3)
  const int row = 8 * (g >> 1) + (i >> 2), col = 16 * (g & 1) + 4 * (i & 3);
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + (row * 32 + col) * 2;
  if (mode == 1) addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + (8 * (g >> 1) * 32 + 16 * (g & 1)) * 2;  // uniform per group
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v[0] & 0xffff);
  out[l * 4 + 1] = (uint16_t)((unsigned)v[0] >> 16);
  out[l * 4 + 2] = (uint16_t)(v[1] & 0xffff);
  out[l * 4 + 3] = (uint16_t)((unsigned)v[1] >> 16);
}
int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (%s addresses)\n", mode, mode ? "uniform-per-group" : "per-lane piece");
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" r%02dc%02d", h[l * 4 + j] / 100, h[l * 4 + j] % 100);
      printf("%s", (l & 3) == 3 ? "\n" : "   ");
    }
  }
  return 0;
}
