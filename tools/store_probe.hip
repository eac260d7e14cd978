// store_probe.hip -- per-CU global store throughput on gfx950 for the GEMM epilogue's access shapes.
// Each wave issues `n` global_store_dwordx4 (1 KiB per instruction); an instruction covers ROWS rows x (1024/ROWS) bytes of
// a row-major matrix with `pitch` bytes per row.  Reports shader cycles per store instruction per CU and bytes/clk/CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0 plain, 1 sc1, 2 nt, 3 sc0 sc1
template <int MODE>
__global__ __launch_bounds__(512) void stores(char* __restrict__ dst, size_t span_mask, int rows, int pitch, int n,
                                              long long* __restrict__ clk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lpr = 64 / rows;  // lanes per row
  // wave's instruction i covers rows [ (i*nw + w)*rows, +rows ) of this block's 256-row panel set
  const size_t lane_off = (size_t)(lane / lpr) * pitch + (size_t)(lane % lpr) * 16;
  const size_t blk = (size_t)blockIdx.x * 256 * pitch;
  i32x4 v = {lane, w, n, rows};
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < n; ++i) {
    const size_t r = (size_t)((i * nw + w) * rows);
    // rows wrap inside a 256-row panel; successive panels advance by 1024/rows*... columns
    const size_t off = (blk + (r & 255) * pitch + (r >> 8) * (1024 / rows) + lane_off) & span_mask;
    char* p = dst + (off & ~(size_t)15);
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE>
static void run(char* dst, size_t span, long long* clk, int waves, int rows, int pitch, int n, const char* what) {
  const int grid = getenv("STORE_PROBE_GRID") ? atoi(getenv("STORE_PROBE_GRID")) : 256;  // fewer workgroups: is the rate per CU or per chip?
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(stores<MODE>, dim3(grid), dim3(waves * 64), 0, 0, dst, span - 1, rows, pitch, n, clk);
  hipDeviceSynchronize();
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), clk, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double c = 0;
  for (auto x : h) c += x;
  c /= grid;
  const double per = c / ((double)n * waves);
  printf("grid %3d %-44s waves %d rows/instr %d pitch %6d  cyc/store/CU %6.1f  B/clk/CU %5.1f\n", grid, what, waves, rows, pitch, per, 1024.0 / per);
}

int main() {
  char* dst;
  long long* clk;
  const size_t big = (size_t)1 << 30, small = (size_t)8 << 20;
  hipMalloc(&dst, big);
  hipMalloc(&clk, 256 * sizeof(long long));
  hipMemset(dst, 0, big);
  const int n = 256;  // stores per wave
  for (int waves : {4, 8}) {
    run<0>(dst, big, clk, waves, 1, 1024, n, "plain, contiguous 1 KiB");
    run<0>(dst, big, clk, waves, 8, 6144, n, "plain, 8 rows x 128 B");
    run<0>(dst, big, clk, waves, 4, 6144, n, "plain, 4 rows x 256 B");
    run<0>(dst, big, clk, waves, 2, 6144, n, "plain, 2 rows x 512 B");
    run<0>(dst, big, clk, waves, 1, 6144, n, "plain, 1 row x 1 KiB");
    run<1>(dst, big, clk, waves, 8, 6144, n, "sc1, 8 rows x 128 B");
    run<2>(dst, big, clk, waves, 8, 6144, n, "nt, 8 rows x 128 B");
    run<3>(dst, big, clk, waves, 8, 6144, n, "sc0 sc1, 8 rows x 128 B");
    run<0>(dst, big, clk, waves, 8, 2048, n, "plain, 8 rows x 128 B");
    run<0>(dst, big, clk, waves, 8, 6144 + 128, n, "plain, 8 rows x 128 B");
    run<0>(dst, small, clk, waves, 8, 6144, n, "plain, 8 rows x 128 B, 8 MiB span (L2)");
    run<0>(dst, small, clk, waves, 1, 1024, n, "plain, contiguous, 8 MiB span (L2)");
  }
  return 0;
}
