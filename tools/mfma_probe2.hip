// mfma_probe2.hip -- the GEMM skeleton of this library (8 waves, wave tile 128 x 64, fragments by ds_read_b128 from a two-stage LDS
// image, 8 LDS-DMA loads per wave per 64-deep K-tile, one barrier per K-tile) with the two MFMA shapes:
//   SHAPE 0  v_mfma_f32_32x32x16_bf16: 6 fragment reads + 8 MFMAs per 16-deep k-step   (what gemm256sp.hip issues)
//   SHAPE 1  v_mfma_f32_16x16x32_bf16: 12 fragment reads + 32 MFMAs per 32-deep k-step (same LDS bytes, same flops)
// Round 4: tools/mfma_power_probe showed the 16x16x32 form sustaining ~10 % more TFLOP/s on register-only loops under the part's
// power management; does that survive the skeleton?  Operands: normal(0, 1) bf16.  Timing only (results are not checked).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe2 tools/mfma_probe2.hip && tools/mfma_probe2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define DSREAD(dst_, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr), "i"(off))

template <int SHAPE, int NLOAD, int NREAD, int BAR>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, float* __restrict__ sink, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 131072 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i];
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds0);
  const int sw = (l31 >> 1) & 7;
  unsigned fMb[2][4], fNb[2][4];  // [stage][k-step] (the stage offset does not fit the 16-bit immediate of ds_read)
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int xk = ((2 * kk + hb) ^ sw) << 4;
    fMb[0][kk] = lds0 + ((w >> 2) * 128 + l31) * 128 + xk;          // + mi*4096
    fNb[0][kk] = lds0 + 32768 + ((w & 3) * 64 + l31) * 128 + xk;    // + ni*4096
    fMb[1][kk] = fMb[0][kk] + 65536;
    fNb[1][kk] = fNb[0][kk] + 65536;
  }
  const unsigned loff = (unsigned)lane * 16u;
  auto one_load = [&](int t, int j) {
    const char* b = src + (size_t)((((t * 8 + w) * NLOAD + j) * 1024) & ((64u << 20) - 1));
    const unsigned dm = lds_base + 131072 + (((w * NLOAD + j) & 15) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(b), "s"(dm) : "memory");
  };
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    i32x4 F0[6], F1[6];
#define READ6(F, buf, kk)                                    \
  if (NREAD) {                                               \
    DSREAD(F[4], fNb[(buf) & 1][kk], 0);                     \
    DSREAD(F[5], fNb[(buf) & 1][kk], 0 + 4096);              \
    DSREAD(F[0], fMb[(buf) & 1][kk], 0);                     \
    DSREAD(F[1], fMb[(buf) & 1][kk], 0 + 4096);              \
    DSREAD(F[2], fMb[(buf) & 1][kk], 0 + 8192);              \
    DSREAD(F[3], fMb[(buf) & 1][kk], 0 + 12288);             \
  }                                                          \
  FENCE();
#define WAIT6() if (NREAD) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); FENCE();
#define MFMA8(F)                                                                                                               \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =              \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
  FENCE();
#define KTILE0(buf, t)                                                              \
  _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) one_load(t, j);                 \
  FENCE();                                                                          \
  READ6(F1, buf, 1) WAIT6() MFMA8(F0)                                               \
  READ6(F0, buf, 2) WAIT6() MFMA8(F1)                                               \
  READ6(F1, buf, 3) WAIT6() MFMA8(F0)                                               \
  if (NREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
  if (NLOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
  FENCE();                                                                          \
  if (BAR) __builtin_amdgcn_s_barrier();                                            \
  FENCE();                                                                          \
  READ6(F0, (buf) ^ 1, 0)                                                           \
  MFMA8(F1)
#pragma unroll
    for (int i = 0; i < 6; ++i) { F0[i] = reinterpret_cast<const i32x4*>(src)[i * 64 + lane]; F1[i] = reinterpret_cast<const i32x4*>(src)[(i + 8) * 64 + lane]; asm volatile("" : "+v"(F0[i]), "+v"(F1[i])); }
    READ6(F0, 0, 0)
#pragma unroll 1
    for (int t = 0; t < ktiles; t += 2) {
      KTILE0(0, t)
      KTILE0(1, t + 1)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    i32x4 P[12], Q[12];  // [0..7] A blocks (16 rows x 32 k), [8..11] B blocks
    // the same 12 x 1 KiB per 32-deep step as two 16-deep steps of the other shape (addresses of k-steps kk, kk + 1)
#define READ12(F, buf, kk)                                     \
  if (NREAD) {                                                 \
    DSREAD(F[8], fNb[(buf) & 1][kk], 0);                       \
    DSREAD(F[9], fNb[(buf) & 1][kk], 0 + 4096);                \
    DSREAD(F[10], fNb[(buf) & 1][(kk) + 1], 0);                \
    DSREAD(F[11], fNb[(buf) & 1][(kk) + 1], 0 + 4096);         \
    DSREAD(F[0], fMb[(buf) & 1][kk], 0);                       \
    DSREAD(F[1], fMb[(buf) & 1][kk], 0 + 4096);                \
    DSREAD(F[2], fMb[(buf) & 1][kk], 0 + 8192);                \
    DSREAD(F[3], fMb[(buf) & 1][kk], 0 + 12288);               \
    DSREAD(F[4], fMb[(buf) & 1][(kk) + 1], 0);                 \
    DSREAD(F[5], fMb[(buf) & 1][(kk) + 1], 0 + 4096);          \
    DSREAD(F[6], fMb[(buf) & 1][(kk) + 1], 0 + 8192);          \
    DSREAD(F[7], fMb[(buf) & 1][(kk) + 1], 0 + 12288);         \
  }                                                            \
  FENCE();
#define WAIT12() if (NREAD) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); FENCE();
#define MFMA32(F)                                                                                                              \
  _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[mi][ni] =              \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[8 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
  FENCE();
#define KTILE1(buf, t)                                                              \
  _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) one_load(t, j);                 \
  FENCE();                                                                          \
  READ12(Q, buf, 2) WAIT12() MFMA32(P)                                              \
  if (NREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
  if (NLOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
  FENCE();                                                                          \
  if (BAR) __builtin_amdgcn_s_barrier();                                            \
  FENCE();                                                                          \
  READ12(P, (buf) ^ 1, 0)                                                           \
  MFMA32(Q)
#pragma unroll
    for (int i = 0; i < 12; ++i) { P[i] = reinterpret_cast<const i32x4*>(src)[i * 64 + lane]; Q[i] = reinterpret_cast<const i32x4*>(src)[(i + 12) * 64 + lane]; asm volatile("" : "+v"(P[i]), "+v"(Q[i])); }
    READ12(P, 0, 0)
#pragma unroll 1
    for (int t = 0; t < ktiles; t += 2) {
      KTILE1(0, t)
      KTILE1(1, t + 1)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  sink[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int SHAPE, int NLOAD, int NREAD, int BAR>
static void run(const char* src, float* sink, const char* what) {
  auto kern = probe<SHAPE, NLOAD, NREAD, BAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256, ktiles = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, src, sink, ktiles);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = (double)grid * ktiles * 2.0 * 256 * 256 * 64;
  printf("  %-70s %7.1f TF  (%.2f ms)\n", what, flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  char* src;
  float* sink;
  const size_t src_bytes = (size_t)64 << 20;
  hipMalloc(&src, src_bytes + (1 << 20));
  hipMalloc(&sink, 256 * 512 * 4);
  std::vector<uint16_t> h(src_bytes / 2);
  unsigned r = 12345u;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (r >> 8) / 16777216.0; };
  for (size_t i = 0; i < h.size(); i += 2) {
    const double a = sqrt(-2.0 * log(rnd() + 1e-12)), b = 6.283185307 * rnd();
    const float f[2] = {(float)(a * cos(b)), (float)(a * sin(b))};
    for (int j = 0; j < 2; ++j) {
      uint32_t u;
      memcpy(&u, &f[j], 4);
      h[i + j] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
    }
  }
  hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<0, 0, 0, 0>(src, sink, "32x32x16: mfma only");
    run<1, 0, 0, 0>(src, sink, "16x16x32: mfma only");
    run<0, 0, 1, 1>(src, sink, "32x32x16: + ds_reads + barrier");
    run<1, 0, 1, 1>(src, sink, "16x16x32: + ds_reads + barrier");
    run<0, 8, 1, 1>(src, sink, "32x32x16: + ds_reads + 8 LDS-DMA loads + barrier  [gemm256sp skeleton]");
    run<1, 8, 1, 1>(src, sink, "16x16x32: + ds_reads + 8 LDS-DMA loads + barrier");
  }
  return 0;
}
