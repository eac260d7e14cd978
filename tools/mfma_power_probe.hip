// mfma_power_probe.hip -- what does the MI355X sustain on register-only MFMA loops of different shapes, under its power management?
// (round 4: the vendor's GEMM uses v_mfma_f32_16x16x32_bf16 with 256 AGPR accumulators on 4 waves; this library
// v_mfma_f32_32x32x16 with 128 VGPR accumulators on 8 waves.  Is one of them cheaper in joules per flop?)
// Every variant issues the same number of flops per CU from operands that were loaded once from random bf16 / fp16 data and
// rotate among NF fragment registers; no memory traffic in the loop.  Reports TFLOP/s and the shader clock (s_memtime per wall).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_power_probe tools/mfma_power_probe.hip && tools/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include <cmath>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// SHAPE 0: 32x32x16 (16 acc regs per block, 32768 flop), SHAPE 1: 16x16x32 (4 acc regs per block, 16384 flop).  NBLK accumulator
// blocks per wave, arranged MA x NB (A fragment i with B fragment j).  F16: fp16 operands.
template <int WAVES, int SHAPE, int MA, int NB, bool F16>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(const i32x4* __restrict__ src, float* __restrict__ sink, long long* __restrict__ clk, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  i32x4 A[MA], B[NB];
#pragma unroll
  for (int i = 0; i < MA; ++i) A[i] = src[(w * 32 + i) * 64 + lane];
#pragma unroll
  for (int j = 0; j < NB; ++j) B[j] = src[(w * 32 + 16 + j) * 64 + lane];
  long long t0 = __builtin_readcyclecounter();
  if constexpr (SHAPE == 0) {
    f32x16 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[i]), __builtin_bit_cast(f16x8, B[j]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[j]), acc[i][j], 0, 0, 0);
        }
      // new operand bits every k-step, as fragments read from the LDS would be (a rotation keeps the value distribution)
#pragma unroll
      for (int i = 0; i < MA; ++i) A[i] = i32x4{A[i].y, A[i].z, A[i].w, A[i].x};
#pragma unroll
      for (int j = 0; j < NB; ++j) B[j] = i32x4{B[j].w, B[j].x, B[j].y, B[j].z};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
  } else {
    f32x4 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[i]), __builtin_bit_cast(f16x8, B[j]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[j]), acc[i][j], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < MA; ++i) A[i] = i32x4{A[i].y, A[i].z, A[i].w, A[i].x};
#pragma unroll
      for (int j = 0; j < NB; ++j) B[j] = i32x4{B[j].w, B[j].x, B[j].y, B[j].z};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

struct Variant {
  const char* name;
  void (*launch)(const i32x4*, float*, long long*, int, int, hipStream_t);
  double flop_per_iter_per_cu;
};

template <int WAVES, int SHAPE, int MA, int NB, bool F16>
static void launch(const i32x4* src, float* sink, long long* clk, int iters, int grid, hipStream_t st) {
  hipLaunchKernelGGL((probe<WAVES, SHAPE, MA, NB, F16>), dim3(grid), dim3(WAVES * 64), 0, st, src, sink, clk, iters);
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 60.0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  // random normal values as bf16 / fp16 bit patterns (std 1): 8 waves x 32 fragments x 64 lanes x 16 B
  std::vector<uint16_t> hb(8 * 32 * 64 * 8), hf(8 * 32 * 64 * 8);
  unsigned r = 12345u;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (r >> 8) / 16777216.0; };
  for (size_t i = 0; i < hb.size(); ++i) {
    const double g = sqrt(-2.0 * log(rnd() + 1e-12)) * cos(6.283185307 * rnd());
    float f = (float)g;
    uint32_t u;
    memcpy(&u, &f, 4);
    hb[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
    _Float16 h = (_Float16)f;
    memcpy(&hf[i], &h, 2);
  }
  i32x4 *db, *df;
  float* sink;
  long long* clk;
  CK(hipMalloc(&db, hb.size() * 2));
  CK(hipMalloc(&df, hf.size() * 2));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&clk, ncu * 8));
  CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // flop per iteration per CU = waves x MA x NB x flop per MFMA
  const Variant vs[] = {
      {"8 waves, 32x32x16 bf16, 4x2 blocks (128 acc regs)  [this library's GEMM]", launch<8, 0, 4, 2, false>, 8.0 * 8 * 32768},
      {"8 waves, 16x16x32 bf16, 8x4 blocks (128 acc regs)", launch<8, 1, 8, 4, false>, 8.0 * 32 * 16384},
      {"4 waves, 32x32x16 bf16, 4x4 blocks (256 acc regs)", launch<4, 0, 4, 4, false>, 4.0 * 16 * 32768},
      {"4 waves, 16x16x32 bf16, 8x8 blocks (256 acc regs)  [the vendor's GEMM]", launch<4, 1, 8, 8, false>, 4.0 * 64 * 16384},
      {"8 waves, 32x32x16 fp16, 4x2 blocks", launch<8, 0, 4, 2, true>, 8.0 * 8 * 32768},
      {"8 waves, 16x16x32 fp16, 8x4 blocks", launch<8, 1, 8, 4, true>, 8.0 * 32 * 16384},
      {"8 waves, 32x32x16 bf16, 1x1 block (same registers every time)", launch<8, 0, 1, 1, false>, 8.0 * 1 * 32768},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%d CUs; every variant runs ~%.0f ms; two rounds\n", ncu, target_ms);
  for (int round = 0; round < 2; ++round)
    for (const Variant& v : vs) {
      const bool f16 = strstr(v.name, "fp16") != nullptr;
      // iterations for ~target_ms at 1.5 PF
      int iters = (int)(target_ms * 1e-3 * 1.5e15 / (v.flop_per_iter_per_cu * ncu));
      v.launch(f16 ? df : db, sink, clk, iters / 8 + 1, ncu, st);  // warm-up
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      v.launch(f16 ? df : db, sink, clk, iters, ncu, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> hc(ncu);
      CK(hipMemcpy(hc.data(), clk, ncu * 8, hipMemcpyDeviceToHost));
      double cyc = 0;
      for (long long c : hc) cyc += (double)c;
      cyc /= ncu;
      const double tf = v.flop_per_iter_per_cu * ncu * iters / (ms * 1e-3) / 1e12;
      printf("  %-78s %7.1f TF  %6.2f ms  clock %.2f GHz  pipe util %.1f%%\n", v.name, tf, ms, cyc / (ms * 1e6),
             100.0 * (v.flop_per_iter_per_cu / 4.0 / 1017.25 * iters) / cyc);
    }
  return 0;
}
