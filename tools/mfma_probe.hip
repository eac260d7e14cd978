// mfma_probe.hip -- what does the gfx950 matrix pipe lose to barriers, LDS fragment reads, LDS-DMA loads and global
// stores issued around v_mfma_f32_32x32x16_bf16?  Standalone probe behind the GEMM design notes in DESIGN.md.
//
// One "K-tile" = 2048 matrix-pipe cycles per SIMD (64 MFMAs of 32 cycles; with 2 waves per SIMD each wave issues 32).
// Each configuration runs `ktiles` K-tiles per workgroup and reports shader cycles (s_memtime) per K-tile, the MFMA
// utilisation = 2048 / that, and the shader clock under this load (s_memtime vs the 100 MHz s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// WAVES: waves per workgroup; NB: N fragments per wave (wave tile 128 x 32*NB); NLOAD: LDS-DMA loads per wave per
// K-tile; NREAD: 1 = read the MA+NB fragments of every k-step from the LDS; BAR: 1 = one s_barrier per K-tile;
// SPREAD: 0 = all loads right after the barrier, 1 = one load behind every (MFMAs/NLOAD)-th MFMA;
// NSTORE: global_store_dwordx4 per wave per K-tile (deferred-epilogue model), spread over the k-steps.
template <int WAVES, int NB, int NLOAD, int NREAD, int BAR, int SPREAD, int NSTORE, int WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void probe(const char* __restrict__ src, char* __restrict__ dst, size_t dst_mask,
                                                    float* __restrict__ sink, long long* __restrict__ clk, int ktiles, int RS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MA = 4, NF = MA + NB, NM = MA * NB;  // NM MFMAs per k-step
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 65536 / 16; i += WAVES * 64) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(src)[i];
  __syncthreads();

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds0);
  const int sw = (l31 >> 1) & 7;
  unsigned fM[4], fN[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int xk = ((2 * kk + hb) ^ sw) << 4;
    fM[kk] = lds0 + l31 * 128 + xk;          // + buf*16384 + mi*4096
    fN[kk] = lds0 + 32768 + l31 * 128 + xk;  // + buf*16384 + ni*4096
  }
  f32x16 acc[MA][NB];
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  i32x4 F0[NF], F1[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    F0[i] = reinterpret_cast<const i32x4*>(src)[(i * 64 + lane) + w * 1024];
    F1[i] = reinterpret_cast<const i32x4*>(src)[(i * 64 + lane) + w * 1024 + 512];
    asm volatile("" : "+v"(F0[i]), "+v"(F1[i]));
  }
  const unsigned loff = (unsigned)lane * 16u;
  i32x4 sdata = F0[0];
  char* sp = dst + (((size_t)blockIdx.x * WAVES + w) * 1024 + lane * 16);
  const size_t sstep = (size_t)gridDim.x * WAVES * 1024;

#define DSREAD(dst_, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr), "i"(off))
#define READ(F, buf, kk)                                                       \
  if (NREAD) {                                                                 \
    DSREAD(F[MA], fN[kk], (buf) * 16384);                                      \
    if (NB > 1) DSREAD(F[MA + 1], fN[kk], (buf) * 16384 + 4096);               \
    if (NB > 2) DSREAD(F[MA + (NB > 2 ? 2 : 0)], fN[kk], (buf) * 16384 + 8192);  \
    if (NB > 3) DSREAD(F[MA + (NB > 3 ? 3 : 0)], fN[kk], (buf) * 16384 + 12288); \
    DSREAD(F[0], fM[kk], (buf) * 16384);                                       \
    DSREAD(F[1], fM[kk], (buf) * 16384 + 4096);                                \
    DSREAD(F[2], fM[kk], (buf) * 16384 + 8192);                                \
    DSREAD(F[3], fM[kk], (buf) * 16384 + 12288);                               \
  }                                                                            \
  FENCE();
#define WAIT_PREV()                                                            \
  if (NREAD) {                                                                 \
    if (NF == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");            \
    else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                    \
  }                                                                            \
  FENCE();
  int ld_j = 0;
  // RS > 0: the GEMM's real address pattern -- an instruction covers 8 rows x 128 B of a row-major operand with row
  // pitch RS bytes; 32 pieces = 256 rows per operand per K-tile; the A panel is this block's own, the W panel is
  // shared by 8 blocks; K = 1024 (16 K-tiles per output tile)
  const unsigned roff = (unsigned)((lane >> 3) * RS + (lane & 7) * 16);
  auto one_load = [&](int t, int j) {
    if (RS > 0) {
      const int p = w * (NLOAD > 0 ? NLOAD : 1) + j, op = (p >> 5) & 1, pr = p & 31;
      const int tile = t >> 4, kt = t & 15;
      const size_t panel = op ? (size_t)(256 + ((tile + (blockIdx.x >> 3)) % 12)) : (size_t)blockIdx.x;
      const char* b = src + (panel * 256 + pr * 8) * (size_t)RS + kt * 128;
      const unsigned dm = lds_base + 65536 + ((p & 15) * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(roff), "s"(b), "s"(dm) : "memory");
      return;
    }
    const char* b = src + (size_t)((((t * WAVES + w) * (NLOAD > 0 ? NLOAD : 1) + j) * 1024) & (2097152 - 1));
    const unsigned dm = lds_base + 65536 + (((w * (NLOAD > 0 ? NLOAD : 1) + j) & 15) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(b), "s"(dm) : "memory");
  };
  auto one_store = [&]() {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(sp), "v"(sdata) : "memory");
    sp = dst + ((size_t)(sp - dst + sstep) & dst_mask);
  };
  // MFMAs of one k-step, with the spread loads (LPS per k-step) and stores behind them
  constexpr int LPS = NLOAD / 4, SPS4 = NSTORE;  // stores: NSTORE per K-tile, the last VMEM ops before the vmcnt wait
#define MFMAS(F, t, kk)                                                                                         \
  _Pragma("unroll") for (int mi = 0; mi < MA; ++mi) _Pragma("unroll") for (int ni = 0; ni < NB; ++ni) {         \
    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[MA + ni]),               \
                                                          __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
    if (SPREAD && LPS > 0) {                                                                                    \
      constexpr int every = NM / (LPS > 0 ? LPS : 1);                                                           \
      if ((mi * NB + ni) % every == every - 1 && (mi * NB + ni) / every < LPS) { FENCE(); one_load(t, (kk) * LPS + (mi * NB + ni) / every); FENCE(); } \
    }                                                                                                           \
    if ((kk) == 2 && mi * NB + ni >= NM - SPS4) { FENCE(); one_store(); FENCE(); } /* youngest VMEM ops before the wait */            \
  }                                                                                                             \
  FENCE();

#define KTILE(buf, t)                                                               \
  if (!SPREAD) { _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) one_load(t, j); } \
  FENCE();                                                                          \
  READ(F1, buf, 1) WAIT_PREV() MFMAS(F0, t, 0)                                      \
  READ(F0, buf, 2) WAIT_PREV() MFMAS(F1, t, 1)                                      \
  READ(F1, buf, 3) WAIT_PREV() MFMAS(F0, t, 2)                                      \
  if (NREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
  if (NLOAD) {                                                                      \
    if (NSTORE) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NSTORE) : "memory");       \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
  }                                                                                 \
  FENCE();                                                                          \
  if (BAR) __builtin_amdgcn_s_barrier();                                            \
  FENCE();                                                                          \
  READ(F0, (buf) ^ 1, 0)                                                            \
  MFMAS(F1, t, 3)

  (void)ld_j;
  READ(F0, 0, 0)
  const long long c0 = __builtin_readcyclecounter();
  const long long r0 = wall_clock64();
#pragma unroll 1
  for (int t = 0; t < ktiles; t += 2) {
    KTILE(0, t)
    KTILE(1, t + 1)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long c1 = __builtin_readcyclecounter();
  const long long r1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#pragma unroll
  for (int i = 0; i < NF; ++i) s += (float)(F0[i][0] + F1[i][1]);
  sink[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
  if (tid == 0) {
    clk[blockIdx.x * 2] = c1 - c0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

struct Bufs {
  char* src;
  char* dst;
  size_t dst_bytes;
  float* sink;
  long long* clk;
};

template <int WAVES, int NB, int NLOAD, int NREAD, int BAR, int SPREAD, int NSTORE, int WG = 1>
static void run(const Bufs& b, int wg_per_cu, const char* what, int RS = 0) {
  auto kern = probe<WAVES, NB, NLOAD, NREAD, BAR, SPREAD, NSTORE, WAVES * WG / 4>;
  const int smem = 65536 + 16384;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, wg_per_cu == 1 ? 160 * 1024 : smem);
  const int grid = 256 * wg_per_cu, ktiles = 2000;
  const int dyn = wg_per_cu == 1 ? 160 * 1024 : smem;  // 160 KiB pins one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), dyn, 0, b.src, b.dst, b.dst_bytes - 1, b.sink, b.clk, ktiles, RS);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 2);
  hipMemcpy(h.data(), b.clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < grid; ++i) {
    cyc += h[2 * i];
    rt += h[2 * i + 1];
  }
  cyc /= grid;
  rt /= grid;
  // matrix-pipe cycles per K-tile per SIMD: waves per SIMD x MFMAs per wave x 32
  const double waves_per_simd = WAVES * wg_per_cu / 4.0;
  const double pipe = waves_per_simd * 4 * 4 * NB * 32.0;
  const double per_kt = cyc / ktiles;
  const double flops = (double)grid * WAVES * ktiles * 4.0 * 4 * NB * 2.0 * 32 * 32 * 16;
  printf("%-58s cyc/ktile %7.0f  pipe %5.0f  util %5.1f%%  clk %.2f GHz  %7.1f TF (%.3f ms)\n", what, per_kt, pipe,
         100.0 * pipe / per_kt, cyc / (rt * 10.0) , flops / (ms * 1e-3) / 1e12, ms);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  Bufs b;
  const size_t src_bytes = (size_t)(256 + 12) * 256 * 16384 + (4 << 20);  // 268 panels of 256 rows at up to 16 KiB pitch
  b.dst_bytes = (size_t)1 << 30;
  hipMalloc(&b.src, src_bytes);
  hipMalloc(&b.dst, b.dst_bytes);
  hipMalloc(&b.sink, 512 * 512 * sizeof(float));
  hipMalloc(&b.clk, 512 * 2 * sizeof(long long));
  {  // random bf16 in (-1, 1)
    std::vector<unsigned short> h((size_t)(64 << 20) / 2);
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      const float f = ((int)(s >> 8) & 0xffff) / 32768.f - 1.f;
      unsigned u;
      memcpy(&u, &f, 4);
      v = (unsigned short)(u >> 16);
    }
    for (size_t o = 0; o < src_bytes; o += (64 << 20))
      hipMemcpy(b.src + o, h.data(), src_bytes - o < (64 << 20) ? src_bytes - o : (64 << 20), hipMemcpyHostToDevice);
  }
  hipMemset(b.dst, 0, b.dst_bytes);
  //   WAVES NB NLOAD NREAD BAR SPREAD NSTORE
  if (getenv("PROBE_PITCH")) {
    const int pitches[] = {0, 2048, 2048 + 128, 2048 + 256, 2048 + 512, 1536, 8192, 8192 + 128, 6144};
    for (int rs : pitches) {
      char nm[96];
      snprintf(nm, sizeof nm, "w8 + ds_reads + 8 loads (burst) + barrier, pitch %d", rs);
      run<8, 2, 8, 1, 1, 0, 0>(b, 1, nm, rs);
      snprintf(nm, sizeof nm, "w8 + 8 loads (burst) + barrier, no reads, pitch %d", rs);
      run<8, 2, 8, 0, 1, 0, 0>(b, 1, nm, rs);
      snprintf(nm, sizeof nm, "w4 + ds_reads + 16 loads (spread) + barrier, pitch %d", rs);
      run<4, 4, 16, 1, 1, 1, 0>(b, 1, nm, rs);
    }
    return 0;
  }
  printf("--- 4 waves (1 per SIMD), wave tile 128x128, one workgroup per CU\n");
  run<4, 4, 0, 0, 0, 0, 0>(b, 1, "w4 mfma only");
  run<4, 4, 0, 0, 1, 0, 0>(b, 1, "w4 + barrier");
  run<4, 4, 0, 1, 0, 0, 0>(b, 1, "w4 + ds_reads");
  run<4, 4, 0, 1, 1, 0, 0>(b, 1, "w4 + ds_reads + barrier");
  run<4, 4, 16, 0, 1, 0, 0>(b, 1, "w4 + 16 loads (burst) + barrier");
  run<4, 4, 16, 0, 1, 1, 0>(b, 1, "w4 + 16 loads (spread) + barrier");
  run<4, 4, 16, 0, 0, 1, 0>(b, 1, "w4 + 16 loads (spread), no barrier");
  run<4, 4, 16, 1, 1, 0, 0>(b, 1, "w4 + ds_reads + 16 loads (burst) + barrier");
  run<4, 4, 16, 1, 1, 1, 0>(b, 1, "w4 + ds_reads + 16 loads (spread) + barrier");
  run<4, 4, 8, 1, 1, 1, 0>(b, 1, "w4 + ds_reads + 8 loads (spread) + barrier");
  run<4, 4, 16, 1, 1, 1, 2>(b, 1, "w4 + ds_reads + 16 loads (spread) + 2 stores + barrier");
  run<4, 4, 16, 1, 1, 1, 4>(b, 1, "w4 + ds_reads + 16 loads (spread) + 4 stores + barrier");
  run<4, 4, 0, 0, 0, 0, 2>(b, 1, "w4 mfma + 2 stores");
  printf("--- 8 waves (2 per SIMD), wave tile 128x64, one workgroup per CU\n");
  run<8, 2, 0, 0, 0, 0, 0>(b, 1, "w8 mfma only");
  run<8, 2, 0, 0, 1, 0, 0>(b, 1, "w8 + barrier");
  run<8, 2, 0, 1, 1, 0, 0>(b, 1, "w8 + ds_reads + barrier");
  run<8, 2, 8, 0, 1, 0, 0>(b, 1, "w8 + 8 loads (burst) + barrier");
  run<8, 2, 8, 1, 1, 0, 0>(b, 1, "w8 + ds_reads + 8 loads (burst) + barrier");
  run<8, 2, 8, 1, 1, 1, 0>(b, 1, "w8 + ds_reads + 8 loads (spread) + barrier");
  run<8, 2, 8, 1, 1, 1, 1>(b, 1, "w8 + ds_reads + 8 loads (spread) + 1 store + barrier");
  printf("--- 2 workgroups of 4 waves per CU (independent barriers), wave tile 128x64\n");
  run<4, 2, 0, 0, 1, 0, 0, 2>(b, 2, "2x w4 + barrier");
  run<4, 2, 8, 1, 1, 0, 0, 2>(b, 2, "2x w4 + ds_reads + 8 loads (burst) + barrier");
  run<4, 2, 8, 1, 1, 1, 0, 2>(b, 2, "2x w4 + ds_reads + 8 loads (spread) + barrier");
  run<4, 2, 12, 1, 1, 1, 1, 2>(b, 2, "2x w4 + ds_reads + 12 loads (spread) + 1 store + barrier");
  return 0;
}
