// gemm256r4.hip -- persistent 256x256 bf16 GEMM, K-tiles of 32 in a 4-deep LDS-DMA ring (GemmArgs.variant == 4).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A bf16 [M, K] activations, W bf16 [N, K] (torch Linear)
//
// Same role, tile, wave layout, fragment order and epilogue as gemm256sp.hip (the linear layers inside
// `model.encode_image/encode_text`, reference clip_retrieval/clip_inference/mapper.py:57,65).  What changes is the
// operand pipeline.  Measured on gemm256sp (PMC, QKV shape): the main loop spends ~3200 cycles per 64-deep K-tile
// against 2048 of MFMA work, and the gap is the LATENCY of the LDS-DMA issued exactly one K-tile (2048 MFMA cycles)
// earlier: 128 KiB of operand LDS only holds two 64-deep K-tiles.  Here the same 128 KiB are a ring of FOUR 32-deep
// K-tiles, so a K-tile's DMA is issued three K-tiles (3072 MFMA cycles) before its first read:
//     K-tile g, step 1:  lgkmcnt(0); vmcnt(6) [K-tile g+1 landed, g+2 and half of g+3 stay in flight]; s_barrier;
//                        ds_read step 0 of K-tile g+1; 8 MFMA with the last two DMAs of K-tile g+3 behind each four
// at the price of one barrier per 32 k instead of per 64.  LDS rows are 64 B (4 chunks of 16 B), chunk position
// XOR ((row>>2)&3): 16 consecutive rows x one k-chunk hit 16 distinct 16-B slots (conflict-free ds_read_b128).
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0 (the launcher in clip_kernels.hip peels ragged rows).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace clipx {

constexpr int R_OPB = 16384;      // bytes of one operand K-tile: [256 rows][64 B]
constexpr int R_NBASE = 65536;    // N operand ring starts here (M ring: stages 0..3 at 0, 16K, 32K, 48K)
constexpr int S_SCRATCH = 131072; // per-wave 4 KiB epilogue scratch starts here

#define S_FENCE() __builtin_amdgcn_sched_barrier(0)

typedef unsigned r4_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(void* p, uint4 v, bool plain) {  // see gemm256sp.hip
  if (plain) {
    *reinterpret_cast<uint4*>(p) = v;
  } else {
    const r4_u32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
  }
}

// DBG (ablation, EPI_BIAS_BF16 only; garbage results): 1 = no staging, 5 = no epilogue, 6 = epilogue without stores
template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void gemm256r4_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ outp,
                                                          const float* __restrict__ table, int T, int N, int K, int ntm,
                                                          int ntn, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int hb = lane >> 5, l31 = lane & 31;
  const int ntiles = ntm * ntn;

  // ---- tile list of this block (same XCD-aware order as gemm256.hip)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  auto tile_of = [&](int j, int& m0, int& n0) -> bool {
    const int logical = (j * 8 + xcd) * cpx + idx;
    if (logical >= ntiles) return false;
    const int per_group = 8 * ntn;
    const int grp = logical / per_group, within = logical - grp * per_group;
    const int gm0 = grp * 8;
    const int gsz = (ntm - gm0) < 8 ? (ntm - gm0) : 8;
    m0 = (gm0 + within % gsz) * 256;
    n0 = (within / gsz) * 256;
    return true;
  };
  int m0, n0;
  if (!tile_of(0, m0, n0)) return;  // before any barrier

  // ---- staging: one LDS-DMA instruction = 1 KiB = 16 rows x 64 B; wave w fills rows [32w, 32w+32) of both operands
  // (pieces 2w, 2w+1).  Source chunk = LDS chunk position ^ ((row>>2)&3), identical for both pieces (rows 16 apart).
  const int srow = w * 32 + (lane >> 2);
  const int c0 = (lane & 3) ^ ((srow >> 2) & 3);
  const unsigned off0 = (unsigned)((srow * K + (c0 << 3)) * 2);
  const size_t jstep = (size_t)16 * K * 2;
  // One K-tile = 4 DMA instructions per wave.  Measured on gemm256sp (ablations DBG 8/9/10): staging costs ~19 cycles per
  // load INSTRUCTION on the CU's load path, whatever its size and whether it lands in LDS or VGPRs, and that time ADDS to
  // the MFMA time instead of hiding under it when every wave issues its DMAs in one burst behind the barrier (the
  // in-order waves sit in VMEM issue while the queue drains, nobody issues MFMAs).  Here the ring gives a stage two
  // K-tiles of slack, so its four DMAs are issued ONE AT A TIME, each behind four MFMAs of the next K-tile.
  // piece i of a K-tile's stage (one DMA instruction): i = 0, 1 -> M operand pieces j = 0, 1; i = 2, 3 -> N operand
  auto stage_piece = [&](const char* baseM, const char* baseN, int slot, int i) {
    if (DBG == 1) return;
    const int j = i & 1;
    if (i < 2)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseM + j * jstep + off0),
                                       (lds_ptr_t)(smem + slot * R_OPB + (w * 2 + j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseN + j * jstep + off0),
                                       (lds_ptr_t)(smem + R_NBASE + slot * R_OPB + (w * 2 + j) * 1024), 16, 0, 0);
  };
  auto stage = [&](const char* baseM, const char* baseN, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece(baseM, baseN, slot, i);
  };

  // ---- fragment read addresses (LDS byte addresses), one per (operand, k-step of the 32-deep K-tile)
  const int sw = (l31 >> 2) & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned fM[2], fN[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int xk = ((2 * kk + hb) ^ sw) << 4;
    fM[kk] = lds0 + (wr * 128 + l31) * 64 + xk;            // + slot*R_OPB + mi*2048
    fN[kk] = lds0 + R_NBASE + (wc * 64 + l31) * 64 + xk;   // + slot*R_OPB + ni*2048
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 F0[6], F1[6];  // [0..3] M fragments (mi), [4..5] N fragments (ni) of one k-step

#define S_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define S_READ(F, slot, kk)                                                                \
  {                                                                                        \
    S_DSREAD(F[4], fN[kk], (slot) * R_OPB);                                                \
    S_DSREAD(F[5], fN[kk], (slot) * R_OPB + 2048);                                         \
    S_DSREAD(F[0], fM[kk], (slot) * R_OPB);                                                \
    S_DSREAD(F[1], fM[kk], (slot) * R_OPB + 2048);                                         \
    S_DSREAD(F[2], fM[kk], (slot) * R_OPB + 4096);                                         \
    S_DSREAD(F[3], fM[kk], (slot) * R_OPB + 6144);                                         \
  }                                                                                        \
  S_FENCE();
#define S_WAIT_PREV() asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); S_FENCE();
#define S_WAIT_ALL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); S_FENCE();
#define S_MFMA(F)                                                                                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0);

// 8 MFMAs of one k-step with one DMA piece behind each group of four
#define S_MFMA_DMA(F, ok, pM, pN, slot, i0)                                                                        \
  _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
  S_FENCE();                                                                                                       \
  if (ok) stage_piece(pM, pN, slot, i0);                                                                           \
  S_FENCE();                                                                                                       \
  _Pragma("unroll") for (int mi = 2; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
  S_FENCE();                                                                                                       \
  if (ok) stage_piece(pM, pN, slot, (i0) + 1);                                                                     \
  S_FENCE();

  const char* curM = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* curN = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;
  const int nk = K >> 5;  // 32-deep K-tiles per output tile (multiple of 4)

  constexpr bool OUT_BF16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16;
  constexpr bool HAS_BIAS = EPI != EPI_TABLE_F32;
  constexpr int S_EPI_ST = (DBG == 5 || DBG == 6) ? 0 : (OUT_BF16 ? 16 : 32);  // stores of one epilogue per wave
  constexpr int S_NB = HAS_BIAS ? 1 : 0;
  const int rrow = lane >> 3, rch = lane & 7;  // epilogue read-back: row 8i + rrow, 16-B chunk rch
  auto load_bias = [&]() {
    if (!HAS_BIAS) return;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bias + n0 + wc * 64 + (lane & 15) * 4),
                                     (lds_ptr_t)(smem + S_SCRATCH + w * 4096), 16, 0, 0);
  };

  // ---- prologue: K-tiles 0..2 of the first tile (K-tile 3 rides on K-tile 0's MFMAs); K-tile 0 landed + first fragments read
  stage(curM, curN, 0);
  stage(curM + 64, curN + 64, 1);
  stage(curM + 128, curN + 128, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  S_FENCE();
  __builtin_amdgcn_s_barrier();
  S_FENCE();
  S_READ(F0, 0, 0)

  // VMEM bookkeeping (vmcnt retires in order).  The stage of K-tile x+3 (slot (x+3)&3 = (x-1)&3, released at K-tile
  // x-1's barrier) rides on K-tile x's 16 MFMAs: pieces 0, 1 in step 0, pieces 2, 3 in step 1 (after the barrier).  At
  // K-tile g's barrier "K-tile g+1 has landed" therefore leaves stage(g+2) (4) and the first two pieces of stage(g+3)
  // in flight: vmcnt(6).  After an epilogue its S_EPI_ST stores and the bias DMA are also younger than the awaited
  // K-tile for K-tile 0 of the new tile (age 0: 6 + bias + stores) and the stores for K-tile 1 (age 1: 6 + stores).  When
  // the stream ends (nothing staged on this K-tile) fewer DMAs are younger: drain.
  int age = 3;
  for (int j = 0;; ++j) {
    int nm0 = 0, nn0 = 0;
    const bool have_next = tile_of(j + 1, nm0, nn0);
    const char* nxtM = reinterpret_cast<const char*>(A) + (size_t)nm0 * K * 2;
    const char* nxtN = reinterpret_cast<const char*>(W) + (size_t)nn0 * K * 2;

    for (int t = 0; t < nk; t += 4) {
      // K-tiles t..t+3 sit in slots 0..3.  During K-tile t+s the stage of K-tile t+s+3 is issued into slot (s+3)&3:
      // s = 0 -> K-tile t+3 of this group (slot 3); s = 1..3 -> K-tiles t+4.. of the next group / next tile (slots 0..2)
      const bool tail = t + 4 >= nk;
      const bool more = !tail || have_next;  // K-tiles t+4.. exist in this block's stream
      const char* sM = tail ? nxtM : curM + (size_t)(t + 4) * 64;
      const char* sN = tail ? nxtN : curN + (size_t)(t + 4) * 64;
      const char* s3M = curM + (size_t)(t + 3) * 64;  // K-tile t+3 (always exists: nk % 4 == 0)
      const char* s3N = curN + (size_t)(t + 3) * 64;

#define S_KTILE(slot, ok, pM, pN, bias_stmt)                                                         \
  S_READ(F1, slot, 1)                                                                                \
  S_WAIT_PREV()                                                                                      \
  S_MFMA_DMA(F0, ok, pM, pN, ((slot) + 3) & 3, 0)                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
  if (!(ok)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
  else if (age == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(6 + S_NB + S_EPI_ST) : "memory");     \
  else if (age == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(6 + S_EPI_ST) : "memory");            \
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                              \
  S_FENCE();                                                                                         \
  __builtin_amdgcn_s_barrier();                                                                      \
  S_FENCE();                                                                                         \
  bias_stmt;                                                                                         \
  if ((slot) < 3 || more) { S_READ(F0, ((slot) + 1) & 3, 0) }                                        \
  S_MFMA_DMA(F1, ok, pM, pN, ((slot) + 3) & 3, 2)                                                    \
  if (age < 3) ++age;

      S_KTILE(0, true, s3M, s3N, (void)0)
      S_KTILE(1, more, sM, sN, (void)0)
      S_KTILE(2, more, sM + 64, sN + 64, (void)0)
      S_KTILE(3, more, sM + 128, sN + 128, if (tail) load_bias())
    }

    // ---- epilogue of this output tile, transposed through the wave's LDS scratch (the next tile's first four
    // K-tiles are in flight / landed and its first fragment set is in F0)
    S_WAIT_ALL()  // the next tile's first fragment set must have landed before hipcc may move/spill its registers
    // bias landed in the scratch: only pieces 2, 3 of the next tile's K-tile 2 were issued after its DMA
    if (have_next) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S_FENCE();
    if (DBG != 5) {
      unsigned char* scr = smem + S_SCRATCH + w * 4096;
      if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16) {
        bf16* yo = reinterpret_cast<bf16*>(outp) + (size_t)(m0 + wr * 128) * N + n0 + wc * 64;
        float4 b4[2][4];  // read before the first transposition pass overwrites the scratch (LDS ops stay in order)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[nt][g] = *reinterpret_cast<const float4*>(scr + (nt * 32 + 8 * g + 4 * hb) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = b4[nt][g];
              float v[4] = {acc[mt][nt][4 * g + 0] + bq.x, acc[mt][nt][4 * g + 1] + bq.y,
                            acc[mt][nt][4 * g + 2] + bq.z, acc[mt][nt][4 * g + 3] + bq.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (EPI == EPI_BIAS_QGELU_BF16) v[e] = quick_gelu(v[e]);
                if (EPI == EPI_BIAS_GELU_BF16) v[e] = gelu_erf(v[e]);
              }
              bf16x4 o;
              o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
              // row l31 = [8 chunks of 16 B]; chunk (4nt + g) holds columns 32nt + 8g .. +8, half hb
              *reinterpret_cast<bf16x4*>(scr + l31 * 128 + (((4 * nt + g) ^ (l31 & 7)) << 4) + hb * 8) = o;
            }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            const uint4 q = *reinterpret_cast<const uint4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
            if (DBG != 6) store16_wt(yo + (size_t)(mt * 32 + row) * N + rch * 8, q, flags & 4);
            else asm volatile("" ::"v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
          }
        }
      } else {
        // f32 output (in-place residual, or + table row): 32 x 32 sub-tile per pass, 128 B per row
        float* xo = reinterpret_cast<float*>(outp) + (size_t)(m0 + wr * 128) * N + n0 + wc * 64;
        float4 b4[2];
        if (HAS_BIAS) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) b4[nt] = *reinterpret_cast<const float4*>(scr + (nt * 32 + rch * 4) * 4);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            float4 ext[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = mt * 32 + 8 * i + rrow;
              if (EPI == EPI_BIAS_RESID_F32)
                ext[i] = *reinterpret_cast<const float4*>(xo + (size_t)row * N + nt * 32 + rch * 4);
              else
                ext[i] = *reinterpret_cast<const float4*>(table + (size_t)((m0 + wr * 128 + row) % T) * N + n0 + wc * 64 +
                                                          nt * 32 + rch * 4);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                                           acc[mt][nt][4 * g + 3]);
              // columns 8g + 4hb .. +4 = 16-B chunk 2g + hb of row l31
              *reinterpret_cast<float4*>(scr + l31 * 128 + (((2 * g + hb) ^ (l31 & 7)) << 4)) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 8 * i + rrow;
              float4 q = *reinterpret_cast<const float4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
              // same association as the 128x128 kernel, x + (acc + bias), so a row's result does not depend on
              // which kernel (i.e. which batch chunking) produced it
              if (EPI == EPI_BIAS_RESID_F32) {
                const float4 bq = b4[nt];
                q.x += bq.x; q.y += bq.y; q.z += bq.z; q.w += bq.w;
              }
              q.x = ext[i].x + q.x; q.y = ext[i].y + q.y; q.z = ext[i].z + q.z; q.w = ext[i].w + q.w;
              store16_wt(xo + (size_t)(mt * 32 + row) * N + nt * 32 + rch * 4, make_uint4(__float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z), __float_as_uint(q.w)), flags & 4);
            }
          }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    if (!have_next) break;
    age = 0;
    m0 = nm0;
    n0 = nn0;
    curM = nxtM;
    curN = nxtN;
  }
}

template <int EPI, int DBG = 0>
static hipError_t launch_r4_epi(const GemmArgs& g, int grid, hipStream_t st) {
  const size_t smem = S_SCRATCH + 8 * 4096;  // 160 KiB: the whole LDS of the CU
  auto kern = gemm256r4_kernel<EPI, DBG>;
  const char* fl = getenv("CLIPX_GEMM_FLAGS");  // A/B switch: bit 2 = plain (L2-resident) output stores
  const int flags = fl ? atoi(fl) : 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.N, g.K, g.M / 256,
                     g.N / 256, flags);
  return hipGetLastError();
}

hipError_t launch_gemm256r4(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (g.M <= 0 || g.M % 256 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K <= 0) return hipErrorInvalidValue;
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;  // one workgroup per CU; multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  if (g.epi == EPI_BIAS_BF16) {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) return launch_r4_epi<EPI_BIAS_BF16, 1>(g, grid, st);
    if (d == 5) return launch_r4_epi<EPI_BIAS_BF16, 5>(g, grid, st);
    if (d == 6) return launch_r4_epi<EPI_BIAS_BF16, 6>(g, grid, st);
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_r4_epi<EPI_BIAS_BF16>(g, grid, st);
    case EPI_BIAS_QGELU_BF16: return launch_r4_epi<EPI_BIAS_QGELU_BF16>(g, grid, st);
    case EPI_BIAS_GELU_BF16: return launch_r4_epi<EPI_BIAS_GELU_BF16>(g, grid, st);
    case EPI_BIAS_RESID_F32: return launch_r4_epi<EPI_BIAS_RESID_F32>(g, grid, st);
    case EPI_TABLE_F32: return launch_r4_epi<EPI_TABLE_F32>(g, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
