// gemm256_tail.h -- the ragged 257th m-tile of the persistent 256x256 GEMM kernels (gemm256sp.hip: 8 waves, gemm256w4.hip: 4 waves),
// computed inside the same launch.  Internal; included by both kernel files.
#pragma once
#include "gemm_common.h"

namespace clipx {

#ifndef S_FENCE
#define S_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// ---- the tail: one ragged m-tile without a second launch -------------------------------------------------------------
// ViT-L/14 at bs 256 has M = 257 x 256 rows: 256 m-tiles fill the 256 CUs in whole rounds and the 257th used to be a separate
// launch of the 128x128 kernel -- 16 .. 64 workgroups on an otherwise idle chip, 14 - 34 us each, 144 launches = 2.25 ms per step
// (5 % of the GEMM time for 0.4 % of the rows).  Here every workgroup, when its tile list is done, takes ONE strip
// of those 256 rows: 32 rows x (NB x 32) columns (NB = 1 at N = 1024, 3 at 3072, 4 at 4096: 256 x N / 256 CUs), i.e. NB MFMA
// blocks of 32 x 32, one per wave, the full K loop each -- the accumulation order of every output element is the k-step order of
// the other kernels, so rows stay bit-identical whichever path computes them.  Operands arrive by LDS-DMA in fragment shape
// (a k-step of 32 rows = 1 KiB, lane-linear: conflict-free ds_read_b128) through a ring of R stages of 8 k-steps; wave w fetches
// k-step w of every operand of a stage, waves < NB multiply.  One barrier per stage.  The epilogue is gemm_store_quad (the code
// of the 128x128 kernel).
template <int EPI, bool F16, int NB, int NW = 8>  // NW: waves of the workgroup (8: gemm256sp.hip, 4: gemm256w4.hip -- each wave then fetches two pieces per operand)
__device__ __forceinline__ void gemm256_tail(const bf16* __restrict__ A, const bf16* __restrict__ W, const float* __restrict__ bias,
                                             void* __restrict__ outp, const float* __restrict__ table, int T, int N, int K,
                                             const float* __restrict__ rowscale, bf16* __restrict__ out16, int tail_m0,
                                             unsigned char* smem, unsigned lds_base, int w, int lane) {
  constexpr int NOP = 1 + NB;     // operands of a stage: the 32 activation rows and NB blocks of 32 weight rows
  constexpr int SB = NOP * 8192;  // stage = 8 k-steps x 1 KiB per operand
  constexpr int R = 131072 / SB;  // ring depth: 8 / 5 / 4 / 3 stages in the 128 KiB the K-tile buffers occupied
  const int ncg = (N >> 5) / NB;  // column groups of a 32-row strip
  const int strip = blockIdx.x;
  if (strip >= 8 * ncg) return;   // (uniform per workgroup; the host makes sure every strip has a workgroup)
  const int rb = strip & 7, cg = strip >> 3;
  const int tm = tail_m0 + rb * 32, tn = cg * NB * 32;
  const int l31 = lane & 31, hb = lane >> 5;
#if CLIPX_MFMA16
  // 16x16x32 form: a stage (128 k) is 4 slabs x 2 row halves = 8 pieces of 1 KiB per 32-row operand; piece p = 2 * slab + half is a
  // fragment as the MFMA wants it: lane (l15, q4) holds row 16 * half + l15, k = 32 * slab + 8 * q4 .. + 8.  Wave w fetches piece w.
  const unsigned voff = (unsigned)(((lane & 15) * K + 8 * (lane >> 4)) * 2);
  const char* baseA = reinterpret_cast<const char*>(A) + (size_t)tm * K * 2;  // + piece: (16 * (p & 1)) rows, (p >> 1) * 64 bytes
  const char* baseW = reinterpret_cast<const char*>(W) + (size_t)tn * K * 2;
#define T_PIECE(p) ((size_t)(16 * ((p) & 1)) * K * 2 + ((p) >> 1) * 64)
  (void)l31; (void)hb;
#else
  const unsigned voff = (unsigned)((l31 * K + 8 * hb) * 2);  // row l31 of the operand, 16 B of the k-step
  const char* baseA = reinterpret_cast<const char*>(A) + (size_t)tm * K * 2;  // + piece * 32: the piece's k-step in a stage
  const char* baseW = reinterpret_cast<const char*>(W) + (size_t)tn * K * 2;
#define T_PIECE(p) ((size_t)(p) * 32)
#endif
  constexpr int PPW = 8 / NW;  // pieces per wave, operand and stage
  const int nch = K >> 7;
#define T_DMA(off, base, dst) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory")
  auto issue = [&](int c, int slot) {
    const int cc = c < nch ? c : nch - 1;  // past the end: reload the last stage (keeps the vmcnt arithmetic uniform)
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
      const int p = w + NW * pp;  // wave w fetches piece(s) w (, w + 4) of every operand of a stage
      const unsigned dst = lds_base + slot * SB + p * 1024;
      T_DMA(voff, baseA + T_PIECE(p) + (size_t)cc * 256, dst);
#pragma unroll
      for (int o = 0; o < NB; ++o) T_DMA(voff, baseW + T_PIECE(p) + (size_t)o * 64 * K + (size_t)cc * 256, dst + (1 + o) * 8192);
    }
    S_FENCE();
  };
#pragma unroll
  for (int c = 0; c < R - 1; ++c) issue(c, c);
#if CLIPX_MFMA16
  f32x4 acc[4];  // the four 16 x 16 quads of the wave's 32 x 32 block
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#define T_ACC(g, e) acc[g][e]
#else
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#define T_ACC(g, e) acc[4 * (g) + (e)]
#endif
  int slot = 0;
  for (int c = 0; c < nch; ++c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * NOP * PPW) : "memory");  // this wave's share of stage c has landed
    S_FENCE();
    __builtin_amdgcn_s_barrier();  // ... and everyone's; every wave is past stage c - 1, whose slot the refill takes
    S_FENCE();
    issue(c + R - 1, slot == 0 ? R - 1 : slot - 1);
    if (w < NB) {
      const unsigned char* st = smem + slot * SB + lane * 16;
#if CLIPX_MFMA16
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const frag_t af0 = *reinterpret_cast<const frag_t*>(st + (2 * sl) * 1024);
        const frag_t af1 = *reinterpret_cast<const frag_t*>(st + (2 * sl + 1) * 1024);
        const frag_t wf0 = *reinterpret_cast<const frag_t*>(st + (1 + w) * 8192 + (2 * sl) * 1024);
        const frag_t wf1 = *reinterpret_cast<const frag_t*>(st + (1 + w) * 8192 + (2 * sl + 1) * 1024);
        mfma_block16<F16>(acc, wf0, wf1, af0, af1);
      }
#else
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const frag_t af = *reinterpret_cast<const frag_t*>(st + ks * 1024);
        const frag_t wf = *reinterpret_cast<const frag_t*>(st + (1 + w) * 8192 + ks * 1024);
        acc = mfma_32x32x16<F16>(wf, af, acc);
      }
#endif
    }
    slot = slot + 1 == R ? 0 : slot + 1;
  }
#undef T_DMA
#undef T_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail reloads still in flight
  S_FENCE();
  if (w < NB) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = tm + quad_m(g, lane);
      const int n = tn + w * 32 + quad_n(g, lane);
      const float4 v = make_float4(T_ACC(g, 0), T_ACC(g, 1), T_ACC(g, 2), T_ACC(g, 3));
      gemm_store_quad<EPI>(v, m, n, N, bias, outp, table, T, 0, rowscale, out16);
    }
  }
#undef T_ACC
}


}  // namespace clipx
