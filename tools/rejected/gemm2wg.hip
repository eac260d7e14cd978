// gemm2wg.hip -- bf16 GEMM as TWO independent 128x256 workgroups per CU (GemmArgs.variant == 5).
//
// STATUS: measured and rejected (round 2); built only into the tools library (make ablate), never into libclipx.so.
// Bitwise equal to the other kernels on every shape and epilogue (profiles/r02h_2wg.log), but 25-35 % SLOWER than
// gemm256sp: two workgroups per CU each stage their own operand tiles (+50 % bytes through the L1), and 76 KiB of LDS per
// workgroup only has room for 32-deep K-tiles, whose rows are 64 B -- half a cache line per request.  Per 64-deep K-tile
// and CU that is 1536 L2 requests instead of 512; an XCD's 16 L2 channels take one request per clock each, so 32 CUs need
// 3072 channel-cycles per K-tile against 2048 cycles of MFMA time: request-rate bound (measured 1.33x the time).  The
// start-up delay that should have put the two workgroups half a tile apart changes nothing.  Kept as the reproducible
// record of that experiment.
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A bf16 [M, K] activations, W bf16 [N, K] (torch Linear)
//
// The linear layers inside `model.encode_image/encode_text` (reference clip_retrieval/clip_inference/mapper.py:57,65).
//
// Why a second persistent kernel next to gemm256sp.hip: there all 8 waves of a CU reach the tile boundary together, and
// so do all 256 CUs -- every matrix pipe idles while 128 KiB (bf16 outputs) or 640 KiB (f32 residual read-modify-write +
// bf16 shadow) per CU go through the memory system in one burst (out-proj: 37 k of 100 k cycles per tile).  Here a CU
// holds two workgroups of 4 waves that run half a tile apart: while one is in its epilogue the other owns the matrix
// pipe.  Same wave tile (128 x 64 = 4x2 v_mfma_f32_32x32x16_bf16), same k order, same epilogue association as gemm256sp
// and the 128x128 kernel: a row's result does not depend on which kernel produced it (bitwise, tools/gemm_bench).
//
//   * workgroup = 256 threads = 4 waves side by side (wave w owns columns 64 w .. +64 of the 128 x 256 tile), persistent
//     over output tiles as one continuous stream of 32-deep K-tiles.
//   * LDS per workgroup 76 KiB (two fit in the CU's 160 KiB): a ring of three K-tile stages of 24 KiB (A 128 rows x 64 B,
//     then W 256 rows x 64 B, 16-B chunk position ^= (row >> 2) & 3: conflict-free ds_read_b128) + 1 KiB per wave for
//     the bias / row scales / prefetch sink.  Stages are filled by LDS-DMA, 6 global_load_lds_dwordx4 per wave per K-tile
//     (wave w fills bytes [6144 w, 6144 w + 6144) of the stage: that region is also its epilogue scratch, so no barrier
//     is needed between the epilogue and the next stage).
//   * K-tile t: {read k-step 1; MFMA k-step 0}; lgkmcnt(0), vmcnt (K-tile t+1 landed), s_barrier; {read k-step 0 of
//     K-tile t+1; MFMA k-step 1}; stage K-tile t+3 into the slot just released.  The barrier sits before the last
//     k-step's MFMAs as in gemm256sp: the slot is free while a quarter of the K-tile's math is still to be issued.
//
// Requirements: M % 128 == 0, N % 256 == 0, K % 128 == 0.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace clipx {

constexpr int W_STAGE = 24576;            // bytes of one ring stage
constexpr int W_NOFF = 8192;              // W rows inside a stage (behind the 128 A rows)
constexpr int W_AUX = 3 * W_STAGE;        // per wave 1 KiB: bias 256 B | row scales 512 B | prefetch sink 256 B
constexpr int W_LDS = W_AUX + 4 * 1024;   // 77824 B

#define W_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef CLIPX_STORE_GUARD_NOPS
#define CLIPX_STORE_GUARD_NOPS 7
#endif
#define W_STR2(x) #x
#define W_STR(x) W_STR2(x)
// see gemm256sp.hip: the data registers of a buffer store with a register soffset must stay live behind it
#define W_STORE_GUARD(v) asm volatile("s_nop " W_STR(CLIPX_STORE_GUARD_NOPS) ::"v"(v))

typedef unsigned w_u32x4 __attribute__((ext_vector_type(4)));
typedef int w_i32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2wg_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                        const float* __restrict__ bias, void* __restrict__ outp,
                                                        const float* __restrict__ table, int T, int N, int K, int ntm,
                                                        int ntn, const float* __restrict__ rowscale, bf16* __restrict__ out16,
                                                        int delay) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = lane >> 5, l31 = lane & 31;
  const int ntiles = ntm * ntn;

  // ---- tile list, XCD-aware like gemm256sp's raster 2: the 64 workgroups of an XCD (blockIdx % 8) work on 16 m-tiles x
  // 4 n-tiles at a time (A panels 4 MiB + W panels 2 MiB at K = 1024); an XCD owns its m-groups and walks their n-slices in
  // consecutive rounds
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  auto tile_of = [&](int j, int& m0, int& n0) -> bool {
    if (wpx == 64 && (ntn & 3) == 0 && (ntm & 127) == 0) {
      const int nsl = ntn >> 2, ng = ntm >> 7;
      if (j >= nsl * ng) return false;
      const int sl = j % nsl, gi = j / nsl;
      m0 = ((gi * 8 + xcd) * 16 + (idx & 15)) * 128;
      n0 = (sl * 4 + (idx >> 4)) * 256;
      return true;
    }
    const int logical = (j * 8 + xcd) * wpx + idx;
    if (logical >= ntiles) return false;
    const int per_group = 16 * ntn;
    const int grp = logical / per_group, within = logical - grp * per_group;
    const int gm0 = grp * 16;
    const int gsz = (ntm - gm0) < 16 ? (ntm - gm0) : 16;
    m0 = (gm0 + within % gsz) * 128;
    n0 = (within / gsz) * 256;
    return true;
  };
  int m0, n0;
  if (!tile_of(0, m0, n0)) return;  // before any barrier

  // the second workgroup of every CU starts `delay` x 8 k cycles late (half a tile): from then on the two run out of phase
  if (blockIdx.x >= (gridDim.x >> 1))
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);

  // ---- staging.  Stage piece p (1 KiB = 16 rows x 64 B): p < 8 -> A rows 16 p .., else W rows 16 (p - 8) ..; wave w
  // fills pieces 6 w .. 6 w + 5.  Lane l writes LDS bytes 16 l .. of its piece = row l >> 2, chunk position l & 3, which
  // holds source chunk (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3).
  unsigned soff[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int p = 6 * w + j;
    const int row = (p < 8 ? p * 16 : (p - 8) * 16) + (lane >> 2);
    const int c = (lane & 3) ^ ((lane >> 4) & 3);
    soff[j] = (unsigned)((row * K + c * 8) * 2);
  }
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
  const unsigned dmw = lds_base + w * 6144;  // this wave's first piece inside a stage
  // pieces 0, 1 of wave w come from A when w <= 1, pieces 2..5 when w == 0
#define W_DMA(off, base, mbase, cimm)                                                                         \
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(mbase), "n"(cimm) \
               : "memory", "scc")
#define W_STAGE_KT(slot, pLo, pHi)                   \
  {                                                  \
    const unsigned mb_ = dmw + (slot);               \
    W_DMA(soff[0], pLo, mb_, 0);                     \
    W_DMA(soff[1], pLo, mb_, 1024);                  \
    W_DMA(soff[2], pHi, mb_, 2048);                  \
    W_DMA(soff[3], pHi, mb_, 3072);                  \
    W_DMA(soff[4], pHi, mb_, 4096);                  \
    W_DMA(soff[5], pHi, mb_, 5120);                  \
  }                                                  \
  W_FENCE();

  // ---- fragment read addresses without the slot offset: one per (operand, k-step)
  const int sw = (l31 >> 2) & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned fA[2], fW[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int xk = ((2 * kk + hb) ^ sw) << 4;
    fA[kk] = lds0 + l31 * 64 + xk;                       // + slot + mi * 2048
    fW[kk] = lds0 + W_NOFF + (w * 64 + l31) * 64 + xk;   // + slot + ni * 2048
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  w_i32x4 F0[6], F1[6];  // [0..3] A fragments (mi), [4..5] W fragments (ni) of one k-step
#define W_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define W_READ(F, aA, aW)          \
  W_DSREAD(F[4], aW, 0);           \
  W_DSREAD(F[5], aW, 2048);        \
  W_DSREAD(F[0], aA, 0);           \
  W_DSREAD(F[1], aA, 2048);        \
  W_DSREAD(F[2], aA, 4096);        \
  W_DSREAD(F[3], aA, 6144);        \
  W_FENCE();
#define W_WAIT_PREV() asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); W_FENCE();
#define W_MFMA(F)                                                                                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0); \
  W_FENCE();
#define W_SYNC(vm)                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(vm) : "memory");  \
  W_FENCE();                                                 \
  __builtin_amdgcn_s_barrier();                              \
  W_FENCE();

  const int nk = K >> 5;  // 32-deep K-tiles per output tile (>= 4)
  constexpr bool OUT_BF16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16;
  constexpr bool HAS_BIAS = EPI != EPI_TABLE_F32;
  // VMEM operations a wave issues besides the stages (vmcnt retires in order): the bias / row scale DMAs of a tile (behind
  // its first K-tile) and the epilogue's stores that may still be in flight at the next tile's first sync (bf16 outputs: 16
  // stores; f32 residual: at least its last 32; table epilogue: compiled code, waited for)
  constexpr int NB = HAS_BIAS ? (OUT_BF16 ? 3 : 1) : 0;
  // L2 prefetch (see gemm256sp.hip): every K-tile each wave touches 6 lines of the 64-deep K-tile PFD ahead with one dword
  // LDS-DMA into its sink; even K-tiles take rows 0..5 of the wave's 12, odd ones rows 6..11.  The workgroups of the XCD that
  // share a panel split it: 32 of the 128 A rows (4 workgroups per m-tile), 16 of the 256 W rows (16 per n-tile).
  constexpr int PFD = 4;
  constexpr int NPF = 1;
  constexpr int NE = OUT_BF16 ? 16 : (EPI == EPI_BIAS_RESID_F32 ? 32 : 0);

  // ring slots (byte offsets of the stage holding K-tile t, t+1, t+2 of the stream)
  unsigned sc = 0, s1 = W_STAGE, s2 = 2 * W_STAGE;
  // stream of K-tiles to stage: pointers of the next one (pieces from A / from W per wave, see above), how many are left in
  // its tile, and the tile that follows
  const char* tA = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* tW = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;
  const char* gLo = (w <= 1) ? tA : tW;
  const char* gHi = (w == 0) ? tA : tW;
  W_STAGE_KT(sc, gLo, gHi)
  W_STAGE_KT(s1, gLo + 64, gHi + 64)
  W_STAGE_KT(s2, gLo + 128, gHi + 128)
  gLo += 192;
  gHi += 192;
  int left = nk - 3;  // K-tiles of the current source tile still to stage
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W_FENCE();
  __builtin_amdgcn_s_barrier();
  W_FENCE();
  unsigned aA0 = fA[0] + sc, aW0 = fW[0] + sc, aA1 = fA[1] + sc, aW1 = fW[1] + sc;
  W_READ(F0, aA0, aW0)

  const unsigned aux_m0 = lds_base + W_AUX + w * 1024;
  const unsigned lane4 = (unsigned)(lane * 4);
  const int rrow = lane >> 3, rch = lane & 7;  // epilogue read-back: row 8i + rrow, 16-B chunk rch

  const int nk64 = K >> 6;
  const bool pf_run = nk64 >= 8 && (size_t)ntm * 128 * K * 2 >= ((size_t)64 << 20);
  auto pf_addr = [&](int tm0, int tn0, int par) -> const char* {
    const int G = w * 12 + (lane % 6) + 6 * par;
    return G < 32 ? reinterpret_cast<const char*>(A) + (size_t)(tm0 + 32 * ((idx >> 4) & 3) + G) * K * 2
                  : reinterpret_cast<const char*>(W) + (size_t)(tn0 + 16 * (idx & 15) + (G - 32)) * K * 2;
  };
  const unsigned pf_m0 = aux_m0 + 768;
  const int pf_step = pf_run ? 128 : 0;
  const char* pfE = pf_addr(m0, n0, 0) + (pf_run ? PFD * 128 : 0);  // used at even K-tiles
  const char* pfO = pf_addr(m0, n0, 1) + (pf_run ? PFD * 128 : 0);  // at odd ones
  int pf_left = pf_run ? nk64 - PFD : 0x7fffffff;  // 64-deep K-tiles of the current panel still to touch (per parity)
#define W_PF_ISSUE(ptr) \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(ptr), "s"(pf_m0) : "memory"); \
  W_FENCE();
  // K-tile parity alternates along the stream (nk is even): `podd` is the parity of the K-tile whose tail runs
#define W_PF(issue)                                                     \
  {                                                                     \
    if (!podd) {                                                        \
      W_PF_ISSUE(pfE)                                                   \
      pfE += pf_step;                                                   \
    } else {                                                            \
      if (issue) { W_PF_ISSUE(pfO) }                                    \
      pfO += pf_step;                                                   \
      if (--pf_left == 0) { /* both parities touched this panel's last K-tile: on to the next tile's panels */ \
        const bool nx_ = more_c;                                        \
        pfE = pf_addr(nx_ ? cm0 : m0, nx_ ? cn0 : n0, 0);               \
        pfO = pf_addr(nx_ ? cm0 : m0, nx_ ? cn0 : n0, 1);               \
        pf_left = nk64;                                                 \
      }                                                                 \
    }                                                                   \
    podd = !podd;                                                       \
  }
  bool podd = false;
  int cm0 = 0, cn0 = 0;
  bool more_c = tile_of(1, cm0, cn0);  // the tile this workgroup computes next
  int jc = 0;  // index (in this workgroup's list) of the tile being computed
  int jn = 1;  // index of the tile the staging stream moves to next
  int nm0 = 0, nn0 = 0;
  bool more = tile_of(jn, nm0, nn0);
  // stage the stream's next K-tile into `slot` (nothing once the last tile's K-tiles are all staged)
#define W_STAGE_NEXT(slot)                                                         \
  {                                                                                \
    if (left == 0 && more) {                                                       \
      tA = reinterpret_cast<const char*>(A) + (size_t)nm0 * K * 2;                 \
      tW = reinterpret_cast<const char*>(W) + (size_t)nn0 * K * 2;                 \
      gLo = (w <= 1) ? tA : tW;                                                    \
      gHi = (w == 0) ? tA : tW;                                                    \
      left = nk;                                                                   \
      ++jn;                                                                        \
      more = tile_of(jn, nm0, nn0);                                                \
    }                                                                              \
    /* behind the end of the list the last K-tile is staged again (into a free slot, never read): every K-tile issues  */ \
    /* its 6 DMAs, so the in-order vmcnt arithmetic of the syncs holds up to the last tile without a drain path        */ \
    const int adv_ = left > 0 ? 64 : 0;                                            \
    W_STAGE_KT(slot, gLo + (adv_ - 64), gHi + (adv_ - 64))                             \
    gLo += adv_;                                                                   \
    gHi += adv_;                                                                   \
    left -= adv_ >> 6;                                                             \
  }
#define W_ROTATE()             \
  {                            \
    const unsigned o_ = sc;    \
    sc = s1;                   \
    s1 = s2;                   \
    s2 = o_;                   \
    aA0 = fA[0] + sc;          \
    aW0 = fW[0] + sc;          \
    aA1 = fA[1] + sc;          \
    aW1 = fW[1] + sc;          \
  }
  // one K-tile that is not the last of its tile: F0 holds its k-step 0
#define W_KT(vm, extra)                                   \
  W_READ(F1, aA1, aW1)                                    \
  W_WAIT_PREV()                                           \
  W_MFMA(F0)                                              \
  W_SYNC(vm)                                              \
  {                                                       \
    const unsigned nA_ = fA[0] + s1, nW_ = fW[0] + s1;    \
    W_READ(F0, nA_, nW_)                                  \
  }                                                       \
  W_MFMA(F1)                                              \
  extra                                                   \
  W_STAGE_NEXT(sc)                                        \
  W_PF(true)                                              \
  W_ROTATE()

  auto load_bias = [&]() {
    if (!HAS_BIAS) return;
    const float* bp = bias + n0 + w * 64;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(bp), "s"(aux_m0) : "memory");
    if (OUT_BF16) {
      const float* rp = rowscale + m0;
      asm volatile("s_add_u32 m0, %2, 256\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(rp), "s"(aux_m0) : "memory", "scc");
      const float* rp2 = rp + 64;
      asm volatile("s_add_u32 m0, %2, 512\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(rp2), "s"(aux_m0) : "memory", "scc");
    }
  };

  for (;;) {
    // ---- K-tile 0: the epilogue's stores (NE) and the stage issued behind it may still be in flight
    W_KT(NE + 6 + 2 * NPF, load_bias(); W_FENCE();)
    // ---- K-tile 1: the bias DMAs are younger than the stage it waits for
    W_KT(NB + 6 + 2 * NPF, )
    // ---- steady state
    for (int t = 2; t < nk - 1; ++t) {
      W_KT(6 + 2 * NPF, )
    }
    // ---- last K-tile of the tile: no pre-read (the epilogue needs the registers), nothing staged before the epilogue
    // (the released slot is its scratch)
    W_READ(F1, aA1, aW1)
    W_WAIT_PREV()
    W_MFMA(F0)
    W_SYNC(6 + 2 * NPF)
    W_MFMA(F1)
    W_PF(false)  // nk is even: the last K-tile is an odd one; no prefetch behind it (the f32 epilogue's first wait would
                 // sit out its miss), the stream pointer still moves on

    // ---- epilogue, transposed through this wave's part of the released slot
    W_FENCE();
    {
      unsigned char* scr = smem + sc + w * 6144;
      const unsigned char* aux = smem + W_AUX + w * 1024;
      if (OUT_BF16) {
        const char* yb = reinterpret_cast<const char*>(outp) + ((size_t)m0 * N + n0 + w * 64) * 2;
        const unsigned voff = (unsigned)((rrow * N + rch * 8) * 2);
        const size_t rstep = (size_t)8 * N * 2;  // 8 rows
        float4 b4[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[nt][g] = *reinterpret_cast<const float4*>(aux + (nt * 32 + 8 * g + 4 * hb) * 4);
        float rr[4];  // row scales of the lane's rows 32 mt + l31
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) rr[mt] = *reinterpret_cast<const float*>(aux + 256 + (mt * 32 + l31) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = b4[nt][g];
              // one fma per element, like gemm_store_quad (gemm_common.h): bit-identical rows from all kernels
              float v[4] = {__builtin_fmaf(acc[mt][nt][4 * g + 0], rr[mt], bq.x), __builtin_fmaf(acc[mt][nt][4 * g + 1], rr[mt], bq.y),
                            __builtin_fmaf(acc[mt][nt][4 * g + 2], rr[mt], bq.z), __builtin_fmaf(acc[mt][nt][4 * g + 3], rr[mt], bq.w)};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (EPI == EPI_BIAS_QGELU_BF16) v[e] = quick_gelu(v[e]);
                if (EPI == EPI_BIAS_GELU_BF16) v[e] = gelu_erf(v[e]);
              }
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const bf16x2_t o01 = __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t);
              const bf16x2_t o23 = __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t);
              uint2 o;
              o.x = __builtin_bit_cast(unsigned, o01);
              o.y = __builtin_bit_cast(unsigned, o23);
              *reinterpret_cast<uint2*>(scr + l31 * 128 + (((4 * nt + g) ^ (l31 & 7)) << 4) + hb * 8) = o;
            }
          w_u32x4 q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            q[i] = *reinterpret_cast<const w_u32x4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const char* base = yb + (size_t)(mt * 4 + i) * rstep;
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(q[i]), "s"(base) : "memory");
          }
        }
      } else if (EPI == EPI_BIAS_RESID_F32) {
        char* xb = reinterpret_cast<char*>(outp) + ((size_t)m0 * N + n0 + w * 64) * 4;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 0x7ffffffe, 0x00020000);
        const int voff = (rrow * N + rch * 4) * 4;
        const int rstep = 8 * N * 4;  // 8 rows
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        bf16* xb16 = out16 ? out16 + ((size_t)m0 * N + n0 + w * 64) : reinterpret_cast<bf16*>(xb);
        const __amdgpu_buffer_rsrc_t xr16 = __builtin_amdgcn_make_buffer_rsrc(xb16, 0, 0x7ffffffe, 0x00020000);
        const bool odd = (rch & 1) != 0;
        const int voff16 = rrow * N * 2 + (odd ? 64 + (rch - 1) * 8 : rch * 8);
        const int rstep16 = 8 * N * 2;
        const bool shadow = out16 != nullptr;
        u32x2 hkeep[4];
        float4 b4[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b4[nt] = *reinterpret_cast<const float4*>(aux + (nt * 32 + rch * 4) * 4);
        W_FENCE();
        w_u32x4 ext[2][4];
#define W_LD_EXT(set, p)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    ext[set][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (((p) >> 1) * 4 + i) * rstep + ((p) & 1) * 128, 0);
        W_LD_EXT(0, 0)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int mt = p >> 1, nt = p & 1;
          if (p < 7) { W_LD_EXT((p + 1) & 1, p + 1) }
          W_FENCE();
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                                         acc[mt][nt][4 * g + 3]);
            *reinterpret_cast<float4*>(scr + l31 * 128 + (((2 * g + hb) ^ (l31 & 7)) << 4)) = v;
          }
          float4 q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            q[i] = *reinterpret_cast<const float4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 bq = b4[nt];
            const w_u32x4 e = ext[p & 1][i];
            // same association as the other kernels: x + (acc + bias)
            float4 o = q[i];
            o.x += bq.x; o.y += bq.y; o.z += bq.z; o.w += bq.w;
            o.x = __uint_as_float(e[0]) + o.x; o.y = __uint_as_float(e[1]) + o.y;
            o.z = __uint_as_float(e[2]) + o.z; o.w = __uint_as_float(e[3]) + o.w;
            const w_u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            __builtin_amdgcn_raw_buffer_store_b128(ov, xr, voff, (mt * 4 + i) * rstep + nt * 128, 0);
            W_STORE_GUARD(ov);
            if (shadow) {
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const bf16x2_t h01 = __builtin_convertvector((f32x2_t){o.x, o.y}, bf16x2_t);
              const bf16x2_t h23 = __builtin_convertvector((f32x2_t){o.z, o.w}, bf16x2_t);
              const u32x2 hv = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
              if (nt == 0) {
                hkeep[i] = hv;
              } else {
                const u32x2 send = odd ? hkeep[i] : hv;
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, false);
                const w_u32x4 full = odd ? w_u32x4{r0, r1, hv.x, hv.y} : w_u32x4{hkeep[i].x, hkeep[i].y, r0, r1};
                __builtin_amdgcn_raw_buffer_store_b128(full, xr16, voff16, (mt * 4 + i) * rstep16, 0);
                W_STORE_GUARD(full);
              }
            }
          }
          W_FENCE();
        }
      } else {
        // + table row (patch embedding: class token / positional rows), f32 out: once per forward, plain code
        float* xo = reinterpret_cast<float*>(outp) + (size_t)m0 * N + n0 + w * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            float4 ext[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = mt * 32 + 8 * i + rrow;
              ext[i] = *reinterpret_cast<const float4*>(table + (size_t)((m0 + row) % T) * N + n0 + w * 64 + nt * 32 + rch * 4);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                                           acc[mt][nt][4 * g + 3]);
              *reinterpret_cast<float4*>(scr + l31 * 128 + (((2 * g + hb) ^ (l31 & 7)) << 4)) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 8 * i + rrow;
              float4 q = *reinterpret_cast<const float4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
              q.x = ext[i].x + q.x; q.y = ext[i].y + q.y; q.z = ext[i].z + q.z; q.w = ext[i].w + q.w;
              *reinterpret_cast<float4*>(xo + (size_t)(mt * 32 + row) * N + nt * 32 + rch * 4) = q;
            }
          }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
    W_FENCE();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    // next tile of this workgroup (its K-tiles 0, 1 are already in the ring)
    if (!more_c) break;
    m0 = cm0;
    n0 = cn0;
    ++jc;
    more_c = tile_of(jc + 1, cm0, cn0);
    // the epilogue's LDS reads are done (in program order) before this wave's DMA overwrites its scratch region
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W_FENCE();
    W_STAGE_NEXT(sc)
    {  // keeps the count of the syncs: one prefetch-type operation behind every stage (the sink's own line here)
      const char* self_ = pfO - pf_step;
      W_PF_ISSUE(self_)
    }
    W_ROTATE()
    W_READ(F0, aA0, aW0)
  }
}

template <int EPI>
static hipError_t launch_2wg_epi(const GemmArgs& g, int grid, int delay, hipStream_t st) {
  auto kern = gemm2wg_kernel<EPI>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W_LDS, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.N, g.K, g.M / 128,
                     g.N / 256, g.rowscale, g.out16, delay);
  return hipGetLastError();
}

hipError_t launch_gemm2wg(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (g.M <= 0 || g.M % 128 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K <= 0) return hipErrorInvalidValue;
  int grid = 2 * ((n_cu > 0 ? n_cu : 256) & ~7);  // two workgroups per CU; multiple of the 8 XCDs
  if (grid < 16) grid = 16;
  // half a tile of head start for the first workgroup of every CU: a tile is K / 32 K-tiles of ~1.1 k cycles when two
  // workgroups share the matrix pipe; s_sleep 127 = 8128 cycles
  int delay = (int)((g.K / 32) * 1100 / 2 / 8128);
#ifdef CLIPX_ABLATE
  if (const char* fl = getenv("CLIPX_GEMM_FLAGS")) delay = atoi(fl);
  static bool told = false;
  if (!told && getenv("CLIPX_2WG_OCC")) {
    told = true;
    int nb = -1;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gemm2wg_kernel<EPI_BIAS_BF16>), 256, W_LDS);
    fprintf(stderr, "gemm2wg: occupancy query -> %d workgroups per CU (%s), LDS %d B\n", nb, hipGetErrorString(oe), W_LDS);
  }
#endif
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_2wg_epi<EPI_BIAS_BF16>(g, grid, delay, st);
    case EPI_BIAS_QGELU_BF16: return launch_2wg_epi<EPI_BIAS_QGELU_BF16>(g, grid, delay, st);
    case EPI_BIAS_GELU_BF16: return launch_2wg_epi<EPI_BIAS_GELU_BF16>(g, grid, delay, st);
    case EPI_BIAS_RESID_F32: return launch_2wg_epi<EPI_BIAS_RESID_F32>(g, grid, delay, st);
    case EPI_TABLE_F32: return launch_2wg_epi<EPI_TABLE_F32>(g, grid, delay, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
