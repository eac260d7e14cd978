// gemm256w4.hip -- persistent 256x256x64 GEMM for gfx950 with FOUR waves per workgroup (one per SIMD), wave tile 128 x 128 as 8 x 8
// blocks of v_mfma_f32_16x16x32_{bf16,f16}: 256 accumulator registers per lane (the AGPR half of the 512-entry file), the shape the
// vendor's assembly kernel and tools/mfma_power_probe both sustain more TFLOP/s with on this part.
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A [M, K] activations, W [N, K] weights (torch Linear), both bf16 or both fp16
//
// The linear layers inside `model.encode_image/encode_text` (reference clip_retrieval/clip_inference/mapper.py:57,65).
//
// Against gemm256sp.hip (8 waves x 128 x 64):
//   * a K-tile moves 128 KiB of fragments LDS -> registers instead of 192 KiB (every fragment feeds 8 MFMAs instead of 4 / 8);
//   * the fragment pipeline is DEEP instead of covered by a SIMD partner: a K-tile is 16 units of 8 MFMAs (unit u = activation block
//     u & 7 of slab u >> 3 against the slab's eight weight fragments); the activation fragment of unit u + 6 and (in units 2..5 /
//     10..13) two weight fragments of the next slab are requested at the top of unit u -- 768 matrix-pipe cycles ahead of their use;
//   * the LDS-DMA of K-tile t + 2 is spread over units 10..15 of K-tile t, one 1-KiB piece behind every fourth MFMA, so that the
//     wave never queues several pieces at the texture addresser at once (a piece issues in ~16 cycles when the addresser is free and
//     blocks the wave's MFMA issue for 60 - 180 cycles when eight are queued back to back: MI355X_MICROARCH, "LDS-DMA piece issue
//     cost");
//   * ONE s_barrier per K-tile, behind unit 9: by then every fragment read of the K-tile's buffer has been issued AND has returned
//     (lgkmcnt(0); the youngest is 256 matrix-pipe cycles old), the next K-tile has landed (vmcnt), and the buffer is released to the DMA.
//   * epilogues: sixteen passes of 16 rows x 64 columns; the f32 accumulators go from the AGPRs straight into wave-private LDS
//     (ds_write_b128 with AGPR data, chunk-XOR swizzled), come back in store layout and are scaled / biased / activated / added to the
//     residual there; every global access is 16 B per lane with 8 consecutive lanes covering one full 128-B line.
// Accumulation order of an output element = ascending 32-deep k-slabs, like every other GEMM kernel of this library (gemm_common.h):
// rows are bit-identical whichever kernel computes them.
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0, K >= 256; 16-bit-output epilogues and the fp16 in-place residual.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"
#include "gemm256_tail.h"

namespace clipx {

constexpr int W4_OPB = 32768;      // bytes of one operand tile ([256 rows][128 B])
constexpr int W4_NBASE = 65536;    // N operand tiles start here
constexpr int W4_SCR = 131072;     // per-wave epilogue scratch starts here
constexpr int W4_SCRW = 8192;      // per wave: two 4-KiB epilogue pass buffers; between epilogues [4096, 5120) receives bias | row scales and
                                   // [7936, 8192) is the L2-prefetch sink

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W4_STR2(x) #x
#define W4_STR(x) W4_STR2(x)
#define W4_STORE_GUARD(v) asm volatile("s_nop 7" ::"v"(v))

__device__ long long g_w4_phase[2048 * 8];

typedef unsigned w4_u32x4 __attribute__((ext_vector_type(4)));
typedef int w4_i32x4 __attribute__((ext_vector_type(4)));

// DBG 16: phase timer (correct results): [0] epilogues, [1] first K-tile after an epilogue, [3] other K-tiles, [5] their number, [6] tiles, [7] kernel
// STATS (fp16 operands, 16-bit-output epilogues: the LayerNorm-folded QKV / fc1): the row scale of the epilogue is not read from
// `rowscale` but computed here -- 1 / sqrt(var(A[m, :]) + stats_eps) over the K = row-length columns of the A rows the workgroup
// holds, from the A fragments of the K loop, in the canonical order of gemm_common.h (ln_rstd_onepass): per unit one fragment
// (16 B per lane) goes through 4 + 4 v_dot2_f32_f16, one behind each MFMA of the unit, into 16 running sums per lane.  (The rows
// of the ragged tail tile still take theirs from `rowscale`.)
template <int EPI, bool F16, int DBG, bool STATS = false>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ outp, int N, int K,
                                                          int ntm, int ntn, const float* __restrict__ rowscale, int tail_m0, int tail_nb, int stagger,
                                                          float stats_eps, int* __restrict__ range_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // DBG (tools build only): 16 = phase timer, results correct.  19 .. 23 = the timer + an ABLATION of the K loop (results are garbage,
  // the timings say what each ingredient costs): 19 no operand DMAs, 20 no fragment reads, 21 neither, 22 no s_barrier, 23 no L2 prefetch
  constexpr bool DBG_TIMER = DBG >= 16;
  constexpr bool AB_NODMA = DBG == 19 || DBG == 21, AB_NOREAD = DBG == 20 || DBG == 21, AB_NOBAR = DBG == 22, AB_NOPF = DBG == 23;
  constexpr bool OUT16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_F16;
  static_assert(OUT16 || EPI == EPI_BIAS_RESID_H16, "epilogue not built for the 4-wave kernel");
  static_assert(!STATS || (OUT16 && F16), "in-kernel LayerNorm statistics: fp16 operands and a 16-bit-output epilogue");
  constexpr int PFD = 4;  // L2 prefetch distance in K-tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int ntiles = ntm * ntn;

  // ---- tile list of this block: the raster of gemm256sp.hip (an XCD's 32 workgroups walk 8 m-tiles x 4 n-tiles together)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  auto tile_of = [&](int j, int& m0, int& n0, int& shA, int& shW) -> bool {
    if ((ntn & 3) == 0 && (ntm & 63) == 0 && cpx == 32) {
      const int nsl = ntn >> 2, ng = ntm >> 6;
      if (j >= nsl * ng) return false;
      const int sl = j % nsl, gi = j / nsl;
      const int gm0 = (gi * 8 + xcd) * 8;
      m0 = (gm0 + (idx & 7)) * 256;
      n0 = (sl * 4 + (idx >> 3)) * 256;
      shA = (idx >> 3) & 3;
      shW = idx & 7;
      return true;
    }
    const int logical = (j * 8 + xcd) * cpx + idx;
    if (logical >= ntiles) return false;
    const int per_group = 8 * ntn;
    const int grp = logical / per_group, within = logical - grp * per_group;
    const int gm0 = grp * 8;
    const int gsz = (ntm - gm0) < 8 ? (ntm - gm0) : 8;
    m0 = (gm0 + within % gsz) * 256;
    n0 = (within / gsz) * 256;
    shA = (within / gsz) & 3;
    shW = (within % gsz) & 7;
    return true;
  };
  int m0, n0, shA0, shW0;
  if (!tile_of(0, m0, n0, shA0, shW0)) return;  // before any barrier
  // ---- de-phase the XCDs.  Every workgroup runs the same tile lengths, so all 256 CUs reach their epilogues in the same few hundred
  // cycles and their output bursts meet at the fabric: 256 CUs storing at once get 14.5 B/clk each, 128 get 28.7, 64 or fewer the
  // CU's own 33 B/clk (tools/store_probe, profiles/r06e_store_probe_grid.log) -- the 128 KiB of a tile cost ~9 k cycles in step and
  // ~4 k alone.  XCD x starts `stagger` * x cycles late (the workgroups that share an L2 stay in step); the price is 7 * stagger cycles
  // at the end of the launch.
  if (stagger > 0 && xcd > 0) {
    const long long t0_ = (long long)__builtin_readcyclecounter();
    const long long wait_ = (long long)stagger * xcd;
    while ((long long)__builtin_readcyclecounter() - t0_ < wait_) __builtin_amdgcn_s_sleep(8);
  }

  // ---- staging: wave w fills rows [64 w, 64 w + 64) of both operands, 8 rows (1 KiB) per instruction; source chunk = LDS chunk
  // position ^ ((row >> 1) & 7) (the swizzle of gemm256sp.hip).  One per-lane byte offset per piece, shared by both operands.
  unsigned soff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = w * 64 + (lane >> 3) + 8 * j;
    soff[j] = (unsigned)((row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
  }
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
  const unsigned dmw = lds_base + w * 8192;  // this wave's first piece inside an operand tile
#define W4_DMA(off, base, cimm)                                                                             \
  if (!AB_NODMA) asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dmw), "n"(cimm) \
               : "memory", "scc")
// piece d of a stage: d & 1 = operand (0: M, 1: N), d >> 1 = which 8 rows
#define W4_PIECE(pM, pN, buf, d)                                                          \
  if (((d) & 1) == 0) { W4_DMA(soff[(d) >> 1], pM, (buf) * W4_OPB + ((d) >> 1) * 1024); } \
  else { W4_DMA(soff[(d) >> 1], pN, (buf) * W4_OPB + W4_NBASE + ((d) >> 1) * 1024); }

  // ---- L2 prefetch: each K-tile every wave touches 24 lines of the K-tile PFD ahead (64 A rows + 32 W rows per CU: the CUs that
  // share a panel split it) with one dword LDS-DMA into a sink in its scratch; never waited for
  auto pf_addr = [&](int tm0, int tn0, int sA, int sW) -> const char* {
    const int Lr = w * 24 + (lane % 24);
    return Lr < 64 ? reinterpret_cast<const char*>(A) + (size_t)(tm0 + 64 * sA + Lr) * K * 2
                   : reinterpret_cast<const char*>(W) + (size_t)(tn0 + 32 * sW + (Lr - 64)) * K * 2;
  };
  const unsigned scr_m0 = lds_base + W4_SCR + w * W4_SCRW;
  const unsigned pf_m0 = scr_m0 + 7936;
#define W4_PF(ptr) \
  if (!AB_NOPF) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(ptr), "s"(pf_m0) : "memory");

  // ---- fragment read addresses: row l15 of a 16-row block (+ 2 KiB per block), k-chunk 4 sl + q4 of the 32-deep slab sl
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned fM[2], fN[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int xk = ((4 * sl + q4) ^ ((l15 >> 1) & 7)) << 4;
    fM[sl] = lds0 + (wr * 128 + l15) * 128 + xk;              // + buf * W4_OPB + blk * 2048
    fN[sl] = lds0 + W4_NBASE + (wc * 128 + l15) * 128 + xk;
  }
  float ssq[8], ssum[8];  // STATS: running sum of x^2 / of x of this lane's 8-value k-chunks of row l15 of activation block mi
  const unsigned ones2 = __builtin_amdgcn_readfirstlane(0x3c003c00u);  // (1.0h, 1.0h)
  f32x4 acc[8][8];  // [activation block mi][weight block ni]: a 16 x 16 MFMA block each
  w4_i32x4 Nd[2][8], Md[8];  // Md: a ring -- the fragment of unit U sits in slot U & 7 (requested at unit U - 6, when unit U - 8 is long done)
#define W4_DSREAD(dst, addr, off) \
  if (AB_NOREAD) asm volatile("" : "+v"(dst)); else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define W4_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");
// the fragment requests at the top of unit u of the K-tile in buffer `buf`
#define W4_READS(u, buf)                                                                          \
  if ((u) >= 2 && (u) <= 5) {                                                                     \
    W4_DSREAD(Nd[1][2 * ((u) - 2)], fN[1], (buf) * W4_OPB + (2 * ((u) - 2)) * 2048);              \
    W4_DSREAD(Nd[1][2 * ((u) - 2) + 1], fN[1], (buf) * W4_OPB + (2 * ((u) - 2) + 1) * 2048);      \
  }                                                                                               \
  if ((u) >= 10 && (u) <= 13) {                                                                   \
    W4_DSREAD(Nd[0][2 * ((u) - 10)], fN[0], ((buf) ^ 1) * W4_OPB + (2 * ((u) - 10)) * 2048);      \
    W4_DSREAD(Nd[0][2 * ((u) - 10) + 1], fN[0], ((buf) ^ 1) * W4_OPB + (2 * ((u) - 10) + 1) * 2048); \
  }                                                                                               \
  if ((u) < 2) { W4_DSREAD(Md[(u) + 6], fM[0], (buf) * W4_OPB + ((u) + 6) * 2048); }              \
  else if ((u) < 8) { W4_DSREAD(Md[((u) + 6) & 7], fM[1], (buf) * W4_OPB + ((u) - 2) * 2048); }   \
  else if ((u) == 8) {                                                                            \
    W4_DSREAD(Md[6], fM[1], (buf) * W4_OPB + 6 * 2048);                                           \
    W4_DSREAD(Md[7], fM[1], (buf) * W4_OPB + 7 * 2048);                                           \
  } else if ((u) >= 10) { W4_DSREAD(Md[(u) - 10], fM[0], ((buf) ^ 1) * W4_OPB + ((u) - 10) * 2048); }
// outstanding requests allowed when unit u multiplies (tools/…: the schedule's issue order); units 10..15 follow the drain of unit 9
#define W4_WAITS(u)                                                         \
  if ((u) == 0) { W4_WAIT_LGKM(4) }                                         \
  else if ((u) <= 3) { W4_WAIT_LGKM(10) }                                   \
  else if ((u) == 4) { W4_WAIT_LGKM(12) }                                   \
  else if ((u) <= 7) { W4_WAIT_LGKM(14) }                                   \
  else if ((u) == 8) { W4_WAIT_LGKM(5) }                                    \
  else if ((u) == 9) { W4_WAIT_LGKM(10) }
// The MFMAs are inline asm with the accumulator constrained to the AGPR file ("a"): with the builtin, hipcc's register allocator kept a
// handful of the 64 accumulator tuples in arch VGPRs and copied them in and out of a transient AGPR tuple around every MFMA
// (16 v_accvgpr_write + 16 v_accvgpr_read per unit in the steady state: -save-temps of the first build).  The epilogue opens with
// the wait states an MFMA result needs before a VALU read (hipcc pads nothing around inline asm).
#define W4_MFMA_ASM(accv, a_, b_, ZERO)                                                                                          \
  if (ZERO) {                                                                                                                    \
    if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(accv) : "v"(a_), "v"(b_));                              \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(accv) : "v"(a_), "v"(b_));                                 \
  } else {                                                                                                                       \
    if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(accv) : "v"(a_), "v"(b_));                             \
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accv) : "v"(a_), "v"(b_));                                \
  }
// STATS: statistic step j = 0..7 of unit u, issued behind the unit's j-th MFMA (steps 0..3: x . x of dword j of the unit's activation
// fragment, 4..7: x . 1 of dword j - 4); ZERO (the tile's first K-tile, slab 0): steps 0 / 4 start their sum from zero
#define W4_STAT_OP(u, j, ZERO)                                                                                                   \
  if (STATS) {                                                                                                                   \
    if ((j) == 0 && (ZERO)) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0" : "=v"(ssq[(u) & 7]) : "v"(Md[(u) & 7][0]));             \
    else if ((j) < 4) { CLIPX_DOT2_SQ(ssq[(u) & 7], Md[(u) & 7][(j) & 3]); }                                                     \
    else if ((j) == 4 && (ZERO)) asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(ssum[(u) & 7]) : "v"(Md[(u) & 7][0]), "s"(ones2)); \
    else { CLIPX_DOT2_SUM(ssum[(u) & 7], Md[(u) & 7][(j) & 3], ones2); }                                                         \
  }
#define W4_MFMA4(u, n0_, ZERO)                                                                                           \
  _Pragma("unroll") for (int ni = (n0_); ni < (n0_) + 4; ++ni) {                                                         \
    W4_MFMA_ASM(acc[(u) & 7][ni], Nd[(u) >> 3][ni], Md[(u) & 7], ZERO)                                                   \
    W4_STAT_OP(u, ni, ZERO)                                                                                              \
  }
// one unit: hook (a trickled store of the previous tile / an early load for this tile's epilogue), requests, wait, 4 MFMAs, X (a DMA
// piece or nothing), 4 MFMAs, Y
#define W4_UNIT(u, buf, ZERO, RD, X, Y, HK)  \
  HK(u)                              \
  W4_FENCE();                        \
  if (RD) { W4_READS(u, buf) }       \
  W4_WAITS(u)                        \
  W4_FENCE();                        \
  W4_MFMA4(u, 0, ZERO)               \
  W4_FENCE();                        \
  X                                  \
  W4_FENCE();                        \
  W4_MFMA4(u, 4, ZERO)               \
  W4_FENCE();                        \
  Y                                  \
  W4_FENCE();
#define W4_SYNC(vm)                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vm) : "memory");  \
  W4_FENCE();                                                \
  if (!AB_NOBAR) __builtin_amdgcn_s_barrier();               \
  W4_FENCE();
#define W4_NOP
#define W4_HK_NONE(u)
// one K-tile in buffer `buf`: units 0..9, the sync, units 10..15 with the stage of K-tile + 2 (pieces pM / pN into `buf`) when STG.
// ZERO: the tile's first K-tile -- slab 0's MFMAs start from zero instead of the accumulators (no 256 v_accvgpr_write per tile).
// PRE: code at unit 0 (bias fetch, prefetch)
#define W4_KTILE(buf, ZERO, SYNC, STG, pM, pN, PRE, RDN, HK)                                                 \
  W4_UNIT(0, buf, ZERO, true, PRE, W4_NOP, HK)  \
  W4_UNIT(1, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(2, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(3, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(4, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(5, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(6, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(7, buf, ZERO, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(8, buf, false, true, W4_NOP, W4_NOP, HK)  \
  W4_UNIT(9, buf, false, true, W4_NOP, W4_NOP, HK)  \
  SYNC                                                                                              \
  W4_UNIT(10, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 0) }, if (STG) { W4_PIECE(pM, pN, buf, 1) }, HK)  \
  W4_UNIT(11, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 2) }, if (STG) { W4_PIECE(pM, pN, buf, 3) }, HK)  \
  W4_UNIT(12, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 4) W4_FENCE(); W4_PIECE(pM, pN, buf, 5) }, if (STG) { W4_PIECE(pM, pN, buf, 6) }, HK)  \
  W4_UNIT(13, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 7) W4_FENCE(); W4_PIECE(pM, pN, buf, 8) }, if (STG) { W4_PIECE(pM, pN, buf, 9) }, HK)  \
  W4_UNIT(14, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 10) W4_FENCE(); W4_PIECE(pM, pN, buf, 11) }, if (STG) { W4_PIECE(pM, pN, buf, 12) }, HK)  \
  W4_UNIT(15, buf, false, RDN, if (STG) { W4_PIECE(pM, pN, buf, 13) W4_FENCE(); W4_PIECE(pM, pN, buf, 14) }, if (STG) { W4_PIECE(pM, pN, buf, 15) }, HK)

  const int nk = K >> 6;  // K-tiles per output tile (even, >= 4)
  const char* curM = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* curN = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;

  // ---- the epilogue's global traffic is taken off the tile boundary.  A CU stores ~21 B/clk when all 256 reach their epilogues
  // together (tools/store_probe): the 128 KiB of a 16-bit tile are ~6 k cycles of drain whatever the instruction count (the first
  // two epilogues of this kernel measured 5.7 k and 6.4 k for the plain form, 12 - 14 k for the in-place residual: 128 KiB in +
  // 128 KiB out; profiles/r06b_*, r06c_*).  So:
  //   * half of a tile's output lines (every second pass: 16 stores of 16 B per lane) stay in 64 VGPRs (`held`) and leave one per
  //     unit during the NEXT tile's first K-tile; the other half is stored from the passes themselves, under their arithmetic;
  //   * fp16 residual: the old rows of the held passes are requested during the tile's second-to-last K-tile, one load per unit,
  //     into `held` itself; those of the first two stored passes right behind them.  (vmcnt retires in order: a load issued
  //     behind a store waits for it, so the loads the epilogue waits for must be few and early.)
  // Store / load address of (pass k, row group i) inside the wave's 128 x 128 panel: rows 16 (k >> 1) + 8 i + sr, column half k & 1.
  // store layout: row sr + 8 i of a 16-row pass, columns 8 sj .. + 8 of its 64 (sr = lane >> 3, sj = lane & 7)
  // (16-bit-output epilogues read their passes back as 4 rows x 256 B per instruction: row lane >> 4 (+ 4 i), 16-B chunk lane & 15)
  const unsigned voff = OUT16 ? (unsigned)(((lane >> 4) * N + (lane & 15) * 8) * 2) : (unsigned)(((lane >> 3) * N + (lane & 7) * 8) * 2);
#define W4_POFF(k_, i_) ((size_t)(((k_) >> 1) * 16 + 8 * (i_)) * N * 2 + ((k_) & 1) * 128)
  // HS = 2: every second pass is held (16 slots: 64 VGPRs), HS = 4: three of four (24 slots: 96 VGPRs -- hipcc then spills 60 - 130
  // registers around the epilogue: not used).  Held slot s <-> (pass k with k % HS != 0, row group i).
  // STATS: 12 slots (the odd passes 1, 3, 5; pass 7 is stored from the epilogue) -- the 16 running sums need the registers.
  constexpr int HS = 2, NH = STATS ? 12 : (HS == 2 ? 16 : 24), NE = 16 / HS;
#define W4_HIDX(k_, i_) (HS == 2 ? ((k_) >> 1) * 2 + (i_) : (((k_) >> 2) * 3 + ((k_) & 3) - 1) * 2 + (i_))
#define W4_HELD_K(s_) (HS == 2 ? 2 * ((s_) >> 1) + 1 : 4 * (((s_) >> 1) / 3) + (((s_) >> 1) % 3) + 1)
#define W4_CLAMP(s_) ((s_) < NH ? ((s_) < 0 ? 0 : (s_)) : NH - 1)
  w4_u32x4 held[NH];
  w4_u32x4 ext[3][2];  // fp16 residual: old rows of the stored passes (pass HS e in slot e % 3), requested two stored passes ahead
  const char* yb_prev = reinterpret_cast<const char*>(outp);  // the previous tile's panel (trickled stores)
  const char* yb = yb_prev;
#define W4_STORE_ASM(data, base) \
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(data), "s"(base) : "memory")  /* s_nop 4: SGPR fresh from v_readlane; s_nop 1: store-data hazard */
// 16-bit-output epilogues: 8 passes of 16 rows x 128 columns, the odd ones held: slot s <-> (pass 2 (s >> 2) + 1, row group s & 3)
#define W4_POFF16(p_, i_) ((size_t)((p_) * 16 + 4 * (i_)) * N * 2)
#define W4_HOFF(s_) (OUT16 ? W4_POFF16(2 * ((s_) >> 2) + 1, (s_) & 3) : W4_POFF(W4_HELD_K(s_), (s_) & 1))
#define W4_TRICKLE(s_) { const char* b_ = yb_prev + W4_HOFF(s_); W4_STORE_ASM(held[W4_CLAMP(s_)], b_); }
// one store per second unit over the first pair (HS = 2: 16 stores in 32 units)
#define W4_HK_TRICKLE0(u) if (((u) & 1) && ((u) >> 1) < NH) { W4_TRICKLE((u) >> 1) }
#define W4_HK_TRICKLE1(u) if (((u) & 1) && 8 + ((u) >> 1) < NH) { W4_TRICKLE(8 + ((u) >> 1)) }
#define W4_LD_HELD(s_) held[W4_CLAMP(s_)] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (int)W4_POFF(W4_HELD_K(s_), (s_) & 1), 0);
#define W4_LD_EXT(e_, i_) ext[(e_) % 3][i_] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (int)W4_POFF(HS * (e_), i_), 0);
#define W4_HK_XHELD0(u) if (!OUT16) { W4_LD_HELD(u) }
#define W4_HK_XHELD1(u) if (!OUT16 && 16 + (u) < NH) { W4_LD_HELD(16 + (u)) } else if (!OUT16 && 16 + (u) < NH + 4) { W4_LD_EXT((16 + (u) - NH) >> 1, (16 + (u) - NH) & 1) }

  // VMEM bookkeeping (vmcnt retires in order): at a K-tile's sync "the next K-tile has landed" = at most the operations issued since
  // its last piece are outstanding: the prefetch of unit 0 (1) in steady state; none in the last pair, except the residual's early
  // loads of units 0..9 (10) -- and its second sync (vmcnt 0) also lands the bias and every early load issued before it; after an
  // epilogue its 2 NE stores + the prefetch + the trickled stores of units 1, 3, .. 9 (the residual's in-pass loads were consumed, i.e.
  // have retired); at the second K-tile's sync its prefetch + its trickled stores of units 1, 3, .. 9.
  // (exact counts: a wait that allows MORE outstanding operations than were issued behind the pieces would let a piece be in flight)
  constexpr int EPI_STORES = 2 * NE + (16 - NH < 0 ? 0 : 16 - NH);       // stores issued by an epilogue (16-bit outputs: 16 + the un-held odd rows)
  constexpr int TR0 = NH < 5 ? NH : 5, TR1 = NH - 8 < 0 ? 0 : (NH - 8 < 5 ? NH - 8 : 5);  // trickled stores of units 1, 3, .. 9 of the first / second K-tile
  constexpr int EPI_VM = EPI_STORES + 1 + TR0, EPI_VM1 = 1 + TR1;

  // ---- prologue: K-tiles 0, 1 of the first tile; K-tile 0 landed; the requests units 10..15 of a previous K-tile would have made
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if ((d & 1) == 0) { W4_DMA(soff[d >> 1], curM, (d >> 1) * 1024); } else { W4_DMA(soff[d >> 1], curN, W4_NBASE + (d >> 1) * 1024); }
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if ((d & 1) == 0) { W4_DMA(soff[d >> 1], curM + 128, W4_OPB + (d >> 1) * 1024); } else { W4_DMA(soff[d >> 1], curN + 128, W4_OPB + W4_NBASE + (d >> 1) * 1024); }
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  W4_FENCE();
  __builtin_amdgcn_s_barrier();
  W4_FENCE();
  // (buf = 1 so that "(buf ^ 1)" is buffer 0: the same statements as units 10..15 of a K-tile in buffer 1)
  W4_READS(10, 1) W4_READS(11, 1) W4_READS(12, 1) W4_READS(13, 1) W4_READS(14, 1) W4_READS(15, 1)
  W4_FENCE();

  // bias (+ row scales) of the current tile: ONE LDS-DMA per wave at the tile's last K-tile drops the wave's 128 bias floats (lanes
  // 0..31) and the 128 row scales of its rows (lanes 32..63) into its scratch; inline asm: hipcc must not know about it
  const unsigned bias_m0 = scr_m0 + 4096;
  auto load_bias = [&]() {
    const float* p = (OUT16 && !STATS && lane >= 32) ? rowscale + m0 + wr * 128 + (lane - 32) * 4 : bias + n0 + wc * 128 + (lane & 31) * 4;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(bias_m0) : "memory");
  };

  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tstamp = DBG_TIMER ? (long long)__builtin_readcyclecounter() : 0;
  const long long tstart = tstamp;
#define W4_STAMP(slot)                                                 \
  if (DBG_TIMER) {                                                     \
    const long long now_ = (long long)__builtin_readcyclecounter();    \
    ph[slot] += now_ - tstamp;                                         \
    tstamp = now_;                                                     \
  }

  const char* sM = curM + 256;  // the next K-tile of this tile to stage (K-tile 2)
  const char* sN = curN + 256;
  const bool pf_run = nk >= 8 && (size_t)ntm * 256 * K * 2 >= ((size_t)64 << 20);
  const int pf_sw = pf_run ? nk - PFD - 1 : -1;
  const int pf_step = pf_run ? 128 : 0;
  const char* pfp = pf_addr(m0, n0, shA0, shW0) + (pf_run ? PFD * 128 : 0);
// (the next tile's prefetch address is computed at the switch, not held across the tile: two VGPRs)
#define W4_PF_NEXT(t) if ((t) == pf_sw) { pfp = have_next ? pf_addr(nm0, nn0, nshA, nshW) : pf_addr(m0, n0, 0, 0); } else { pfp += pf_step; }
  bool first = false;  // the K-tile about to run is the first one after an epilogue
  for (int j = 0;; ++j) {
    int nm0 = 0, nn0 = 0, nshA = 0, nshW = 0;
    const bool have_next = tile_of(j + 1, nm0, nn0, nshA, nshW);
    const char* nxtM = reinterpret_cast<const char*>(A) + (size_t)nm0 * K * 2;
    const char* nxtN = reinterpret_cast<const char*>(W) + (size_t)nn0 * K * 2;

    // ---- first pair (K-tiles 0, 1): stage K-tiles 2, 3; after an epilogue the held half of the previous tile leaves, a store per unit
    if (first) {
      W4_KTILE(0, true, W4_SYNC(EPI_VM), true, sM, sN, W4_PF(pfp), true, W4_HK_TRICKLE0)
      W4_PF_NEXT(0)
      W4_STAMP(1)
      W4_KTILE(1, false, W4_SYNC(EPI_VM1), true, sM + 128, sN + 128, W4_PF(pfp), true, W4_HK_TRICKLE1)
    } else {
      W4_KTILE(0, true, W4_SYNC(1), true, sM, sN, W4_PF(pfp), true, W4_HK_NONE)
      W4_PF_NEXT(0)
      W4_STAMP(1)
      W4_KTILE(1, false, W4_SYNC(1), true, sM + 128, sN + 128, W4_PF(pfp), true, W4_HK_NONE)
    }
    W4_PF_NEXT(1)
    sM += 256;
    sN += 256;
    if (DBG_TIMER) { W4_STAMP(3) ph[5] += 1; }
    // ---- steady state (K-tiles 2 .. nk-3): branch-free
    for (int t = 2; t < nk - 2; t += 2) {
      W4_KTILE(0, false, W4_SYNC(1), true, sM, sN, W4_PF(pfp), true, W4_HK_NONE)
      W4_PF_NEXT(t)
      W4_KTILE(1, false, W4_SYNC(1), true, sM + 128, sN + 128, W4_PF(pfp), true, W4_HK_NONE)
      W4_PF_NEXT(t + 1)
      sM += 256;
      sN += 256;
      if (DBG_TIMER) { W4_STAMP(3) ph[5] += 2; }
    }
    // ---- last pair (K-tiles nk-2, nk-1): stage the next tile's K-tiles 0, 1; fetch this tile's bias and, for the residual, the old
    // rows the epilogue starts with.  No prefetch: its sink is in the scratch the epilogue is about to use
    yb = reinterpret_cast<const char*>(outp) + ((size_t)(m0 + wr * 128) * N + n0 + wc * 128) * 2;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(yb), 0, 0x7ffffffe, 0x00020000);
    if (OUT16) {
      W4_KTILE(0, false, W4_SYNC(0), have_next, nxtM, nxtN, W4_NOP, true, W4_HK_NONE)
    } else {
      W4_KTILE(0, false, W4_SYNC(10), have_next, nxtM, nxtN, W4_NOP, true, W4_HK_XHELD0)
    }
    W4_KTILE(1, false, W4_SYNC(0), have_next, nxtM + 128, nxtN + 128, load_bias();, false, W4_HK_XHELD1)
    W4_PF_NEXT(nk - 2)
    W4_PF_NEXT(nk - 1)
    if (DBG_TIMER) { W4_STAMP(3) ph[5] += 2; }

    // ---- epilogue of this output tile (the next tile's K-tile 0 has landed, its K-tile 1 is in flight; the bias landed before the last
    // K-tile's sync).  Sixteen passes of 16 rows x 64 columns: the raw f32 accumulators go STRAIGHT from the AGPRs into the wave's LDS
    // scratch (ds_write_b128 takes AGPR data: no v_accvgpr_read, 256 of them per tile otherwise), come back in store layout -- a lane
    // owns 8 consecutive columns of one row -- and all the arithmetic (row scale, bias, activation / residual add, rounding) happens
    // there.  Two 4-KiB pass buffers: pass k + 1 is written while pass k is read back.  Per element the operations are those of
    // gemm_store_quad: bit-identical rows.
    W4_FENCE();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last MFMAs' results -> ds_write data (inline-asm MFMAs: nobody else pads)
    W4_FENCE();
    {
      // lane-constant LDS addresses of the epilogue are recomputed per tile from an opaque copy of the lane number: hoisted to the
      // kernel entry (loop-invariant code motion) they were spilled and reloaded inside the K-tile code -- a scratch reload waits
      // vmcnt(0) (the first builds of this epilogue: -save-temps)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int l15 = ln & 15, q4 = ln >> 4, sr = ln >> 3, sj = ln & 7;
      unsigned char* scr = smem + W4_SCR + w * W4_SCRW;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
      if (OUT16) {
        // ---- 16-bit outputs: arithmetic in ACCUMULATOR layout, 16-bit values through the LDS.  The LDS takes ~80 B/clk/CU of 8- / 16-byte
        // writes (MI355X_MICROARCH, LDS table): the f32 transposition of the residual path below moves 256 KiB per tile into it (3.3 k
        // cycles of the 4.9 k its first measurement took, profiles/r06f_*), 16-bit values half of that.  Eight passes of 16 rows x 128
        // columns: the block row's accumulators leave the AGPRs (v_accvgpr_read), scale + bias + activation + rounding, 8-B writes in
        // accumulator layout (chunk ^ row), 16-B reads in store layout (a lane: 8 consecutive columns; 16 lanes: 256 contiguous bytes
        // of a row), even passes stored from here, odd passes held.  Pass p + 1 is computed and written while pass p's reads fly.
        float4 b4[8];  // bias of the lane's columns 16 nb + 4 q4 .. + 4
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) b4[nb] = *reinterpret_cast<const float4*>(scr + 4096 + (nb * 16 + 4 * q4) * 4);
        float rr[8];
        if (STATS) {
          // the four k-parts of a row sit in lanes l15 + 16 q4: xor-16 / xor-32 butterfly, then the canonical closing formula
          const float inv_d = 1.f / (float)K;
          bool bad = false;
#pragma unroll
          for (int pq = 0; pq < 8; ++pq) {
            float s1 = ssum[pq], s2 = ssq[pq];
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            bad |= !(fabsf(s1) <= 3.0e38f);  // inf / NaN in the row (clipx.h: CLIPX_E_RANGE; rowstats_kernel has the same guard)
            rr[pq] = ln_rstd_onepass(s1, s2, inv_d, stats_eps);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the shuffles are LDS-queue operations: none may be in flight when the counted waits below start)
          if (bad) {  // error path: the batch is reported invalid
            if (range_flag) atomicOr(range_flag, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the atomic counts in vmcnt: nothing assumes a number of outstanding operations across it)
          }
          W4_FENCE();
        } else {
#pragma unroll
          for (int pq = 0; pq < 8; ++pq) rr[pq] = *reinterpret_cast<const float*>(scr + 4096 + 512 + (pq * 16 + l15) * 4);
        }
        unsigned qpos[8], rp16[4];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) qpos[nb] = lds0 + W4_SCR + w * W4_SCRW + l15 * 256 + (((2 * nb + (q4 >> 1)) ^ l15) << 4) + (q4 & 1) * 8;
        const int lr = ln >> 4, lc = ln & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) rp16[i] = lds0 + W4_SCR + w * W4_SCRW + (lr + 4 * i) * 256 + ((lc ^ (lr + 4 * i)) << 4);
        w4_u32x4 qd[2][4];
#define W4_ACC_HERE(p_) _Pragma("unroll") for (int nb_ = 0; nb_ < 8; ++nb_) asm volatile("" : "+a"(acc[p_][nb_]));
#define W4_CONV_PASS(p_)                                                                                                              \
  W4_ACC_HERE(p_)                                                                                                                     \
  _Pragma("unroll") for (int nb = 0; nb < 8; ++nb) {                                                                                  \
    const float4 bq = b4[nb];                                                                                                         \
    float v[4];                                                                                                                       \
    if (EPI == EPI_BIAS_QGELU_BF16) {                                                                                                 \
      const f32x2_t r2 = {rr[p_], rr[p_]}, kk = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};                       \
      const f32x2_t one = {1.f, 1.f};                                                                                                 \
      _Pragma("unroll") for (int e = 0; e < 4; e += 2) {                                                                              \
        const f32x2_t a2 = {acc[p_][nb][e], acc[p_][nb][e + 1]};                                                                      \
        const f32x2_t b2 = {e == 0 ? bq.x : bq.z, e == 0 ? bq.y : bq.w};                                                              \
        const f32x2_t x2 = __builtin_elementwise_fma(a2, r2, b2);                                                                     \
        const f32x2_t t2 = x2 * kk;                                                                                                   \
        const f32x2_t d2 = (f32x2_t){__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])} + one;                             \
        const f32x2_t y2 = x2 * (f32x2_t){__builtin_amdgcn_rcpf(d2[0]), __builtin_amdgcn_rcpf(d2[1])};                                \
        v[e] = y2[0];                                                                                                                 \
        v[e + 1] = y2[1];                                                                                                             \
      }                                                                                                                               \
    } else {                                                                                                                          \
      v[0] = __builtin_fmaf(acc[p_][nb][0], rr[p_], bq.x);                                                                            \
      v[1] = __builtin_fmaf(acc[p_][nb][1], rr[p_], bq.y);                                                                            \
      v[2] = __builtin_fmaf(acc[p_][nb][2], rr[p_], bq.z);                                                                            \
      v[3] = __builtin_fmaf(acc[p_][nb][3], rr[p_], bq.w);                                                                            \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) if (EPI == EPI_BIAS_GELU_BF16) v[e] = gelu_erf(v[e]);                             \
    }                                                                                                                                 \
    uint2 o;                                                                                                                          \
    if (EPI == EPI_BIAS_F16) {                                                                                                        \
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); /* never v_fma_mix*_f16 (see gemm_store_quad) */             \
      o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, f16x2_t));                                    \
      o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, f16x2_t));                                    \
    } else {                                                                                                                          \
      o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));                                   \
      o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));                                   \
    }                                                                                                                                 \
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(qpos[nb]), "v"(o), "n"(((p_) & 1) * 4096) : "memory");                         \
  }
#define W4_READ16(p_) \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qd[(p_) & 1][i_]) : "v"(rp16[i_]), "n"(((p_) & 1) * 4096));
        W4_CONV_PASS(0)
        W4_READ16(0)
        W4_FENCE();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          if (p + 1 < 8) { W4_CONV_PASS(p + 1) }
          // pass p's reads have landed when at most the 8 writes issued since are outstanding
          if (p + 1 < 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          W4_FENCE();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if ((p & 1) && (p >> 1) * 4 + i < NH) {
              held[W4_CLAMP((p >> 1) * 4 + i)] = qd[p & 1][i];
              asm volatile("" : "+v"(held[W4_CLAMP((p >> 1) * 4 + i)]));  // here, not sunk to its store in the next tile's K-loop
            } else {
              const char* base = yb + W4_POFF16(p, i);
              W4_STORE_ASM(qd[p & 1][i], base);
            }
          }
          if (p + 1 < 8) { W4_READ16(p + 1) }
          W4_FENCE();
        }
      } else {
      float4 bA[2][2];  // bias of the lane's 8 columns, per column half h
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) bA[h][e] = *reinterpret_cast<const float4*>(scr + 4096 + (h * 64 + sj * 8 + e * 4) * 4);
      float rs[8][2];  // row scales of the lane's rows 16 pr + sr + 8 i
      if (OUT16) {
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
#pragma unroll
          for (int i = 0; i < 2; ++i) rs[pr][i] = *reinterpret_cast<const float*>(scr + 4096 + 512 + (pr * 16 + sr + 8 * i) * 4);
      }
      // accumulator layout -> scratch: quad nbl (columns 16 nbl + 4 q4 .. + 4 of the pass = 16-B chunk 4 nbl + q4 of row l15), chunk
      // position ^ row; store layout <- scratch: chunks 2 sj, 2 sj + 1 of row sr + 8 i
      unsigned wpos[4], rpos[2][2];
#pragma unroll
      for (int nbl = 0; nbl < 4; ++nbl) wpos[nbl] = lds0 + W4_SCR + w * W4_SCRW + l15 * 256 + (((4 * nbl + q4) ^ l15) << 4);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) rpos[i][e] = lds0 + W4_SCR + w * W4_SCRW + (sr + 8 * i) * 256 + (((2 * sj + e) ^ (sr + 8 * i)) << 4);
#define W4_WRITE_PASS(k_)                                                                                              \
  _Pragma("unroll") for (int nbl = 0; nbl < 4; ++nbl)                                                                  \
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wpos[nbl]), "a"(acc[(k_) >> 1][4 * ((k_) & 1) + nbl]), "n"(((k_) & 1) * 4096) : "memory");
      // software pipeline over the passes (one wave per SIMD: nothing else hides an LDS round trip): pass k + 2 is written and pass
      // k + 1 read back while pass k's arithmetic runs.  Reads are inline asm with counted waits (hipcc's own waits would also sit out
      // the requests just issued for the next pass).
      f32x4 ab[2][2][2];  // [k & 1][row group i][chunk e]
#define W4_READ_PASS(k_)                                                                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_)                         \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ab[(k_) & 1][i_][e_]) : "v"(rpos[i_][e_]), "n"(((k_) & 1) * 4096));
      W4_WRITE_PASS(0)
      W4_WRITE_PASS(1)
      W4_READ_PASS(0)
      W4_FENCE();
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int pr = k >> 1, h = k & 1;
        if (k + 1 < 16) { W4_READ_PASS(k + 1) }
        if (k + 2 < 16) { W4_WRITE_PASS(k + 2) }
        // pass k has landed when at most the requests issued since are outstanding (4 reads + 4 writes)
        if (k + 2 < 16) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if (k + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        f32x4 a[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) a[i][e] = ab[k & 1][i][e];
        const bool stored = (k % HS) == 0;  // these passes leave from here; the others are held
        if (!OUT16 && stored && k / HS + 2 < NE) { W4_LD_EXT(k / HS + 2, 0) W4_LD_EXT(k / HS + 2, 1) }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float4 bq = bA[h][e];
            const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
            if (EPI == EPI_BIAS_QGELU_BF16) {
              // packed f32 arithmetic for everything but the two transcendentals: each packed lane is the IEEE operation of
              // gemm_common.h's quick_gelu, so the bits do not change
              const f32x2_t r2 = {rs[pr][i], rs[pr][i]}, kk = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};
              const f32x2_t one = {1.f, 1.f};
#pragma unroll
              for (int c = 0; c < 4; c += 2) {
                const f32x2_t a2 = {a[i][e][c], a[i][e][c + 1]};
                const f32x2_t b2 = {bb[c], bb[c + 1]};
                const f32x2_t x2 = __builtin_elementwise_fma(a2, r2, b2);
                const f32x2_t t2 = x2 * kk;
                const f32x2_t d2 = (f32x2_t){__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])} + one;
                const f32x2_t y2 = x2 * (f32x2_t){__builtin_amdgcn_rcpf(d2[0]), __builtin_amdgcn_rcpf(d2[1])};
                v[4 * e + c] = y2[0];
                v[4 * e + c + 1] = y2[1];
              }
            } else if (OUT16) {
              // one fma per element, like gemm_store_quad (gemm_common.h): bit-identical rows from every kernel
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                v[4 * e + c] = __builtin_fmaf(a[i][e][c], rs[pr][i], bb[c]);
                if (EPI == EPI_BIAS_GELU_BF16) v[4 * e + c] = gelu_erf(v[4 * e + c]);
              }
            } else {
              // fp16 in-place residual x16 = fp16(f32(x16) + (acc + bias)): the association of gemm_store_quad
              const w4_u32x4 xo = stored ? ext[(k / HS) % 3][i] : held[W4_CLAMP(W4_HIDX(k, i))];
              const f16x4 xh = __builtin_bit_cast(f16x4, (uint2){xo[2 * e], xo[2 * e + 1]});
#pragma unroll
              for (int c = 0; c < 4; ++c) v[4 * e + c] = (float)xh[c] + (a[i][e][c] + bb[c]);
            }
          }
          w4_u32x4 o;
          if (EPI == EPI_BIAS_F16 || !OUT16) {
            // f32 result first, THEN the rounding to fp16 -- never v_fma_mix*_f16 (see gemm_store_quad)
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * c], v[2 * c + 1]}, f16x2_t));
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * c], v[2 * c + 1]}, bf16x2_t));
          }
          if (!stored) {
            held[W4_CLAMP(W4_HIDX(k, i))] = o;
            asm volatile("" : "+v"(held[W4_CLAMP(W4_HIDX(k, i))]));  // here, not sunk to its store in the next tile's K-loop
          } else {
            const char* base = yb + W4_POFF(k, i);
            W4_STORE_ASM(o, base);
          }
        }
        W4_FENCE();
      }
    }  // (fp16 residual path)
    }
    if (DBG_TIMER) { W4_STAMP(0) ph[6] += 1; }
    yb_prev = yb;
    if (!have_next) break;
    // the next tile's first fragment requests (what units 10..15 of a K-tile ask for; the last K-tile of a tile leaves them out: 56
    // fragment registers live across the epilogue made hipcc spill them -- an asm ds_read's destination stored before it has landed)
    W4_READS(10, 1) W4_READS(11, 1) W4_READS(12, 1) W4_READS(13, 1) W4_READS(14, 1) W4_READS(15, 1)
    W4_FENCE();
    first = true;
    m0 = nm0;
    n0 = nn0;
    sM = nxtM + 256;
    sN = nxtN + 256;
  }
  // the last tile's held half
#pragma unroll
  for (int u = 0; u < NH; ++u) W4_TRICKLE(u)
  W4_FENCE();
  // ---- the ragged 257th m-tile (gemm256_tail.h): every workgroup takes one 32-row strip of it
  if (tail_nb > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores count in vmcnt: start the tail's arithmetic at zero
    W4_FENCE();
    __builtin_amdgcn_s_barrier();                     // every wave is done with the K-tile buffers
    W4_FENCE();
    if (tail_nb == 1) gemm256_tail<EPI, F16, 1, 4>(A, W, bias, outp, nullptr, 1, N, K, rowscale, nullptr, tail_m0, smem, lds_base, w, lane);
    else if (tail_nb == 2) gemm256_tail<EPI, F16, 2, 4>(A, W, bias, outp, nullptr, 1, N, K, rowscale, nullptr, tail_m0, smem, lds_base, w, lane);
    else if (tail_nb == 3) gemm256_tail<EPI, F16, 3, 4>(A, W, bias, outp, nullptr, 1, N, K, rowscale, nullptr, tail_m0, smem, lds_base, w, lane);
    else gemm256_tail<EPI, F16, 4, 4>(A, W, bias, outp, nullptr, 1, N, K, rowscale, nullptr, tail_m0, smem, lds_base, w, lane);
  }
  if (DBG_TIMER && tid == 0 && blockIdx.x < 2048) {
    ph[7] = (long long)__builtin_readcyclecounter() - tstart;
#pragma unroll
    for (int i = 0; i < 8; ++i) g_w4_phase[blockIdx.x * 8 + i] = ph[i];
  }
}

#ifdef CLIPX_ABLATE
extern "C" int clipx_dbg_phase_cycles_w4(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_phase), (size_t)n * sizeof(long long));
}
#endif

// cycles by which XCD x + 1 starts behind XCD x (gemm256w4_kernel: "de-phase the XCDs"); CLIPX_GEMM_FLAGS overrides it in the tools build
static int w4_stagger() {
#ifdef CLIPX_ABLATE
  if (const char* fl = getenv("CLIPX_GEMM_FLAGS")) return atoi(fl);
#endif
  return 0;
}

template <int EPI, bool F16, int DBG = 0, bool STATS = false>
static hipError_t launch_w4_epi(const GemmArgs& g, int grid, hipStream_t st) {
  const size_t smem = W4_SCR + 4 * W4_SCRW;  // 160 KiB: the whole LDS of the CU
  auto kern = gemm256w4_kernel<EPI, F16, DBG, STATS>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, g.A, g.W, g.bias, g.out, g.N, g.K, g.M / 256, g.N / 256, g.rowscale, g.tail_m0, g.tail_nb, w4_stagger(),
                     g.stats_eps, g.range_flag);
  return hipGetLastError();
}

// true when the 4-wave kernel computes the LayerNorm row scales of this GEMM itself (GemmArgs.stats_eps > 0)
bool gemm256w4_fuses_stats(const GemmArgs& g) {
  return g.stats_eps > 0.f && g.f16 && gemm256w4_supports(g) &&
         (g.epi == EPI_BIAS_BF16 || g.epi == EPI_BIAS_F16 || g.epi == EPI_BIAS_QGELU_BF16 || g.epi == EPI_BIAS_GELU_BF16);
}

// true when the 4-wave kernel has this (epilogue, operand type, shape)
bool gemm256w4_supports(const GemmArgs& g) {
  if (g.M <= 0 || g.M % 256 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K < 256) return false;
  if (g.f16) return g.epi == EPI_BIAS_BF16 || g.epi == EPI_BIAS_F16 || g.epi == EPI_BIAS_QGELU_BF16 || g.epi == EPI_BIAS_GELU_BF16;
  return g.epi == EPI_BIAS_BF16 || g.epi == EPI_BIAS_F16 || g.epi == EPI_BIAS_QGELU_BF16 || g.epi == EPI_BIAS_GELU_BF16 || g.epi == EPI_BIAS_RESID_H16;
}

hipError_t launch_gemm256w4(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (!gemm256w4_supports(g)) return hipErrorInvalidValue;
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;
  if (grid < 8) grid = 8;
#ifdef CLIPX_ABLATE
  {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    if (dbg && atoi(dbg) == 16) {
      if (g.f16 && g.epi == EPI_BIAS_F16) return launch_w4_epi<EPI_BIAS_F16, true, 16>(g, grid, st);
      if (g.f16 && g.epi == EPI_BIAS_QGELU_BF16) return launch_w4_epi<EPI_BIAS_QGELU_BF16, true, 16>(g, grid, st);
      if (!g.f16 && g.epi == EPI_BIAS_BF16) return launch_w4_epi<EPI_BIAS_BF16, false, 16>(g, grid, st);
      if (!g.f16 && g.epi == EPI_BIAS_RESID_H16) return launch_w4_epi<EPI_BIAS_RESID_H16, false, 16>(g, grid, st);
    }
    if (dbg && atoi(dbg) >= 19 && atoi(dbg) <= 23 && g.f16 && g.epi == EPI_BIAS_F16) {  // K-loop ablations: the QKV form only
      switch (atoi(dbg)) {
        case 19: return launch_w4_epi<EPI_BIAS_F16, true, 19>(g, grid, st);
        case 20: return launch_w4_epi<EPI_BIAS_F16, true, 20>(g, grid, st);
        case 21: return launch_w4_epi<EPI_BIAS_F16, true, 21>(g, grid, st);
        case 22: return launch_w4_epi<EPI_BIAS_F16, true, 22>(g, grid, st);
        default: return launch_w4_epi<EPI_BIAS_F16, true, 23>(g, grid, st);
      }
    }
  }
#endif
  if (gemm256w4_fuses_stats(g)) {
    switch (g.epi) {
      case EPI_BIAS_BF16: return launch_w4_epi<EPI_BIAS_BF16, true, 0, true>(g, grid, st);
      case EPI_BIAS_F16: return launch_w4_epi<EPI_BIAS_F16, true, 0, true>(g, grid, st);
      case EPI_BIAS_QGELU_BF16: return launch_w4_epi<EPI_BIAS_QGELU_BF16, true, 0, true>(g, grid, st);
      case EPI_BIAS_GELU_BF16: return launch_w4_epi<EPI_BIAS_GELU_BF16, true, 0, true>(g, grid, st);
      default: return hipErrorInvalidValue;
    }
  }
  if (g.f16) {
    switch (g.epi) {
      case EPI_BIAS_BF16: return launch_w4_epi<EPI_BIAS_BF16, true>(g, grid, st);
      case EPI_BIAS_F16: return launch_w4_epi<EPI_BIAS_F16, true>(g, grid, st);
      case EPI_BIAS_QGELU_BF16: return launch_w4_epi<EPI_BIAS_QGELU_BF16, true>(g, grid, st);
      case EPI_BIAS_GELU_BF16: return launch_w4_epi<EPI_BIAS_GELU_BF16, true>(g, grid, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_w4_epi<EPI_BIAS_BF16, false>(g, grid, st);
    case EPI_BIAS_F16: return launch_w4_epi<EPI_BIAS_F16, false>(g, grid, st);
    case EPI_BIAS_QGELU_BF16: return launch_w4_epi<EPI_BIAS_QGELU_BF16, false>(g, grid, st);
    case EPI_BIAS_GELU_BF16: return launch_w4_epi<EPI_BIAS_GELU_BF16, false>(g, grid, st);
    case EPI_BIAS_RESID_H16: return launch_w4_epi<EPI_BIAS_RESID_H16, false>(g, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
