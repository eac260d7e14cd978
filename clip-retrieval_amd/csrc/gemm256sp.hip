// gemm256sp.hip -- persistent 256x256x64 software-pipelined bf16 GEMM for gfx950, 8 waves (GemmArgs.variant == 3; since round 6 the default,
// variant 6, sends the forms gemm256w4.hip has to that 4-wave kernel and everything else here).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A bf16 [M, K] activations, W bf16 [N, K] (torch Linear)
//
// The linear layers inside `model.encode_image/encode_text` (reference clip_retrieval/clip_inference/mapper.py:57,65).
//
//   * one 512-thread workgroup per CU, persistent over output tiles as one continuous K-tile stream.
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 blocks of v_mfma_f32_16x16x32_{bf16,f16} (128 accumulator VGPRs;
//     CLIPX_MFMA16 = 1 since round 4 -- the 32x32x16 form is kept behind CLIPX_MFMA16 = 0 for A/B only).
//   * LDS: 2 K-tile buffers x (M operand 256 rows + N operand 256 rows) x 128 B = 128 KiB, filled by LDS-DMA
//     (global_load_lds_dwordx4, 8 per wave per K-tile, SGPR base + one of four per-lane offsets, M0 = one s_add),
//     chunk-XOR swizzled through the source address; + 32 KiB of per-wave scratch for the epilogue transposition.
//   * per K-tile each wave runs 8 units of {2 or 6 ds_read_b128 for the NEXT unit, 8 MFMA of this unit} from two
//     register sets (32x32x16 form: 4 k-steps of 6 reads + 8 MFMA); the single s_barrier of the K-tile sits BEFORE the last
//     unit's MFMAs:
//         step 3:  lgkmcnt(0); vmcnt(1)  [K-tile g+1 landed; the L2 prefetch may stay in flight]; s_barrier;
//                  ds_read step 0 of K-tile g+1; 8 MFMA; stage K-tile g+2 into the buffer just released;
//                  one dword LDS-DMA that pulls 12 lines of K-tile g+4 into the L2 (see "L2 prefetch" below)
//     Everything between the barrier and those MFMAs is on the critical path of all 8 waves at once, so the steady
//     state is branch-free straight-line code: the first and last K-tile pair of a tile (epilogue bookkeeping, next
//     tile's operands, bias) are separate copies of the K-tile body.
//   * epilogue: accumulators (+bias, activation) are transposed through the wave's private LDS scratch so that
//     every global access is 16 B per lane with 8 consecutive lanes covering one full 128-B line of a row; the
//     stores are SGPR-base + one VGPR offset and are never waited for (a CU stores ~20 B/clk: the 128 KiB of a bf16
//     tile drain under the next tile's first K-tiles).  Measured with the phase timer (DBG 16): an earlier version whose
//     address registers spilled paid an `s_waitcnt vmcnt(0)` per store = 17.7 k cycles per tile; now ~4 k.
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0 (the launcher in clip_kernels.hip peels ragged rows).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"
#include "gemm256_tail.h"

namespace clipx {

// LDS map: M operand of K-tile buffer b at b*32 KiB, N operand at 64 KiB + b*32 KiB (each [256 rows][128 B]), so that
// every fragment read is <per-lane base of (operand, k-step)> + a 16-bit immediate.
constexpr int S_OPB = 32768;      // bytes of one operand tile
constexpr int S_NBASE = 65536;    // N operand tiles start here
constexpr int S_SCRATCH = 131072; // per-wave 4 KiB epilogue scratch starts here

#define S_FENCE() __builtin_amdgcn_sched_barrier(0)
// Store-data guard.  On gfx950 a buffer_store_dwordx4 reads its four data dwords over many cycles AFTER it has issued, and
// hipcc assumes no hazard at all for the register-soffset form these epilogues use (GCNHazardRecognizer: the ">64-bit store
// data" hazard is only modelled for stores WITHOUT a soffset register): when the register allocator reused a data register
// 5-7 instructions behind the store, the store wrote the NEW contents (found by tests/test_clip_gpu.py::
// test_gemm_layernorm_fold_hooks_both_kernels: z / w lanes of the f32 residual rows overwritten by the bf16 shadow
// exchange).  The guard keeps the data registers live (an asm input) across 8 wait states behind the store (s_nop 3, 7 and 15 all pass; what matters is that the registers stay live).
#ifndef CLIPX_STORE_GUARD_NOPS
#define CLIPX_STORE_GUARD_NOPS 7
#endif
#define S_STR2(x) #x
#define S_STR(x) S_STR2(x)
#define S_STORE_GUARD(v) asm volatile("s_nop " S_STR(CLIPX_STORE_GUARD_NOPS) ::"v"(v))

// DBG 16: per-phase shader-cycle totals of every block's wave 0 (8 counters per block): [0] epilogues, [1] / [2] first and
// second K-tile after an epilogue, [3] steady-state K-tiles, [4] last K-tile pair of a tile, [5] number of steady K-tiles,
// [6] tiles, [7] whole kernel.  Read back with clipx_dbg_phase_cycles().
__device__ long long g_sp_phase[2048 * 8];

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (the ragged 257th m-tile: gemm256_tail.h)

// DBG (ablation, EPI_BIAS_BF16 only; garbage results): 1 = no staging, 3 = no staging and no ds_reads,
// 5 = no epilogue at all, 6 = epilogue without its global stores, 16 = phase timer (correct results),
// 19 / 20 = phase timer + ablations 1 / 3, 21 = phase timer + L2-resident operands, 22 / 24 = no L2 prefetch (with / without timer)
template <int EPI, int DBG, bool F16 = false>  // F16: A and W hold IEEE fp16 (the LayerNorm-folded GEMMs read the fp16 residual stream)
__global__ __launch_bounds__(512, 2) void gemm256sp_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ outp,
                                                          const float* __restrict__ table, int T, int N, int K, int ntm,
                                                          int ntn, const float* __restrict__ rowscale, bf16* __restrict__ out16,
                                                          int raster, int tail_m0, int tail_nb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool DBG_TIMER = DBG == 16 || DBG == 19 || DBG == 20 || DBG == 21 || DBG == 22;
  constexpr bool DBG_L2HOT = DBG == 21;
  // L2 prefetch distance in K-tiles (0 = off: DBG 22 with the phase timer, 24 without).  Every CU prefetching all 512
  // lines of its K-tile instead of its share of the panels is slower than no prefetch at all (3155 vs 3041 vs 2816 cycles).
  constexpr int PFD = (DBG == 22 || DBG == 24) ? 0 : 4;
  constexpr int NPF = PFD ? 1 : 0;  // prefetch instructions per wave per K-tile  // every CU stages the operand panels of tile (0, 0) (garbage results)
  constexpr bool DBG_NO_STAGE = DBG == 1 || DBG == 3 || DBG == 19 || DBG == 20;
  constexpr bool DBG_NO_READ = DBG == 3 || DBG == 20;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;  // waves w and w+4 share a SIMD (measured placement); rows differ, harmless here
  const int hb = lane >> 5, l31 = lane & 31;
  const int ntiles = ntm * ntn;

  // ---- tile list of this block, XCD-aware: the 32 workgroups of an XCD (blockIdx % 8) walk 8 m-tiles x all n-tiles of a
  // group together, so the A panels and the W panels they share stay in that XCD's L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  // shA / shW: which quarter of its A panel / eighth of its W panel this CU prefetches into the L2 (the panel is shared
  // with the CUs of the XCD that work on the other n-tiles / m-tiles of the round)
  // Rasters (profiles/r02_gemm_raster.log, QKV 65536 x 3072 x 1024): every one reads 607-611 MB per launch past the L2
  // (FETCH_SIZE x 2) = the output-stationary bound of an XCD round of 8 x 4 tiles ((8 + 4) x 0.5 MB per 32 tiles = 576 MB):
  // a round streams 6 MB through a 4 MB L2, nothing of the previous round survives in it, the re-reads are served by the
  // 256 MB Infinity Cache.  What the order changes is WHEN panels are re-read:
  //   2 (the product's, when ntn % 4 == 0 and ntm % 64 == 0): an XCD owns the m-groups g = xcd (mod 8) and walks all
  //     n-slices of a group in consecutive rounds (its 4 MB of A panels are re-read while still hot in the Infinity
  //     Cache): steady K-tile 2681 cycles, QKV 1067 TF, fc1 1034 TF;
  //   0 (fallback for other shapes): chunks in m-group-major order dealt round-robin to the XCDs: 2842 cycles, 1045 / 1006 TF;
  //   1 (ablation only): the W slice is kept across rounds instead: 2798 cycles, 1024 / 991 TF.
  auto tile_of = [&](int j, int& m0, int& n0, int& shA, int& shW) -> bool {
    if ((raster & 3) != 0 && (ntn & 3) == 0 && (ntm & 63) == 0 && cpx == 32) {
      const int nsl = ntn >> 2, ng = ntm >> 6;  // n-slices of 4 tiles; m-groups (of 8 tiles) per XCD
      if (j >= nsl * ng) return false;
      const int sl = (raster & 3) == 1 ? j / ng : j % nsl, gi = (raster & 3) == 1 ? j % ng : j / nsl;
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

  // ---- staging: wave w fills rows [32w, 32w+32) of both operands, 8 rows (1 KiB) per instruction (piece 4w + j =
  // rows 32w + 8j .. +8).  Source chunk = LDS chunk position ^ ((row>>1)&7).  One per-lane byte offset per piece
  // (both operands are [rows, K] row-major, so they share them); the K-tile's base address is an SGPR pair and the
  // LDS destination (M0) is <wave base> + immediate: a stage is 8 x {s_add_u32 m0; s_nop 0; global_load_lds_dwordx4}.
  // (s_nop 0: M0 needs one wait state before the DMA reads it.  hipcc's hazard recognizer does not look inside inline
  // asm: if it ever reloads a spilled base SGPR with v_readlane right before this statement, the VALU-written SGPR
  // would need 5 wait states before the VMEM instruction reads it.  Padding every DMA for that costs 3 % of the steady
  // state (A/B: 2780 vs 2685 cycles per K-tile), so the generated code is linted instead: tools/check_isa.py, run by
  // tests/test_abi.py, fails on such a pair -- raise the s_nop here if it ever does.)
  const int srow = w * 32 + (lane >> 3);
  unsigned soff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = srow + 8 * j;
    soff[j] = (unsigned)((row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
  }
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
  const unsigned dmw = lds_base + w * 4096;  // this wave's first piece inside an operand tile
#define S_DMA(off, base, cimm)                                                                              \
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dmw), "n"(cimm) \
               : "memory", "scc")
#define S_STAGE(pM, pN, buf)                                      \
  if (!DBG_NO_STAGE) {                                            \
    S_DMA(soff[0], pM, (buf) * S_OPB);                            \
    S_DMA(soff[0], pN, (buf) * S_OPB + S_NBASE);                  \
    S_DMA(soff[1], pM, (buf) * S_OPB + 1024);                     \
    S_DMA(soff[1], pN, (buf) * S_OPB + S_NBASE + 1024);           \
    S_DMA(soff[2], pM, (buf) * S_OPB + 2048);                     \
    S_DMA(soff[2], pN, (buf) * S_OPB + S_NBASE + 2048);           \
    S_DMA(soff[3], pM, (buf) * S_OPB + 3072);                     \
    S_DMA(soff[3], pN, (buf) * S_OPB + S_NBASE + 3072);           \
  }                                                               \
  S_FENCE();

  // ---- L2 prefetch.  With operands that are already in the L2 a steady-state K-tile takes ~2340 cycles, with the real
  // stream ~2760-3090 (phase timer, DBG 21 vs 16): the DMA of a K-tile has one K-tile of lead, less than an L2 miss takes.
  // So every K-tile each wave also touches 12 lines of the K-tile PFD ahead (64 A rows + 32 W rows per CU: the CUs that
  // share a panel split it) with one dword LDS-DMA whose 256 bytes land in an unused corner of the wave's idle epilogue
  // scratch -- no destination register, and its miss latency is never waited for (vmcnt(NPF) at the K-tile's sync).
  auto pf_addr = [&](int tm0, int tn0, int sA, int sW) -> const char* {
    const int L = w * 12 + (lane % 12);
    return L < 64 ? reinterpret_cast<const char*>(A) + (size_t)(tm0 + 64 * sA + L) * K * 2
                  : reinterpret_cast<const char*>(W) + (size_t)(tn0 + 32 * sW + (L - 64)) * K * 2;
  };
  const unsigned pf_m0 = lds_base + S_SCRATCH + w * 4096 + 3584;
#define S_PF(ptr)                                                                                             \
  if (PFD) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(ptr), "s"(pf_m0) : "memory"); \
  S_FENCE();

  // ---- fragment read addresses (LDS byte addresses): one per (operand, k-step); buffer and fragment index are
  // immediates.  The reads are inline asm so that THIS file places the lgkmcnt waits (hipcc's own placement waits
  // lgkmcnt(0) right after issuing the next step's reads, which serialises LDS latency with the MFMAs).
  const int sw = (l31 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned fM[4], fN[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int xk = ((2 * kk + hb) ^ sw) << 4;
    fM[kk] = lds0 + (wr * 128 + l31) * 128 + xk;            // + buf*S_OPB + mi*4096
    fN[kk] = lds0 + S_NBASE + (wc * 64 + l31) * 128 + xk;   // + buf*S_OPB + ni*4096
  }

  // 16x16x32 form (CLIPX_MFMA16): row l15 of a 16-row half-block (+ 2 KiB per half, + 4 KiB per 32-row block), k-chunk 4 sl + q4 of
  // the 32-deep slab sl; (row >> 1) & 7 of the swizzle only depends on l15 (the block bases are multiples of 16 rows)
  const int l15 = lane & 15, q4 = lane >> 4;
  unsigned fM16[2], fN16[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int xk = ((4 * sl + q4) ^ ((l15 >> 1) & 7)) << 4;
    fM16[sl] = lds0 + (wr * 128 + l15) * 128 + xk;            // + buf*S_OPB + mi*4096 + h2*2048
    fN16[sl] = lds0 + S_NBASE + (wc * 64 + l15) * 128 + xk;   // + buf*S_OPB + ni*4096 + j2*2048
  }
  // accumulators of the wave's 128 x 64 tile: 32 x 32 blocks (mt, nt), quad g of a block, element e of a quad (gemm_common.h)
#if CLIPX_MFMA16
  f32x4 acc[8][4];  // [2 mt + m-half][2 nt + n-half]: one 16 x 16 MFMA block each
#define ACC(mt, nt, g, e) acc[2 * (mt) + ((g) >> 1)][2 * (nt) + ((g) & 1)][e]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  f32x16 acc[4][2];
#define ACC(mt, nt, g, e) acc[mt][nt][4 * (g) + (e)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#endif

  typedef int i32x4 __attribute__((ext_vector_type(4)));
#define S_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#if CLIPX_MFMA16
  // A 64-deep K-tile is eight units u = 4 * slab + mi: unit (sl, mi) multiplies the slab's four weight fragments Nd[sl][2 ni + j2]
  // (16 rows x 32 k each) with the two activation fragments Md[u & 1][h2] of the 32-row block mi: 8 v_mfma_f32_16x16x32 = 128
  // matrix-pipe cycles.  The fragments of unit u + 1 are read while unit u multiplies: 2 reads (the next block's activations), or
  // 6 (+ the next slab's weights).  48 fragment registers, as the 32x32x16 loop had: with 16-MFMA units (64 registers) hipcc
  // spilled around the tile boundaries.  Fragments are held as 4 x b32 so that each stays one 128-bit register tuple.
  i32x4 Nd[2][4], Md[2][2];
  if (DBG_NO_READ) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Nd[0][i] = Nd[1][i] = i32x4{0, 0, 0, 0};
      asm volatile("" : "+v"(Nd[0][i]), "+v"(Nd[1][i]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Md[0][i] = Md[1][i] = i32x4{0, 0, 0, 0};
      asm volatile("" : "+v"(Md[0][i]), "+v"(Md[1][i]));
    }
  }
#define S_READ_N(sl, buf)                                              \
  S_DSREAD(Nd[sl][0], fN16[sl], (buf) * S_OPB);                        \
  S_DSREAD(Nd[sl][1], fN16[sl], (buf) * S_OPB + 2048);                 \
  S_DSREAD(Nd[sl][2], fN16[sl], (buf) * S_OPB + 4096);                 \
  S_DSREAD(Nd[sl][3], fN16[sl], (buf) * S_OPB + 6144);
#define S_READ_M(u, buf)                                                                     \
  S_DSREAD(Md[(u) & 1][0], fM16[(u) >> 2], (buf) * S_OPB + ((u) & 3) * 4096);                  \
  S_DSREAD(Md[(u) & 1][1], fM16[(u) >> 2], (buf) * S_OPB + ((u) & 3) * 4096 + 2048);
// unit u of the K-tile in buffer `buf`: u = 0 / 4 also bring the slab's weight fragments
#define S_READ_U(u, buf)                                                 \
  if (!DBG_NO_READ) {                                                    \
    if (((u) & 3) == 0) { S_READ_N((u) >> 2, buf) }                      \
    S_READ_M(u, buf)                                                     \
  }                                                                      \
  S_FENCE();
#define S_WAIT_N(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); S_FENCE();
#define S_MFMA_U(u)                                                                                                            \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int h2 = 0; h2 < 2; ++h2)                            \
      _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2)                                                                         \
          acc[2 * ((u) & 3) + h2][2 * ni + j2] =                                                                               \
              mfma_16x16x32<F16>(Nd[(u) >> 2][2 * ni + j2], Md[(u) & 1][h2], acc[2 * ((u) & 3) + h2][2 * ni + j2]);            \
  S_FENCE();
  // One K-tile = S_KT_HEAD (units 0..6 and the LDS drain), a vmcnt wait + barrier, the stage of a later K-tile into the buffer just
  // released, then S_KT_TAIL (pre-read of the next K-tile's first unit + unit 7's MFMAs).
#define S_KT_HEAD(buf)                                       \
  S_READ_U(1, buf) S_WAIT_N(2) S_MFMA_U(0)                   \
  S_READ_U(2, buf) S_WAIT_N(2) S_MFMA_U(1)                   \
  S_READ_U(3, buf) S_WAIT_N(2) S_MFMA_U(2)                   \
  S_READ_U(4, buf) S_WAIT_N(6) S_MFMA_U(3)                   \
  S_READ_U(5, buf) S_WAIT_N(2) S_MFMA_U(4)                   \
  S_READ_U(6, buf) S_WAIT_N(2) S_MFMA_U(5)                   \
  S_READ_U(7, buf) S_WAIT_N(2) S_MFMA_U(6)                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  S_FENCE();
#define S_READ_FIRST(buf) S_READ_U(0, buf)
#define S_MFMA_LAST() S_MFMA_U(7)
#else
  // [0..3] M fragments (mi), [4..5] N fragments (ni) of one k-step; held as 4 x b32 so that hipcc keeps each
  // fragment one 128-bit register tuple across the loop back-edge (as 8 x bf16 it re-packs them with v_perm_b32)
  i32x4 F0[6], F1[6];
  if (DBG_NO_READ) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      F0[i] = F1[i] = i32x4{0, 0, 0, 0};
      asm volatile("" : "+v"(F0[i]), "+v"(F1[i]));
    }
  }
#define S_READ(F, buf, kk)                                                                 \
  if (!DBG_NO_READ) {                                                                      \
    S_DSREAD(F[4], fN[kk], (buf) * S_OPB);                                                 \
    S_DSREAD(F[5], fN[kk], (buf) * S_OPB + 4096);                                          \
    S_DSREAD(F[0], fM[kk], (buf) * S_OPB);                                                 \
    S_DSREAD(F[1], fM[kk], (buf) * S_OPB + 4096);                                          \
    S_DSREAD(F[2], fM[kk], (buf) * S_OPB + 8192);                                          \
    S_DSREAD(F[3], fM[kk], (buf) * S_OPB + 12288);                                         \
  }                                                                                        \
  S_FENCE();
// the fragment set read one step earlier has landed when at most the 6 reads issued since are outstanding
#define S_WAIT_PREV() asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); S_FENCE();
#define S_MFMA(F)                                                                                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      mfma_32x32x16<F16>(F[4 + ni], F[mi], acc[mi][ni]);                                                            \
  S_FENCE();

  // One K-tile = S_KT_HEAD (k-steps 0..2 and the LDS drain), a vmcnt wait + barrier, the stage of a later K-tile into
  // the buffer just released, then S_KT_TAIL (pre-read of the next K-tile's first fragments + k-step 3's MFMAs).
#define S_KT_HEAD(buf)                                       \
  S_READ(F1, buf, 1) S_WAIT_PREV() S_MFMA(F0)                \
  S_READ(F0, buf, 2) S_WAIT_PREV() S_MFMA(F1)                \
  S_READ(F1, buf, 3) S_WAIT_PREV() S_MFMA(F0)                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  S_FENCE();
#define S_READ_FIRST(buf) S_READ(F0, buf, 0)
#define S_MFMA_LAST() S_MFMA(F1)
#endif
#define S_KT_SYNC(vm)                                        \
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(vm) : "memory");  \
  S_FENCE();                                                 \
  __builtin_amdgcn_s_barrier();                              \
  S_FENCE();
#define S_KT_TAIL(buf, preread)                              \
  if (preread) { S_READ_FIRST((buf) ^ 1) }                   \
  S_MFMA_LAST()
// the stage of a K-tile goes behind k-step 3's MFMAs so that its 8 DMA issues overlap with their execution (measured:
// the same as staging right behind the barrier, 2835 vs 2845 cycles per steady K-tile)
#define S_KT_END(buf, preread, stage_stmt)                   \
  S_KT_TAIL(buf, preread)                                    \
  stage_stmt

  const int nk = K >> 6;  // K-tiles per output tile (even, >= 2)
  const char* curM = reinterpret_cast<const char*>(A) + (DBG_L2HOT ? (size_t)0 : (size_t)m0 * K * 2);
  const char* curN = reinterpret_cast<const char*>(W) + (DBG_L2HOT ? (size_t)0 : (size_t)n0 * K * 2);

  // VMEM bookkeeping (vmcnt retires in order): "the stage of the next K-tile has landed" is vmcnt(0) in steady state.
  // On the first K-tile after an epilogue the epilogue's own VMEM operations are younger than that stage and may stay in
  // flight: vmcnt(S_EPI_VM).  bf16 outputs: 16 stores.  f32 residual: 32 loads + 32 stores, the loads consumed already.
  constexpr bool OUT_BF16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_F16;  // 16-bit outputs
  constexpr int S_EPI_VM = (DBG == 5 || DBG == 6) ? 0 : ((OUT_BF16 || EPI == EPI_BIAS_RESID_H16) ? 16 : (EPI == EPI_BIAS_RESID_F32 ? 40 : 32));

  // ---- prologue: K-tiles 0, 1 of the first tile; K-tile 0 landed + first fragment set read
  S_STAGE(curM, curN, 0)
  S_STAGE(curM + 128, curN + 128, 1)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  S_FENCE();
  __builtin_amdgcn_s_barrier();
  S_FENCE();
  S_READ_FIRST(0)

  // bias of the current tile: one LDS-DMA per wave at the tile's last K-tile drops the wave's 64 bias floats into
  // its (then idle) epilogue scratch, AHEAD of the next tile's K-tile-1 stage, so the epilogue neither queues
  // behind that nor holds bias registers across the main loop.  (All four 16-lane groups fetch the same 256 B.)
  // Inline asm: hipcc must not know about this DMA, or it waits vmcnt(0) before the epilogue's first LDS read.
  constexpr bool HAS_BIAS = EPI != EPI_TABLE_F32;
  const int rrow = lane >> 3, rch = lane & 7;  // epilogue read-back: row 8i + rrow, 16-B chunk rch
  const unsigned bias_m0 = lds_base + S_SCRATCH + w * 4096;
  const unsigned bias_off = (unsigned)((lane & 15) * 16);
  // bf16-output epilogues also fetch the 128 row scales of the wave's panel (out = act(acc * rowscale[m] + bias[n]): the
  // LayerNorm 1/std of a LayerNorm-folded GEMM) the same way, 1 KiB further into the scratch (both 32-lane halves fetch
  // the same 512 B)
  const unsigned rs_m0 = lds_base + S_SCRATCH + w * 4096 + 1024;
  const unsigned rs_off = (unsigned)((lane & 31) * 16);
  auto load_bias = [&]() {
    if (!HAS_BIAS) return;
    const float* bp = bias + n0 + wc * 64;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(bias_off), "s"(bp), "s"(bias_m0) : "memory");
    if (OUT_BF16) {
      const float* rp = rowscale + m0 + wr * 128;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(rs_off), "s"(rp), "s"(rs_m0) : "memory");
    }
  };

  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tstamp = DBG_TIMER ? (long long)__builtin_readcyclecounter() : 0;
  const long long tstart = tstamp;
#define S_STAMP(slot)                                                  \
  if (DBG_TIMER) {                                                     \
    const long long now_ = (long long)__builtin_readcyclecounter();    \
    ph[slot] += now_ - tstamp;                                         \
    tstamp = now_;                                                     \
  }

  const char* sM = curM + 256;  // the next K-tile of this tile to stage (K-tile 2)
  const char* sN = curN + 256;
  // prefetch stream: at K-tile t the prefetch targets K-tile t + PFD of the block's K-tile stream; it moves on to the next
  // tile's panels after K-tile nk - PFD - 1.  Short K (< 8 K-tiles): the pointer stays on valid lines of K-tile 0.
  // (operands of a few tens of MB mostly stay in the L2 / Infinity Cache anyway: there the extra requests cost ~3 %)
  const bool pf_run = PFD && nk >= 8 && (size_t)ntm * 256 * K * 2 >= ((size_t)64 << 20);
  const int pf_sw = pf_run ? nk - PFD - 1 : -1;
  const int pf_step = pf_run ? 128 : 0;
  const char* pfp = pf_addr(m0, n0, shA0, shW0) + (pf_run ? PFD * 128 : 0);
#define S_PF_NEXT(t) pfp = ((t) == pf_sw) ? pf_nxt : pfp + pf_step;
  bool first = false;  // the K-tile about to run is the first one after an epilogue
  for (int j = 0;; ++j) {
    int nm0 = 0, nn0 = 0, nshA = 0, nshW = 0;
    const bool have_next = tile_of(j + 1, nm0, nn0, nshA, nshW);
    const char* nxtM = reinterpret_cast<const char*>(A) + (DBG_L2HOT ? (size_t)0 : (size_t)nm0 * K * 2);
    const char* nxtN = reinterpret_cast<const char*>(W) + (DBG_L2HOT ? (size_t)0 : (size_t)nn0 * K * 2);
    const char* pf_nxt = have_next ? pf_addr(nm0, nn0, nshA, nshW) : pf_addr(m0, n0, 0, 0);

    if (nk > 2) {
      // ---- first pair (K-tiles 0, 1): stage K-tiles 2, 3
      S_KT_HEAD(0)
      if (first) { S_KT_SYNC(S_EPI_VM) } else { S_KT_SYNC(0) }
      S_KT_END(0, true, S_STAGE(sM, sN, 0))
      S_PF(pfp)
      S_PF_NEXT(0)
      S_STAMP(1)
      S_KT_HEAD(1)
      S_KT_SYNC(NPF)
      S_KT_END(1, true, S_STAGE(sM + 128, sN + 128, 1))
      S_PF(pfp)
      S_PF_NEXT(1)
      S_STAMP(2)
      sM += 256;
      sN += 256;
      // ---- steady state (K-tiles 2 .. nk-3): branch-free
      for (int t = 2; t < nk - 2; t += 2) {
        S_KT_HEAD(0)
        S_KT_SYNC(NPF)
        S_KT_END(0, true, S_STAGE(sM, sN, 0))
        S_PF(pfp)
        S_PF_NEXT(t)
        S_KT_HEAD(1)
        S_KT_SYNC(NPF)
        S_KT_END(1, true, S_STAGE(sM + 128, sN + 128, 1))
        S_PF(pfp)
        S_PF_NEXT(t + 1)
        sM += 256;
        sN += 256;
        if (DBG_TIMER) { S_STAMP(3) ph[5] += 2; }
      }
      // ---- last pair (K-tiles nk-2, nk-1): stage the next tile's K-tiles 0, 1; fetch this tile's bias.  No prefetch (the
      // epilogue's entry wait would have to sit out its miss); the stream pointer still advances.
      S_KT_HEAD(0)
      S_KT_SYNC(NPF)
      load_bias();  // lands before the next sync: the epilogue does not have to wait for it
      S_KT_END(0, true, if (have_next) { S_STAGE(nxtM, nxtN, 0) })
      S_KT_HEAD(1)
      S_KT_SYNC(0)
      S_KT_END(1, false, if (have_next) { S_STAGE(nxtM + 128, nxtN + 128, 1) })  // the next tile's first fragments are read after the epilogue (24 VGPRs it needs first)
      S_PF_NEXT(nk - 2)
      S_PF_NEXT(nk - 1)
      S_STAMP(4)
    } else {
      // ---- K = 128: the only pair is the first and the last one (no prefetch)
      S_KT_HEAD(0)
      if (first) { S_KT_SYNC(S_EPI_VM) } else { S_KT_SYNC(0) }
      load_bias();
      S_KT_END(0, true, if (have_next) { S_STAGE(nxtM, nxtN, 0) })
      S_KT_HEAD(1)
      S_KT_SYNC(0)
      S_KT_END(1, false, if (have_next) { S_STAGE(nxtM + 128, nxtN + 128, 1) })
      S_STAMP(4)
    }

    // ---- epilogue of this output tile, transposed through the wave's LDS scratch (the next tile's K-tile 0 has landed,
    // its K-tile 1 is in flight; the bias landed in the scratch before the last K-tile's sync)
    S_FENCE();
    // Where quad g (four consecutive columns of one row, gemm_common.h) of the 32 x 32 block (mt, nt) sits in the wave's transposition
    // scratch -- rows of 128 B, one 32-row pass at a time -- and which bias / row-scale entries it needs, for both MFMA forms:
#if CLIPX_MFMA16
#define Q_ROW(g) (16 * ((g) >> 1) + l15)                        // row inside the 32-row pass
#define Q_CH16(nt, g) (4 * (nt) + 2 * ((g) & 1) + (q4 >> 1))    // 16-B chunk of eight 16-bit outputs (64 columns = 8 chunks)
#define Q_HALF(g) (q4 & 1)                                      // 8-B half of that chunk
#define Q_CH32(g) (4 * ((g) & 1) + q4)                          // 16-B chunk of four f32 outputs inside the 32-column block
#define Q_NCOL(nt, g) ((nt) * 32 + 16 * ((g) & 1) + 4 * q4)     // first column of the quad inside the wave's 64 columns
#else
#define Q_ROW(g) l31
#define Q_CH16(nt, g) (4 * (nt) + (g))
#define Q_HALF(g) hb
#define Q_CH32(g) (2 * (g) + hb)
#define Q_NCOL(nt, g) ((nt) * 32 + 8 * (g) + 4 * hb)
#endif
#define Q_POS16(nt, g) (Q_ROW(g) * 128 + ((Q_CH16(nt, g) ^ (Q_ROW(g) & 7)) << 4) + Q_HALF(g) * 8)
#define Q_POS32(g) (Q_ROW(g) * 128 + ((Q_CH32(g) ^ (Q_ROW(g) & 7)) << 4))
    if (DBG != 5) {
      unsigned char* scr = smem + S_SCRATCH + w * 4096;
      if (OUT_BF16) {
        // SGPR base + one 32-bit VGPR offset: no per-row address registers
        const char* yb = reinterpret_cast<const char*>(outp) + ((size_t)(m0 + wr * 128) * N + n0 + wc * 64) * 2;
        const unsigned voff = (unsigned)((rrow * N + rch * 8) * 2);
        const size_t rstep = (size_t)8 * N * 2;  // 8 rows
        float4 b4[2][4];  // read before the first transposition pass overwrites the scratch (LDS ops stay in order)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[nt][g] = *reinterpret_cast<const float4*>(scr + Q_NCOL(nt, g) * 4);
        float rr[4][2];  // row scales of the lane's rows 32 mt + Q_ROW(g): one per row half (g >> 1) in the 16x16x32 form
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) rr[mt][hh] = *reinterpret_cast<const float*>(scr + 1024 + (mt * 32 + Q_ROW(2 * hh)) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 bq = b4[nt][g];
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              float v[4];
              if (EPI == EPI_BIAS_QGELU_BF16) {
                // two elements per VALU instruction for everything but the two transcendentals (v_pk_fma_f32, v_pk_mul_f32,
                // v_pk_add_f32): each packed lane is the IEEE operation of gemm_common.h's quick_gelu, so the bits do not change
                const f32x2_t r2 = {rr[mt][g >> 1], rr[mt][g >> 1]}, kk = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};
                const f32x2_t one = {1.f, 1.f};
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                  const f32x2_t a2 = {ACC(mt, nt, g, e), ACC(mt, nt, g, e + 1)};
                  const f32x2_t b2 = {e == 0 ? bq.x : bq.z, e == 0 ? bq.y : bq.w};
                  const f32x2_t x2 = __builtin_elementwise_fma(a2, r2, b2);
                  const f32x2_t t2 = x2 * kk;
                  const f32x2_t d2 = (f32x2_t){__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])} + one;
                  const f32x2_t y2 = x2 * (f32x2_t){__builtin_amdgcn_rcpf(d2[0]), __builtin_amdgcn_rcpf(d2[1])};
                  v[e] = y2[0];
                  v[e + 1] = y2[1];
                }
              } else {
                // one fma per element, like gemm_store_quad (gemm_common.h): bit-identical rows from both kernels
                v[0] = __builtin_fmaf(ACC(mt, nt, g, 0), rr[mt][g >> 1], bq.x);
                v[1] = __builtin_fmaf(ACC(mt, nt, g, 1), rr[mt][g >> 1], bq.y);
                v[2] = __builtin_fmaf(ACC(mt, nt, g, 2), rr[mt][g >> 1], bq.z);
                v[3] = __builtin_fmaf(ACC(mt, nt, g, 3), rr[mt][g >> 1], bq.w);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (EPI == EPI_BIAS_GELU_BF16) v[e] = gelu_erf(v[e]);
              }
              // two v_cvt_pk_bf16_f32 per quad (element-wise casts into a bf16x4 make hipcc convert one value at a time
              // and assemble the pairs with v_perm / v_alignbit: 9 VALU per quad instead of 4)
              uint2 o;
              if (EPI == EPI_BIAS_F16) {  // v_cvt_pk_f16_f32: IEEE fp16, round to nearest even (the same bits as gemm_store_quad)
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));  // never v_fma_mix*_f16 (see gemm_store_quad)
                typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, f16x2_t));
                o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, f16x2_t));
              } else {
                o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
                o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
              }
              // row l31 = [8 chunks of 16 B]; chunk (4nt + g) holds columns 32nt + 8g .. +8, half hb
              *reinterpret_cast<uint2*>(scr + Q_POS16(nt, g)) = o;
            }
          u32x4 q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            q[i] = *reinterpret_cast<const u32x4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const char* base = yb + (size_t)(mt * 4 + i) * rstep;
            // s_nop 4: the base SGPR may come straight from a v_readlane; s_nop 1: store-data hazard (hipcc cannot see that
            // this is a store and may rewrite q right behind it)
            if (DBG == 31) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(q[i]), "s"(base) : "memory");  // ablation: streaming stores
            else if (DBG == 32) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" ::"v"(voff), "v"(q[i]), "s"(base) : "memory");
            else if (DBG == 33) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(q[i]), "s"(base) : "memory");
            else if (DBG != 6) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(q[i]), "s"(base) : "memory");
            else asm volatile("" ::"v"(q[i]));
          }
        }
      } else if (EPI == EPI_BIAS_RESID_F32) {
        // f32 in-place residual x += acc + bias: 8 passes of a 32 x 32 sub-tile (128 B per row).  The x rows of pass p+1 are
        // requested before pass p stores.  Buffer loads / stores (SGPR descriptor of this wave's 128-row panel + SGPR
        // offset of the 8-row group + one 32-bit lane offset): no address registers, and hipcc counts them itself -- with
        // asm loads the register allocator may copy a result register before a hand-placed wait (seen: nondeterministic
        // results), and with asm stores hipcc's own counts would wait for the stores.
        char* xb = reinterpret_cast<char*>(outp) + ((size_t)(m0 + wr * 128) * N + n0 + wc * 64) * 4;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 0x7ffffffe, 0x00020000);
        const int voff = (rrow * N + rch * 4) * 4;
        const int rstep = 8 * N * 4;  // 8 rows
        // bf16 shadow of the new x rows (read by the next LayerNorm-folded GEMM): 8 B per lane, same lane -> element map
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        bf16* xb16 = out16 ? out16 + ((size_t)(m0 + wr * 128) * N + n0 + wc * 64) : reinterpret_cast<bf16*>(xb);
        const __amdgpu_buffer_rsrc_t xr16 = __builtin_amdgcn_make_buffer_rsrc(xb16, 0, 0x7ffffffe, 0x00020000);
        // Full 128-B lines: a row's 64 shadow columns of this wave are produced in two passes (nt = 0, 1), 4 columns per
        // lane each; lane pairs (rch, rch ^ 1) swap one 8-B half so that the even lane owns columns 4 rch .. 4 rch + 7 of
        // the nt = 0 half and the odd lane columns 32 + 4 (rch - 1) .. + 7 of the nt = 1 half: ONE 16-B store per row pair
        // instead of two 8-B half-line stores (the store pipe is what this epilogue waits for).
        const bool odd = (rch & 1) != 0;
        const int voff16 = rrow * N * 2 + (odd ? 64 + (rch - 1) * 8 : rch * 8);
        const int rstep16 = 8 * N * 2;
        const bool shadow = out16 != nullptr;
        u32x2 hkeep[4];
        float4 b4[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b4[nt] = *reinterpret_cast<const float4*>(scr + (nt * 32 + rch * 4) * 4);
        S_FENCE();  // hipcc waits vmcnt(0) for the bias DMA before these LDS reads: keep the x loads behind that wait
        // x rows are requested RESID_LA passes ahead of their use (the fragment registers are dead here: room for it)
#ifndef CLIPX_RESID_LA
#define CLIPX_RESID_LA 1
#endif
        constexpr int LA = (DBG >= 41 && DBG <= 43) ? DBG - 40 : CLIPX_RESID_LA;  // DBG 41..43: A/B of the lookahead
        u32x4 ext[LA + 1][4];
#define S_LD_EXT(set, p)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    ext[set][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, (((p) >> 1) * 4 + i) * rstep + ((p) & 1) * 128, 0);
#pragma unroll
        for (int p = 0; p < LA; ++p) { S_LD_EXT(p, p) }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int mt = p >> 1, nt = p & 1;
          if (p + LA < 8) { S_LD_EXT((p + LA) % (LA + 1), p + LA) }
          S_FENCE();
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = make_float4(ACC(mt, nt, g, 0), ACC(mt, nt, g, 1), ACC(mt, nt, g, 2),
                                         ACC(mt, nt, g, 3));
            // columns 8g + 4hb .. +4 = 16-B chunk 2g + hb of row l31
            *reinterpret_cast<float4*>(scr + Q_POS32(g)) = v;
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
            const u32x4 e = ext[p % (LA + 1)][i];
            // same association as the 128x128 kernel, x + (acc + bias), so a row's result does not depend on
            // which kernel (i.e. which batch chunking) produced it
            float4 o = q[i];
            o.x += bq.x; o.y += bq.y; o.z += bq.z; o.w += bq.w;
            o.x = __uint_as_float(e[0]) + o.x; o.y = __uint_as_float(e[1]) + o.y;
            o.z = __uint_as_float(e[2]) + o.z; o.w = __uint_as_float(e[3]) + o.w;
            const u32x4 ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            __builtin_amdgcn_raw_buffer_store_b128(ov, xr, voff, (mt * 4 + i) * rstep + nt * 128, 0);
            S_STORE_GUARD(ov);
            if (shadow) {
              typedef float f32x2_t __attribute__((ext_vector_type(2)));
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              const bf16x2_t h01 = __builtin_convertvector((f32x2_t){o.x, o.y}, bf16x2_t);
              const bf16x2_t h23 = __builtin_convertvector((f32x2_t){o.z, o.w}, bf16x2_t);
              const u32x2 hv = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
#ifdef CLIPX_ABLATE
              if (raster & 4) {  // A/B: two 8-B half-line stores per row instead of the paired 16-B store
                __builtin_amdgcn_raw_buffer_store_b64(hv, xr16, (rrow * N + rch * 4) * 2, (mt * 4 + i) * rstep16 + nt * 64, 0);
                S_STORE_GUARD(hv);
              } else
#endif
              if (nt == 0) {
                hkeep[i] = hv;
              } else {
                const u32x2 send = odd ? hkeep[i] : hv;  // odd lanes give their nt = 0 half, even lanes their nt = 1 half
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, false);
                const u32x4 full = odd ? u32x4{r0, r1, hv.x, hv.y} : u32x4{hkeep[i].x, hkeep[i].y, r0, r1};
                __builtin_amdgcn_raw_buffer_store_b128(full, xr16, voff16, (mt * 4 + i) * rstep16, 0);
                S_STORE_GUARD(full);
              }
            }
          }
          S_FENCE();
        }
      } else if (EPI == EPI_BIAS_RESID_H16) {
        // fp16 in-place residual x16 = fp16(f32(x16) + (acc + bias)): the residual stream is stored ONCE, in IEEE fp16 (what the
        // next LayerNorm-folded GEMM reads as its A operand), so this epilogue moves 128 KiB in + 128 KiB out per tile instead of
        // the 256 + 256 + 128 KiB of the f32 stream + bf16 shadow (EPI_BIAS_RESID_F32 above).  Four passes of a 32-row x 64-column
        // sub-tile (one 128-B line per row): the old x rows arrive by 16-B buffer loads (lane = row 8i + rrow, chunk rch: whole
        // lines), go through the wave's LDS scratch into the accumulator layout (lane = row l31, 4 columns), are added in f32,
        // rounded, written back to the same scratch words and leave as 16-B buffer stores.  Same association as the 128x128
        // kernel (gemm_store_quad): x + (acc + bias).
        _Float16* xb = reinterpret_cast<_Float16*>(outp) + ((size_t)(m0 + wr * 128) * N + n0 + wc * 64);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xb, 0, 0x7ffffffe, 0x00020000);
        const int voff = (rrow * N + rch * 8) * 2;
        const int rstep = 8 * N * 2;  // 8 rows
        float4 b4[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[nt][g] = *reinterpret_cast<const float4*>(scr + Q_NCOL(nt, g) * 4);
        S_FENCE();  // hipcc waits vmcnt(0) for the bias DMA before these LDS reads: keep the x loads behind that wait
        u32x4 ext[2][4];
#define S_LD_X16(set, mt_)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
    ext[set][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, ((mt_) * 4 + i) * rstep, 0);
        S_LD_X16(0, 0)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (mt + 1 < 4) { S_LD_X16((mt + 1) & 1, mt + 1) }
          S_FENCE();
          // old rows -> scratch, in the position the read-back below (and the final store) uses
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            *reinterpret_cast<u32x4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4)) = ext[mt & 1][i];
          }
          // read all eight 8-B slots of this lane first, then add, then write them back: written as three loops so that the
          // LDS round trips overlap (one slot at a time hipcc emits read; wait; write eight times in a row)
          uint2 xo[2][4];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)  // row l31, columns 32 nt + 8 g + 4 hb .. + 4: chunk (4 nt + g), half hb
              xo[nt][g] = *reinterpret_cast<const uint2*>(scr + Q_POS16(nt, g));
          S_FENCE();
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f16x4 xh = __builtin_bit_cast(f16x4, xo[nt][g]);
              const float4 bq = b4[nt][g];
              f16x4 o;
              o[0] = (_Float16)((float)xh[0] + (ACC(mt, nt, g, 0) + bq.x));
              o[1] = (_Float16)((float)xh[1] + (ACC(mt, nt, g, 1) + bq.y));
              o[2] = (_Float16)((float)xh[2] + (ACC(mt, nt, g, 2) + bq.z));
              o[3] = (_Float16)((float)xh[3] + (ACC(mt, nt, g, 3) + bq.w));
              xo[nt][g] = __builtin_bit_cast(uint2, o);
            }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<uint2*>(scr + Q_POS16(nt, g)) = xo[nt][g];
          S_FENCE();
          u32x4 q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            q[i] = *reinterpret_cast<const u32x4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
          }
          S_FENCE();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_buffer_store_b128(q[i], xr, voff, (mt * 4 + i) * rstep, 0);
            S_STORE_GUARD(q[i]);
          }
          S_FENCE();
        }
#undef S_LD_X16
      } else {
        // + table row (patch embedding: class token / positional rows), f32 out: once per forward, plain code
        float* xo = reinterpret_cast<float*>(outp) + (size_t)(m0 + wr * 128) * N + n0 + wc * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            float4 ext[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = mt * 32 + 8 * i + rrow;
              ext[i] = *reinterpret_cast<const float4*>(table + (size_t)((m0 + wr * 128 + row) % T) * N + n0 + wc * 64 +
                                                        nt * 32 + rch * 4);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = make_float4(ACC(mt, nt, g, 0), ACC(mt, nt, g, 1), ACC(mt, nt, g, 2),
                                           ACC(mt, nt, g, 3));
              *reinterpret_cast<float4*>(scr + Q_POS32(g)) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 8 * i + rrow;
              float4 q = *reinterpret_cast<const float4*>(scr + row * 128 + ((rch ^ (row & 7)) << 4));
              q.x = ext[i].x + q.x; q.y = ext[i].y + q.y; q.z = ext[i].z + q.z; q.w = ext[i].w + q.w;
              *reinterpret_cast<float4*>(xo + (size_t)(mt * 32 + row) * N + nt * 32 + rch * 4) = q;
            }
          }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(ACC(mt, nt, g, 0)), "v"(ACC(mt, nt, g, 1)), "v"(ACC(mt, nt, g, 2)), "v"(ACC(mt, nt, g, 3)));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ACC(mt, nt, r >> 2, r & 3) = 0.f;
    if (DBG_TIMER) { S_STAMP(0) ph[6] += 1; }
    if (!have_next) break;
    S_READ_FIRST(0)  // first fragment set of the next tile (its K-tile 0 landed before the barrier of this tile's last K-tile);
                      // asking for it earlier, inside the epilogue, does not help: hipcc sinks the block behind the last store
    first = true;
    m0 = nm0;
    n0 = nn0;
    sM = nxtM + 256;
    sN = nxtN + 256;
  }
  // ---- the ragged 257th m-tile (see gemm256_tail above): every workgroup takes one 32-row strip of it
  if (tail_nb > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last epilogue's stores count in vmcnt: start the tail's arithmetic at zero
    S_FENCE();
    __builtin_amdgcn_s_barrier();                     // every wave is done with the K-tile buffers
    S_FENCE();
    if (tail_nb == 1) gemm256_tail<EPI, F16, 1>(A, W, bias, outp, table, T, N, K, rowscale, out16, tail_m0, smem, lds_base, w, lane);
    else if (tail_nb == 2) gemm256_tail<EPI, F16, 2>(A, W, bias, outp, table, T, N, K, rowscale, out16, tail_m0, smem, lds_base, w, lane);
    else if (tail_nb == 3) gemm256_tail<EPI, F16, 3>(A, W, bias, outp, table, T, N, K, rowscale, out16, tail_m0, smem, lds_base, w, lane);
    else gemm256_tail<EPI, F16, 4>(A, W, bias, outp, table, T, N, K, rowscale, out16, tail_m0, smem, lds_base, w, lane);
  }
  if (DBG_TIMER && tid == 0 && blockIdx.x < 2048) {
    ph[7] = (long long)__builtin_readcyclecounter() - tstart;
#pragma unroll
    for (int i = 0; i < 8; ++i) g_sp_phase[blockIdx.x * 8 + i] = ph[i];
  }
}

#ifdef CLIPX_ABLATE
// debug hook (tools build only; not part of include/clipx.h): copies the DBG 16 phase counters of the last launch to the host
extern "C" int clipx_dbg_phase_cycles(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sp_phase), (size_t)n * sizeof(long long));
}
#endif

template <int EPI, int DBG = 0, bool F16 = false>
static hipError_t launch_sp_epi(const GemmArgs& g, int grid, hipStream_t st) {
  int raster = 2;
#ifdef CLIPX_ABLATE
  if (const char* fl = getenv("CLIPX_GEMM_FLAGS")) raster = atoi(fl) & 7;  // bits 0-1: raster, bit 2: 8-B shadow stores
#endif
  const size_t smem = S_SCRATCH + 8 * 4096;  // 160 KiB: the whole LDS of the CU
  auto kern = gemm256sp_kernel<EPI, DBG, F16>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.N, g.K, g.M / 256,
                     g.N / 256, g.rowscale, g.out16, raster, g.tail_m0, g.tail_nb);
  return hipGetLastError();
}

hipError_t launch_gemm256sp(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (g.M <= 0 || g.M % 256 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K <= 0) return hipErrorInvalidValue;
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;  // one workgroup per CU; multiple of the 8 XCDs
  if (grid < 8) grid = 8;
#ifdef CLIPX_ABLATE
  // Ablation / phase-timer instantiations exist only in the tools build (make ablate -> lib/libclipx_ablate.so); the
  // product library has no environment switch that changes what the hot path computes.
  if (g.epi == EPI_BIAS_BF16) {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) return launch_sp_epi<EPI_BIAS_BF16, 1>(g, grid, st);
    if (d == 3) return launch_sp_epi<EPI_BIAS_BF16, 3>(g, grid, st);
    if (d == 5) return launch_sp_epi<EPI_BIAS_BF16, 5>(g, grid, st);
    if (d == 6) return launch_sp_epi<EPI_BIAS_BF16, 6>(g, grid, st);
    if (d == 16) return launch_sp_epi<EPI_BIAS_BF16, 16>(g, grid, st);
    if (d == 19) return launch_sp_epi<EPI_BIAS_BF16, 19>(g, grid, st);
    if (d == 20) return launch_sp_epi<EPI_BIAS_BF16, 20>(g, grid, st);
    if (d == 21) return launch_sp_epi<EPI_BIAS_BF16, 21>(g, grid, st);
    if (d == 22) return launch_sp_epi<EPI_BIAS_BF16, 22>(g, grid, st);
    if (d == 24) return launch_sp_epi<EPI_BIAS_BF16, 24>(g, grid, st);
    if (d == 31) return launch_sp_epi<EPI_BIAS_BF16, 31>(g, grid, st);  // store cache policies: nt / sc1 nt / sc0 sc1 (correct results)
    if (d == 32) return launch_sp_epi<EPI_BIAS_BF16, 32>(g, grid, st);
    if (d == 33) return launch_sp_epi<EPI_BIAS_BF16, 33>(g, grid, st);
  }
  if (g.epi == EPI_BIAS_RESID_F32) {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 16) return launch_sp_epi<EPI_BIAS_RESID_F32, 16>(g, grid, st);  // phase timer
    if (d == 41) return launch_sp_epi<EPI_BIAS_RESID_F32, 41>(g, grid, st);  // x-load lookahead 1 / 2 / 3 passes
    if (d == 42) return launch_sp_epi<EPI_BIAS_RESID_F32, 42>(g, grid, st);
    if (d == 43) return launch_sp_epi<EPI_BIAS_RESID_F32, 43>(g, grid, st);
  }
  {  // phase timer of the encoder's own forms (fp16 operands: QKV / fc1; fp16 in-place residual: out_proj / fc2)
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    if (dbg && atoi(dbg) == 16) {
      if (g.f16 && g.epi == EPI_BIAS_F16) return launch_sp_epi<EPI_BIAS_F16, 16, true>(g, grid, st);
      if (g.f16 && g.epi == EPI_BIAS_QGELU_BF16) return launch_sp_epi<EPI_BIAS_QGELU_BF16, 16, true>(g, grid, st);
      if (g.f16 && g.epi == EPI_BIAS_BF16) return launch_sp_epi<EPI_BIAS_BF16, 16, true>(g, grid, st);
      if (!g.f16 && g.epi == EPI_BIAS_RESID_H16) return launch_sp_epi<EPI_BIAS_RESID_H16, 16>(g, grid, st);
    }
  }
#endif
  if (g.f16) {
    switch (g.epi) {
      case EPI_BIAS_BF16: return launch_sp_epi<EPI_BIAS_BF16, 0, true>(g, grid, st);
      case EPI_BIAS_F16: return launch_sp_epi<EPI_BIAS_F16, 0, true>(g, grid, st);
      case EPI_BIAS_QGELU_BF16: return launch_sp_epi<EPI_BIAS_QGELU_BF16, 0, true>(g, grid, st);
      case EPI_BIAS_GELU_BF16: return launch_sp_epi<EPI_BIAS_GELU_BF16, 0, true>(g, grid, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_sp_epi<EPI_BIAS_BF16>(g, grid, st);
    case EPI_BIAS_F16: return launch_sp_epi<EPI_BIAS_F16>(g, grid, st);
    case EPI_BIAS_QGELU_BF16: return launch_sp_epi<EPI_BIAS_QGELU_BF16>(g, grid, st);
    case EPI_BIAS_GELU_BF16: return launch_sp_epi<EPI_BIAS_GELU_BF16>(g, grid, st);
    case EPI_BIAS_RESID_F32: return launch_sp_epi<EPI_BIAS_RESID_F32>(g, grid, st);
    case EPI_BIAS_RESID_H16: return launch_sp_epi<EPI_BIAS_RESID_H16>(g, grid, st);
    case EPI_TABLE_F32: return launch_sp_epi<EPI_TABLE_F32>(g, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
