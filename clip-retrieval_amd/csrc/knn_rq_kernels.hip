// knn_rq_kernels.hip -- the "register-stationary queries" (RQ) flat scan: up to 256 queries per pass over HBM.
//
// Stands in for the same faiss `IndexFlatIP.search` arithmetic as knn_kernels.hip (reference call sites
// clip_retrieval/clip_back.py:362, clip_filter.py:55), for LARGE query batches (B > 64: SURVEY config 3 asks for B = 256).
//
// Why a second scan kernel.  The scan of knn_kernels.hip keeps the queries as the stationary MFMA operand in LDS and
// streams index rows HBM -> VGPR.  128 queries x 768 x fp16 = 192 KiB do not fit the 160 KiB of LDS, so that design
// stops at 64 queries per pass (2 470 QPS at 100 M rows, 0.19 of the B = 256 ceiling).  The only on-chip store large
// enough for more queries is the register file (512 KiB per CU), and a register operand is private to its wave.  So the
// roles are swapped:
//   * every wave keeps ITS OWN 32 x QBW queries in registers for the whole kernel (fp16 "hi" parts as MFMA B fragments:
//     QBW x d/16 x 4 VGPRs = 384 at d = 768, QBW = 2; one wave per SIMD, 512 registers each);
//   * index rows go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4; no VGPRs, fragment-shaped so that the LDS image of a
//     k-step is lane-linear: ds_read_b128 at <slot> + 1 KiB * kstep + 16 * lane is conflict-free) into a ring of 32-row
//     tiles, and ALL waves of the workgroup read every tile: an X byte is fetched from HBM once and used by 4 waves x 64
//     queries.  One s_barrier per tile; the DMA of tile t+2 is issued while tile t is being multiplied.
//   * no LDS candidate queues (they were 80 B x 64 entries per query: another 160 KiB at 256 queries).  The scan is a
//     RANGE scan with one threshold per query: a lane owns one query column of the 32 x 32 score tile and compares its 16
//     scores with that query's threshold held in a register; hits (a few thousand per query per scan) are compacted by
//     wave ballot into a small wave-private LDS staging list and flushed to per-query global hit lists.
//   The thresholds come from a SAMPLE pass: the 64-query scan of knn_kernels.hip over every S-th tile (1/S of the bytes);
//   the threshold of a query is the J-th best approximate score of the sample, J = k + 8.  Sample rows are index rows, so
//   at least J >= k rows reach the threshold -- deterministically, whatever the data distribution -- and about J * S do.
//   Afterwards the hits are re-scored exactly (fp32 FMA, the arithmetic of knn_rescore_kernel), the top-k selected by
//   (score desc, id asc), and exactness is PROVEN per query: every row that is not a hit has approximate score < thr,
//   hence exact score < thr + eps; if the k-th exact score among the hits is >= thr + eps no outsider belongs to the
//   top-k.  A query whose proof fails (hit list overflow, or a k-th score within eps of the threshold) is re-run by the
//   exact 32-query scan, gated on the device like the fallback of the 64-query wide scan.
//
// HBM-bound as long as the matrix pipe keeps up: per 32-row tile (48 KiB at d = 768) a SIMD issues 96
// v_mfma_f32_32x32x16_f16 = 3 072 cycles; at ~6 TB/s a CU receives a tile every 2.05 us, i.e. the kernel stays on the
// HBM roof while the shader clock is >= 1.5 GHz.  Algorithmic bytes per launch: N * d * 2 (the same as the 64-query scan).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <float.h>
#include <algorithm>
#include "knn_kernels.h"

namespace knnx {

// KNNX_MFMA16 (default 1, round 4): the scan and the assignment kernel multiply with v_mfma_f32_16x16x32_f16 instead of
// v_mfma_f32_32x32x16_f16 -- on this power-managed part the 16 x 16 x 32 shape sustains more TFLOP/s (the same switch as
// the encoder's GEMMs, CLIPX_MFMA16; -DCLIPX_MFMA16=0 / -DKNNX_MFMA16=0 builds the previous kernels for the A/B).  What changes is
// the fragment shape, nothing else: a 1-KiB LDS piece is (16 rows x 32 columns) -- lane (r = l & 15, q4 = l >> 4) holds row r,
// columns 8 q4 .. + 8 of the 32-column slab -- instead of (32 rows x 16 columns); a 32-row tile is still d / 16 pieces, piece
// p = 2 * slab + (row half); a wave's queries are blocks of 16 (lane: query l & 15, the same columns); a 16 x 16 result block gives
// lane (query l & 15) the rows 4 q4 .. + 4 of its half.
#ifndef KNNX_MFMA16
#ifdef CLIPX_MFMA16
#define KNNX_MFMA16 CLIPX_MFMA16
#else
#define KNNX_MFMA16 1
#endif
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rq_enc_f(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float rq_dec_f(int e) { return __int_as_float(e >= 0 ? e : (e ^ 0x7fffffff)); }

// ---------------------------------------------------------------------------------------------
// prep: f32 queries -> fp16-hi B fragments in wave-block order; thresholds from the sample pass; counters reset
//   qfrag [nblk][KS][64 lanes][8 halves]: block b = queries 32b .. 32b+31; lane (n = l & 31, h = l >> 5) holds
//   q_n[16 s + 8 h + j].  thr[q] = J-th best sample score (samp [nq, kw] descending, -FLT_MAX padded; fewer than J sample
//   rows -> -inf: every row is a hit, which only happens on indexes far too small for this path); unused columns +inf.
// ---------------------------------------------------------------------------------------------
__global__ void knn_rq_prep_kernel(const float* __restrict__ q, int nq, int d, int nblk, _Float16* __restrict__ qfrag,
                                   const float* __restrict__ samp, int kw, int J, float slack, float* __restrict__ thr,
                                   unsigned* __restrict__ cnt, unsigned* __restrict__ lost) {
  const int s = blockIdx.x, b = blockIdx.y;  // k-step (KNNX_MFMA16: 32-column slab), query block (KNNX_MFMA16: of 16)
  const int lane = threadIdx.x;
#if KNNX_MFMA16
  // qfrag [nblk16][d / 32][64 lanes][8 halves]: lane (n = l & 15, q4 = l >> 4) holds q_n[32 s + 8 q4 + j]
  const int n = 16 * b + (lane & 15), h = lane >> 4;
  constexpr int QB = 16, KW = 32;
#else
  const int n = 32 * b + (lane & 31), h = lane >> 5;
  constexpr int QB = 32, KW = 16;
#endif
  half8 hi;
#pragma unroll
  for (int j = 0; j < 8; ++j) hi[j] = (n < nq) ? (_Float16)q[(size_t)n * d + KW * s + 8 * h + j] : (_Float16)0.f;
  reinterpret_cast<half8*>(qfrag)[((size_t)b * gridDim.x + s) * 64 + lane] = hi;
  if (s == 0 && lane < QB && thr) {  // (thr == null: fragments only -- the fp16 rest pass of a partial int8 copy brings its own thresholds)
    float t = INFINITY;
    if (n < nq) {
      const float v = samp[(size_t)n * kw + (J - 1)];
      // (KNNX_MFMA16: the sample scan accumulates 16 products per MFMA, this scan 32 -- the same exact products in another f32
      // summation order; a relative 1e-5 keeps the sample rows themselves above their own threshold)
      t = v > -FLT_MAX ? v - slack - (KNNX_MFMA16 ? 1e-5f * fabsf(v) : 0.f) : -INFINITY;
    }
    thr[n] = t;
    cnt[n] = 0u;
    lost[n] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// the scan
// ---------------------------------------------------------------------------------------------
constexpr int RQ_STAGE = 128;   // entries of a wave's private staging list
constexpr int RQ_FLUSH_AT = 48; // flush when at least this many are staged (a tile step adds a handful)

// LDS-DMA of tile `t` into ring slot `slot`: this wave's piece IDX (of DPW = KS / NW) -- wave w takes the DPW consecutive
// k-steps w DPW .. w DPW + DPW - 1 (4 consecutive k-steps are the same 128-B lines of the 32 rows), piece IDX = k-step
// w DPW + IDX.  Inline asm: hipcc must not know about the DMA, or it drains vmcnt(0) before every LDS
// read; s_nop 0: M0 needs a wait state before the DMA reads it.  Past the end of the index the last tile is re-loaded,
// which keeps the vmcnt arithmetic of the main loop uniform.
struct RqTile {
  const char* base;  // tile + this wave's offset
  unsigned m0b;      // LDS address of the slot + this wave's offset
  unsigned vo;       // per-lane byte offset (row, half)
  unsigned vo1;      // KNNX_MFMA16: the same for the pieces of the tile's second row half (rows 16 .. 31)
};
// per-lane source offsets of a fragment-shaped DMA: [0] ordinary tiles, [1] the (possibly ragged) last tile, whose rows >= N re-read
// row N - 1 (never admitted by the filter).  32x32x16: lane (row = l & 31, h = l >> 5) fetches 16 B of row `row` at column 8 h of
// the k-step.  KNNX_MFMA16: lane (r = l & 15, q4 = l >> 4) fetches 16 B of row 16 * half + r at column 8 q4 of the slab.
struct RqLaneOff {
  unsigned v[2], v1[2];
};
__device__ __forceinline__ RqLaneOff rq_lane_offsets(int lane, int D, int lrow /* last valid row inside the last tile */) {
  RqLaneOff o;
#if KNNX_MFMA16
  const int r = lane & 15, q4 = lane >> 4;
  o.v[0] = (unsigned)(r * D * 2 + q4 * 16);
  o.v1[0] = (unsigned)((16 + r) * D * 2 + q4 * 16);
  o.v[1] = (unsigned)((r < lrow ? r : lrow) * D * 2 + q4 * 16);
  o.v1[1] = (unsigned)((16 + r < lrow ? 16 + r : lrow) * D * 2 + q4 * 16);
#else
  const int row = lane & 31, hb = lane >> 5;
  o.v[0] = o.v1[0] = (unsigned)(row * D * 2 + hb * 16);
  o.v[1] = o.v1[1] = (unsigned)((row < lrow ? row : lrow) * D * 2 + hb * 16);
#endif
  return o;
}
template <int KS, int NW>
__device__ __forceinline__ RqTile rq_tile(const _Float16* __restrict__ X, int64_t t, int64_t ntile, int64_t last, const RqLaneOff& lo,
                                          unsigned lds_base, int slot, int w) {
  constexpr int TILE_BYTES = KS * 1024, DPW = KS / NW;
  const int64_t tt = t < ntile ? t : last;
  RqTile r;
  r.vo = tt == last ? lo.v[1] : lo.v[0];
  r.vo1 = tt == last ? lo.v1[1] : lo.v1[0];
  // the wave's DPW pieces: 32x32x16 -- k-steps w DPW .. (32 B each); KNNX_MFMA16 -- slabs w DPW / 2 .. (64 B each), both halves
  r.base = reinterpret_cast<const char*>(X) + (size_t)tt * TILE_BYTES + w * (DPW * 32);
  r.m0b = lds_base + slot * TILE_BYTES + w * (DPW * 1024);
  return r;
}
#define RQ_TILE(t_, slot_) rq_tile<KS, NW>(X, (t_), ntile, last, lane_off, lds_base, (slot_), w)
template <int NW, int IDX>
__device__ __forceinline__ void rq_issue_one(const RqTile& r) {
#if KNNX_MFMA16
  const char* p = r.base + (IDX >> 1) * 64;  // piece IDX of the wave = slab IDX >> 1, row half IDX & 1
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((IDX & 1) ? r.vo1 : r.vo), "s"(p), "s"(r.m0b),
               "n"(IDX * 1024)
               : "memory", "scc");
#else
  constexpr int KOFF = IDX;  // k-step minus the wave's first one
  const char* p = r.base + KOFF * 32;
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(r.vo), "s"(p), "s"(r.m0b), "n"(KOFF * 1024)
               : "memory", "scc");
#endif
}
template <int NW, int DPW, int IDX = 0>
__device__ __forceinline__ void rq_issue_all(const RqTile& r) {
  if constexpr (IDX < DPW) {
    rq_issue_one<NW, IDX>(r);
    rq_issue_all<NW, DPW, IDX + 1>(r);
  }
}

// ---- the k-loop of one tile, as a compile-time recursion over the k-step S (every LDS offset, wait count and DMA piece
// index must be a literal of an inline-asm statement).  A fragments (index rows) go through a 4-deep register ring, three
// k-steps ahead of the MFMAs that use them.  The reads are inline asm so that THIS file places the lgkmcnt waits (hipcc
// waits lgkmcnt(0) right behind each read, which puts the LDS latency of every k-step in front of its MFMAs).
template <int OFF>
__device__ __forceinline__ void rq_dsread(i32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void rq_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
#if KNNX_MFMA16
// piece S = 2 * slab + half: one LDS read feeds the 2 * QBW MFMAs of the wave's 16-query blocks against that half of the slab
#define RQ_NQB(QBW) (2 * (QBW))
#define RQ_NSL(KS) ((KS) / 2)
typedef float4v rq_acc_t[2];  // [row half] of one 16-query block
#else
#define RQ_NQB(QBW) (QBW)
#define RQ_NSL(KS) (KS)
typedef float16v rq_acc_t;
#endif
template <int KS, int QBW, int NW, int DPW, int S>
__device__ __forceinline__ void rq_ksteps(unsigned xa, i32x4 (&A)[4], rq_acc_t (&acc)[RQ_NQB(QBW)],
                                          const half8 (&Q)[RQ_NQB(QBW)][RQ_NSL(KS)], const RqTile& refill) {
  if constexpr (S < KS) {
    if constexpr (S + 3 < KS) rq_dsread<(S + 3) * 1024>(A[(S + 3) & 3], xa);
    rq_wait_lgkm<(KS - 1 - S < 3 ? KS - 1 - S : 3)>();  // reads issued after the one this step needs may stay in flight
    __builtin_amdgcn_sched_barrier(0);
#if KNNX_MFMA16
#pragma unroll
    for (int b = 0; b < 2 * QBW; ++b)
      acc[b][S & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, A[S & 3]), Q[b][S >> 1], acc[b][S & 1], 0, 0, 0);
#else
#pragma unroll
    for (int b = 0; b < QBW; ++b)
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A[S & 3]), Q[b][S], acc[b], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // DMA piece S / (KS / DPW) of the refill behind this step's MFMAs: the DPW pieces are spread over the tile so that
    // their issue time overlaps with MFMAs already queued instead of holding all waves at the top of the tile
    if constexpr (S % (KS / DPW) == 1) {
      rq_issue_one<NW, S / (KS / DPW)>(refill);
      __builtin_amdgcn_sched_barrier(0);
    }
    rq_ksteps<KS, QBW, NW, DPW, S + 1>(xa, A, acc, Q, refill);
  }
}

// KS = d / 16 k-steps; QBW = 32-query blocks per wave (1 or 2); NW = waves per workgroup (one per SIMD); NSLOT = ring slots
template <int KS, int QBW, int NW, int NSLOT>
__global__ __launch_bounds__(NW * 64, NW / 4) void knn_rq_scan_kernel(
    const _Float16* __restrict__ X, int64_t N, const _Float16* __restrict__ qfrag, const float* __restrict__ thr,
    unsigned* __restrict__ g_cnt, unsigned cap, float* __restrict__ hit_s, uint32_t* __restrict__ hit_r,
    unsigned* __restrict__ g_lost, const unsigned* __restrict__ gate, uint32_t row_off /* added to the row of a hit: X may be a slice */) {
  constexpr int D = KS * 16;
  constexpr int TILE_BYTES = KS * 1024;        // 32 rows x D x 2 B, stored as KS lane-linear 1 KiB k-step blocks
  constexpr int DPW = KS / NW;                 // DMA instructions per wave per tile: NW waves x DPW consecutive k-steps
  static_assert(KS % NW == 0, "k-steps must divide evenly among the waves");
  if (gate && *gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS map: NSLOT tiles, then per-wave staging lists {score, row, query}[RQ_STAGE]
  unsigned char* stage = smem + NSLOT * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // query blocks of QBS queries, NQB per wave, NSL fragments each; lane = (query column qcol, hb) -- hb is the row-group index of
  // the lane inside a result block: 32x32x16: rows 4 hb + 8 i + e (hb < 2), KNNX_MFMA16: rows 4 hb + e of a 16-row half (hb < 4)
  constexpr int NQB = RQ_NQB(QBW), NSL = RQ_NSL(KS), QBS = KNNX_MFMA16 ? 16 : 32;
  const int qcol = lane & (QBS - 1), hb = lane / QBS;
  float* st_s = reinterpret_cast<float*>(stage + w * RQ_STAGE * 12);
  uint32_t* st_r = reinterpret_cast<uint32_t*>(st_s + RQ_STAGE);
  uint32_t* st_q = st_r + RQ_STAGE;

  // ---- this wave's queries: B fragments for the whole kernel, and the thresholds of this lane's query columns
  half8 Q[NQB][NSL];
  float tq[NQB];
#pragma unroll
  for (int b = 0; b < NQB; ++b) {
    const int blk = w * NQB + b;
    const half8* src = reinterpret_cast<const half8*>(qfrag) + (size_t)blk * NSL * 64 + lane;
#pragma unroll
    for (int s = 0; s < NSL; ++s) Q[b][s] = src[s * 64];
    tq[b] = thr[blk * QBS + qcol];
  }
  // every fragment is "used" here once, so hipcc waits for these loads HERE: otherwise it counts them down at their first
  // uses inside the tile loop, and the last of those waits (vmcnt(0)) would drain the DMA ring on every tile
#pragma unroll
  for (int b = 0; b < NQB; ++b) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      // the second 32 queries live in accumulation registers for good (MFMA reads B operands from either file; without the
      // constraint hipcc "spills" them to AGPRs and copies 4 registers back before every MFMA)
      if (b < NQB / QBW) asm volatile("" : "+v"(Q[b][s]));
      else asm volatile("" : "+a"(Q[b][s]));
    }
    asm volatile("" : "+v"(tq[b]));
  }

  const int64_t ntile = (N + 31) >> 5;
  const int64_t last = ntile - 1;
  // lane offsets of the fragment-shaped DMAs (rq_lane_offsets); the piece and the tile are added to the SGPR base
  const RqLaneOff lane_off = rq_lane_offsets(lane, D, (int)(N - 1 - last * 32));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);

  // (helpers above instead of a lambda: a lambda capturing by reference makes hipcc keep the closure in scratch memory;
  // RQ_TILE is defined at file level, below rq_tile)

  int64_t t = blockIdx.x;
  const int64_t gstride = gridDim.x;
  // prologue: NSLOT - 1 tiles in flight
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) rq_issue_all<NW, DPW>(RQ_TILE(t + (int64_t)i * gstride, i));

  int nst = 0;  // entries in this wave's staging list (wave-uniform)
  int slot = 0;
  for (; t < ntile; t += gstride) {
    // tile t has landed (this wave's share: all but the DPW * (NSLOT - 2) youngest DMAs are done), then everyone's
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (NSLOT - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // refill of the slot that tile t - 1 (of this workgroup's sequence) occupied -- every wave is past it.  The DPW DMA
    // instructions are spread over the k-loop below, one every KS / DPW k-steps, so that their issue time (tens of
    // cycles each) overlaps with MFMAs already queued instead of holding all waves at the top of the tile.
    const RqTile refill = RQ_TILE(t + (int64_t)(NSLOT - 1) * gstride, slot == 0 ? NSLOT - 1 : slot - 1);

    rq_acc_t acc[NQB];
#if KNNX_MFMA16
#pragma unroll
    for (int b = 0; b < NQB; ++b) acc[b][0] = acc[b][1] = float4v{0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
    for (int b = 0; b < QBW; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    }
#endif
    // the k-loop: rq_ksteps<...> above (A fragments through a 4-deep register ring, the refill DMAs spread over the steps)
    const unsigned xa = lds_base + slot * TILE_BYTES + lane * 16;
    i32x4 A[4];
    rq_dsread<0>(A[0], xa);
    rq_dsread<1024>(A[1], xa);
    rq_dsread<2048>(A[2], xa);
    __builtin_amdgcn_sched_barrier(0);
    rq_ksteps<KS, QBW, NW, DPW, 0>(xa, A, acc, Q, refill);

    // ---- filter: lane (qcol, hb) owns rows row0 + (r & 3) + 8 (r >> 2) of its query column in each block
    const int64_t row0 = t * 32 + 4 * hb;
    // any hit in this wave?  One v_max3 tree per block and ONE compare instead of 16 compares + 16 mask ORs: this stretch
    // runs with the matrix pipe idle.  (hipcc keeps the accumulators in AGPRs -- forcing them into VGPRs only adds copies --
    // so every value still costs one v_accvgpr_read.)
    bool any = false;
#if KNNX_MFMA16
    // RQ_SCORE(b, r): the lane's r-th score of block b, r = 4 * half + e -> row row0 + 16 * half + e
#define RQ_SCORE(b, r) acc[b][(r) >> 2][(r) & 3]
#define RQ_ROWOFF(r) (16 * ((r) >> 2) + ((r) & 3))
    constexpr int NSC = 8;
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
      const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[b][0][0], acc[b][0][1]), acc[b][0][2]);
      const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[b][0][3], acc[b][1][0]), acc[b][1][1]);
      const float m2 = __builtin_fmaxf(acc[b][1][2], acc[b][1][3]);
      any |= __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2) >= tq[b];
    }
#else
#define RQ_SCORE(b, r) acc[b][r]
#define RQ_ROWOFF(r) (((r) & 3) + 8 * ((r) >> 2))
    constexpr int NSC = 16;
#pragma unroll
    for (int b = 0; b < QBW; ++b) {
      float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[b][0], acc[b][1]), acc[b][2]);
      float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[b][3], acc[b][4]), acc[b][5]);
      float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[b][6], acc[b][7]), acc[b][8]);
      float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[b][9], acc[b][10]), acc[b][11]);
      float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[b][12], acc[b][13]), acc[b][14]);
      m0 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
      m3 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[b][15]);
      any |= __builtin_fmaxf(m0, m3) >= tq[b];
    }
#endif
    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // a hit somewhere in the wave: ~1 tile step in 8
      bool vmem = false;
#pragma unroll
      for (int b = 0; b < NQB; ++b) {
#pragma unroll
        for (int r = 0; r < NSC; ++r) {
          const int64_t row = row0 + RQ_ROWOFF(r);
          const bool hit = RQ_SCORE(b, r) >= tq[b] && row < N;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m != 0ull) {
            const int pos = nst + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned qq = (unsigned)((w * NQB + b) * QBS + qcol);
            if (hit) {
              if (pos < RQ_STAGE) {
                st_s[pos] = RQ_SCORE(b, r);
                st_r[pos] = (uint32_t)row + row_off;
                st_q[pos] = qq;
              } else {
                g_lost[qq] = 1u;  // staging overflow (flood of hits in one tile step): the query falls back
                vmem = true;
              }
            }
            nst += __builtin_popcountll(m);
          }
        }
      }
      if (nst > RQ_STAGE) nst = RQ_STAGE;
      if (nst >= RQ_FLUSH_AT) {
        vmem = true;
        for (int i = lane; i < nst; i += 64) {
          const unsigned qq = st_q[i];
          const unsigned pos = atomicAdd(&g_cnt[qq], 1u);
          if (pos < cap) {
            hit_s[(size_t)qq * cap + pos] = st_s[i];
            hit_r[(size_t)qq * cap + pos] = st_r[i];
          }
        }
        nst = 0;
      }
      // a flush (or a lost flag) issued VMEM operations that hipcc counts itself: drain, so that the DMA arithmetic at the
      // loop top holds.  Rare (once per ~RQ_FLUSH_AT hits); a hit that is only staged in LDS touches no VMEM counter.
      if (__builtin_amdgcn_ballot_w64(vmem) != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  // drain the DMAs still in flight (tail reloads) before the LDS is released, then the last staged hits
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < nst; i += 64) {
    const unsigned qq = st_q[i];
    const unsigned pos = atomicAdd(&g_cnt[qq], 1u);
    if (pos < cap) {
      hit_s[(size_t)qq * cap + pos] = st_s[i];
      hit_r[(size_t)qq * cap + pos] = st_r[i];
    }
  }
}

// =============================================================================================
// int8 first stage (round 4).  The flat index keeps, next to its fp16 rows, an int8 copy x8 = rint(x / c) with one scale per
// COLUMN (c_j = max_i |x_ij| / 127), and the scan of a query batch reads THAT: half the bytes of a pass over HBM, and
// v_mfma_i32_16x16x64_i8 at twice the fp16 rate.  Its scores are approximations with a PROVEN error bound, used only to decide
// which rows are re-scored exactly from the fp16 rows (knn_rq_rescore_kernel) -- the result is the exact top-k, as before:
//   u = q * c (per column), u8 = rint(u / s_u), s_u = max |u| / 127;   approx(q, i) = s_u * sum_j u8_j x8_ij   (int32, exact)
//   exact - approx = sum_j u_j (y_ij - x8_ij) + sum_j (u_j - s_u u8_j) x8_ij,   y = x / c
//   |exact - approx| <= |u| * A + |u - s_u u8| * B =: eps8(q),   A = max_i |y_i - x8_i|, B = max_i |x8_i|   (2-norms; A, B are
//   maxima over the rows actually stored, computed by the quantisation kernel)
// Threshold: T(q) = (J-th best score of the exact sample pass, J >= k) - eps_hi(q): at least J rows have exact score >= T, so
// the top-k all have, and every row with exact >= T has approx >= T - eps8, i.e. int32 sum >= (T - eps8) / s_u: ONE integer
// compare per score in the scan.  Rows that pass go to the per-query hit lists; if a list overflows the query falls back to
// the exact scan (as in the fp16 register-stationary path, whose sample pass, hit lists, re-scoring and merge this path shares).
// How many rows pass depends on the data: for unit vectors with components of similar size (the bench's synthetic index)
// eps8 ~ 0.02 |q|, ~0.5 sigma of the score distribution -- a few times 10^4 hits per query at 10^8 rows; embeddings with a few
// dominant columns would quantise the QUERY coarsely (one scale for all of u) and pass more rows, up to the fallback: those indexes
// get the dominant-column form (I8Dom, round 5: the dominant components as 14-bit integers outside the MFMA product) or, with more
// than four such columns, two int8 planes per query (round 4) -- knnx_api.hip i8_ensure decides at every full build of the copy.
// Geometry: an int8 row of d bytes is an fp16 row of d / 2 columns to the tile / DMA helpers above: KS = d / 32 pieces of
// (16 rows x 64 bytes) per 32-row tile, lane (r = l & 15, q4 = l >> 4) holds bytes 16 q4 .. + 16 of the piece's 64.
// =============================================================================================
#if KNNX_MFMA16
typedef int i32x4v __attribute__((ext_vector_type(4)));

// column maxima: colmax_enc[j] = max_i |x_ij| as the bits of a non-negative float (integer order = float order)
__global__ __launch_bounds__(256) void knn_i8_colmax_kernel(const _Float16* __restrict__ X, int64_t N, int d, int* __restrict__ colmax_enc) {
  // a block walks rows blockIdx.x, + gridDim.x, ...; thread t owns columns 4 t .. 4 t + 3 (one 8-byte load per row; d <= 1024)
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  const int c = 4 * threadIdx.x;
  if (c < d) {
    for (int64_t r = blockIdx.x; r < N; r += gridDim.x) {
      const uint2 raw = *reinterpret_cast<const uint2*>(X + (size_t)r * d + c);
      const _Float16* h = reinterpret_cast<const _Float16*>(&raw);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], fabsf((float)h[e]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMax(&colmax_enc[c + e], __float_as_int(m[e]));
  }
}
__global__ void knn_i8_colscale_kernel(const int* __restrict__ colmax_enc, int d, float* __restrict__ colscale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < d) {
    const float m = __int_as_float(colmax_enc[c]);
    colscale[c] = m > 0.f ? m / 127.f : 1.f;
  }
}
// Layout of the int8 copy (round 5): NOT row-major.  X8 is the LDS image of the scan, tile after tile: tile t (rows 32 t .. + 32) is
// d / 32 pieces of 1 KiB, piece p = 2 * slab + half, and inside a piece lane l = 16 * q4 + r holds the 16 bytes (columns 64 slab +
// 16 q4 .. + 16) of row 32 t + 16 half + r.  One global_load_lds_dwordx4 of the scan then moves ONE CONTIGUOUS KiB (eight full 128-B
// lines) instead of 16 rows x 64 B -- half a line per request, which held the row-major pass at 0.61 of HBM.  The copy
// is private to this library (rows are re-scored and reconstructed from the fp16 rows), so its layout is free.
// Quantisation: a wave takes one 16-row half tile at a time (grid-stride) and walks its slabs: lane (r, q4) reads its 32 B of fp16
// (four lanes = one full line of the row), writes its 16 B of the piece (the wave = the contiguous KiB), and keeps the row's
// partial sums of A^2 = |y - x8|^2 and B^2 = |x8|^2; rows at or beyond N are written as zeros (the scan's filter never admits
// them).  Half tiles [h0, h1) are (re)written: appended rows re-quantise the rows that share their first half tile with the same
// scales, i.e. to the same bytes.  ONE atomic pair per wave at the end for A = max |y - x8|, B = max |x8| (2-norms per row).
__global__ __launch_bounds__(256) void knn_i8_quant_kernel(const _Float16* __restrict__ X, int64_t N, int64_t h0, int64_t h1, int d,
                                                          const float* __restrict__ colscale, const I8Dom dom,
                                                          int8_t* __restrict__ X8, int* __restrict__ ab_enc) {
  // dom (knn_kernels.h): position p of the image holds column p, except the dom.nfix positions dom.pos[f], which hold column dom.src[f]
  // (dominant columns swapped into positions 0 .. dom.n - 1); A and B are norms over all positions, i.e. over all columns
  const int lane = threadIdx.x & 63, r = lane & 15, q4 = lane >> 4;
  const int nsl = d >> 6;
  float ma = 0.f, mb = 0.f;
  for (int64_t h = h0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); h < h1; h += (int64_t)gridDim.x * 4) {
    const int64_t row = h * 16 + r;
    const bool live = row < N;
    const _Float16* xrow = X + (size_t)(live ? row : 0) * d;
    const _Float16* src = xrow + 16 * q4;
    int8_t* dst = X8 + (size_t)(h >> 1) * 32 * d + (size_t)(h & 1) * 1024 + lane * 16;
    float ea = 0.f, eb = 0.f;
    for (int sl = 0; sl < nsl; ++sl) {
      uint4 raw[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
      if (live) {
        raw[0] = *reinterpret_cast<const uint4*>(src + 64 * sl);
        raw[1] = *reinterpret_cast<const uint4*>(src + 64 * sl + 8);
      }
      const _Float16* hv = reinterpret_cast<const _Float16*>(raw);
      const float4* cs4 = reinterpret_cast<const float4*>(colscale + 64 * sl + 16 * q4);
      float xv[16], sc[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 cs = cs4[g];
        sc[4 * g] = cs.x; sc[4 * g + 1] = cs.y; sc[4 * g + 2] = cs.z; sc[4 * g + 3] = cs.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[4 * g + e] = (float)hv[4 * g + e];
      }
      for (int f = 0; f < dom.nfix; ++f) {
        if ((dom.pos[f] >> 4) == 4 * sl + q4) {
          const int e0 = dom.pos[f] & 15;
          const float xs = live ? (float)xrow[dom.src[f]] : 0.f;
          const float ss = colscale[dom.src[f]];
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (e == e0) { xv[e] = xs; sc[e] = ss; }
        }
      }
      unsigned packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = xv[4 * g + e] / sc[4 * g + e];
          float v = rintf(y);
          v = fminf(fmaxf(v, -127.f), 127.f);
          ea += (y - v) * (y - v);
          eb += v * v;
          packed[g] |= ((unsigned)(int)v & 0xffu) << (8 * e);
        }
      }
      *reinterpret_cast<uint4*>(dst + (size_t)sl * 2048) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
    // the four q4 lanes of a row hold its partial sums
    ea += __shfl_xor(ea, 16); ea += __shfl_xor(ea, 32);
    eb += __shfl_xor(eb, 16); eb += __shfl_xor(eb, 32);
    ma = fmaxf(ma, ea);
    mb = fmaxf(mb, eb);
  }
  for (int o = 8; o > 0; o >>= 1) {
    ma = fmaxf(ma, __shfl_xor(ma, o));
    mb = fmaxf(mb, __shfl_xor(mb, o));
  }
  if (lane == 0) {
    // one ulp above the rounded-to-nearest root (bit pattern + 1: the values are non-negative and finite), so that the stored maximum
    // is never below the true root; the fp32 summation error of ma / mb themselves is covered by the 1 + 1e-3 of knn_i8_prep_kernel
    atomicMax(&ab_enc[0], __float_as_int(sqrtf(ma)) + 1);
    atomicMax(&ab_enc[1], __float_as_int(sqrtf(mb)) + 1);
  }
}

// queries: one wave per query slot (grid = queries per pass).  Writes the int8 B fragments (qfrag8 [nblk16][d / 64][64 lanes][16 B]:
// lane (n = l & 15, q4 = l >> 4) holds u8_n[64 s + 16 q4 .. + 16]), the integer admission threshold thr_i and the exact-score
// lower bound thr_lb (= T) of the proof; resets the hit counters.  Unused slots: zero fragments, thr_i = INT_MAX.
__global__ __launch_bounds__(64) void knn_i8_prep_kernel(const float* __restrict__ q, int nq, int d, const float* __restrict__ colscale,
                                                        const I8Dom dom, const int* __restrict__ ab_enc, const int* __restrict__ maxnorm,
                                                        const float* __restrict__ samp, int kw, int J, int planes, int refine,
                                                        int8_t* __restrict__ qfrag8, int8_t* __restrict__ qdom, int* __restrict__ thr_i,
                                                        float* __restrict__ thr_lb,
                                                        float* __restrict__ thr_rest, unsigned* __restrict__ cnt,
                                                        unsigned* __restrict__ lost) {
  // thr_rest (may be null): the admission threshold of the fp16 register-stationary pass over the rows the int8 copy does not hold
  // (a partial copy, knnx_api.hip): that pass compares fp16-hi approximations, so T - eps_hi(q) admits every row with exact >= T.
  // refine = 1 (second call of a two-level sample, knnx_api.hip scan_topk_i8): samp holds EXACT scores (re-scored hits of an int8 pass
  // over every 32nd tile) -- no eps_hi -- and the bound only ever rises: T = max(previous T, J-th best); fewer than J sample hits
  // leave the previous threshold in place.
  // planes = 2: the query is TWO int8 planes, u ~ s_u u8 + (s_u / 128) u8b (the second quantises what the first left), and the scan
  // compares 128 * sum(u8 x8) + sum(u8b x8) -- for indexes whose columns differ widely in size (CLIP embeddings have a few dominant
  // dimensions): one scale for all of u then leaves |u - s_u u8| * B as the whole error (1.25 sigma of the scores in a simulation with
  // two columns 5 x the rest, 3 x 10^5 admitted rows per query at 10^8 rows; two planes: 0.26 sigma, 9 x 10^3).  Second fragment set
  // at qfrag8 + 256 * d.
  // dom.n > 0 (one plane): positions 0 .. dom.n - 1 of the image hold the index's dominant columns (I8Dom).  The query's components there
  // do not take part in the scale: s_u = max over the OTHER positions / 127, their fragment bytes are zero, and each is quantised to a
  // 14-bit integer t = 128 hi + lo (|hi| <= 127, |lo| <= 64) with the same s_u -- qdom[n][4] = lo, qdom[1024 + n][4] = hi, which the scan
  // multiplies with bytes 0..3 of each row on the vector ALU.  The bound is unchanged: u8 is the integer vector (t_0 .. t_{n-1}, u8 of the
  // rest), res = u - s_u u8 over ALL positions, |u|, A, B as before -- only the integers at the dominant positions have 14 bits
  // instead of 7, so a dominant component costs the error of ITS rounding (s_u / 2) instead of setting s_u for everything else.
  const int n = blockIdx.x, lane = threadIdx.x;
  const int nsl = d / 64;
  int8_t* dst = qfrag8 + ((size_t)(n >> 4) * nsl * 64 + (n & 15)) * 16;  // + (s * 64 + q4 * 16) * 16 + byte
  int8_t* dst2 = dst + (size_t)256 * d;
  float u[16];  // columns lane + 64 e
  float mu = 0.f, mud = 0.f, e2 = 0.f, n2 = 0.f;  // mud: the largest dominant component
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = lane + 64 * e;
    u[e] = 0.f;
    if (c < d && n < nq) {
      int sc = c;
      for (int f = 0; f < dom.nfix; ++f)
        if (dom.pos[f] == c) sc = dom.src[f];
      const float v = q[(size_t)n * d + sc];
      const float rr = v - (float)(_Float16)v;
      e2 += rr * rr;
      n2 += v * v;
      u[e] = v * colscale[sc];
      if (c >= dom.n) mu = fmaxf(mu, fabsf(u[e]));
      else mud = fmaxf(mud, fabsf(u[e]));
    }
  }
  float nu2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) nu2 += u[e] * u[e];
  for (int o = 32; o > 0; o >>= 1) {
    mu = fmaxf(mu, __shfl_xor(mu, o));
    mud = fmaxf(mud, __shfl_xor(mud, o));
    e2 += __shfl_xor(e2, o);
    n2 += __shfl_xor(n2, o);
    nu2 += __shfl_xor(nu2, o);
  }
  // a query that is zero outside the dominant columns takes its scale from THEM (their 14-bit range): with s_u = 1 the digits would
  // round to zero, eps8 would admit every row and the query would fall back to the exact scan
  const float su = mu > 0.f ? mu / 127.f : (mud > 0.f ? mud / 16256.f : 1.f);
  float er2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int c = lane + 64 * e;
    if (c < d) {
      float v = rintf(u[e] / su);
      // position c = 64 s + 16 q4 + byte
      const size_t fo = (size_t)((c >> 6) * 64 + ((c >> 4) & 3) * 16) * 16 + (c & 15);
      if (c < dom.n) {  // (e = 0, lane < dom.n)
        v = fminf(fmaxf(v, -16256.f), 16256.f);
        const float hi = rintf(v * (1.f / 128.f));
        qdom[n * 4 + c] = (int8_t)(int)(v - 128.f * hi);
        qdom[1024 + n * 4 + c] = (int8_t)(int)hi;
        const float resd = u[e] - su * v;
        er2 += resd * resd;
        dst[fo] = 0;
        continue;
      }
      v = fminf(fmaxf(v, -127.f), 127.f);
      float res = u[e] - su * v;
      dst[fo] = (int8_t)(int)v;
      if (planes == 2) {
        const float su2 = su * (1.f / 128.f);
        float v2 = rintf(res / su2);
        v2 = fminf(fmaxf(v2, -127.f), 127.f);
        res -= su2 * v2;
        dst2[fo] = (int8_t)(int)v2;
      }
      er2 += res * res;
    }
  }
  for (int o = 32; o > 0; o >>= 1) er2 += __shfl_xor(er2, o);
  if (qdom && lane < 4 && (lane >= dom.n || n >= nq)) qdom[n * 4 + lane] = qdom[1024 + n * 4 + lane] = 0;
  if (lane == 0) {
    int ti = 0x7fffffff;
    float lb = INFINITY;
    bool keep = false;
    if (n < nq) {
      const float v = samp[(size_t)n * kw + (J - 1)];
      if (refine && !(v > -FLT_MAX)) {
        keep = true;
      } else if (v > -FLT_MAX) {
        // eps_hi: the sample scores are fp16-hi approximations of the exact scores (cf. knn_rq_proof_kernel)
        // Every norm below is an fp32 sum of up to d non-negative terms followed by a square root: relative error <= (d + 2) 2^-24
        // < 6.2e-5 at d = 1024, and the products / sums of the bound add a few ulp.  The factor 1 + 1e-3 covers all of it with an
        // order of magnitude to spare (it widens the admission band by 0.1 %); the same factor is in oracle.knn_oracle.Int8FirstStage.
        const float eps_q = (sqrtf(e2) + (float)d * 1.2e-7f * sqrtf(n2)) * rq_dec_f(*maxnorm) * 1.001f;
        const float eps_hi = refine ? 0.f : eps_q;
        lb = v - eps_hi;
        if (refine) lb = fmaxf(lb, thr_lb[n]);
        if (thr_rest) thr_rest[n] = lb - eps_q - 1e-6f * fabsf(lb);
        const float A = __int_as_float(ab_enc[0]), B = __int_as_float(ab_enc[1]);
        const float eps8 = (sqrtf(nu2) * A + sqrtf(er2) * B) * 1.001f + 1e-6f * fabsf(lb);
        const float t = (lb - eps8) / (planes == 2 ? su * (1.f / 128.f) : su);
        // floor - 1: the float division / subtraction above may round up by an ulp
        ti = t <= -2.0e9f ? (int)0x80000000 : (t >= 2.0e9f ? 0x7fffffff : (int)floorf(t) - 1);
      } else {
        lb = -INFINITY;  // fewer than J sample rows: every row is a hit (indexes far too small for this path)
        ti = (int)0x80000000;
        if (thr_rest) thr_rest[n] = -INFINITY;
      }
    } else if (thr_rest) {
      thr_rest[n] = INFINITY;  // unused query slot
    }
    if (!keep) {
      thr_i[n] = ti;
      thr_lb[n] = lb;
    }
    cnt[n] = 0u;
    lost[n] = 0u;
  }
}

// DMA of the int8 tiles: wave w fetches pieces w, w + NW, w + 2 NW, ... of the tile -- each one contiguous KiB of the tile-ordered
// copy (knn_i8_quant_kernel), lane l at byte 16 l, landing lane-linear at the same piece index of the ring slot.
struct Rq8Tile {
  const char* base;  // tile + this wave's first piece
  unsigned m0b;      // LDS address of the slot + this wave's first piece
};
// (t counts the tiles the pass VISITS: tile t * tstep of the index -- tstep > 1 is the sample pass of a two-level threshold; nj = visited
// tiles; past the end the last visited tile is re-loaded, which keeps the vmcnt arithmetic of the main loop uniform)
template <int KS, int NW>
__device__ __forceinline__ Rq8Tile rq8_tile(const int8_t* __restrict__ X8, int64_t t, int64_t nj, int tstep, unsigned lds_base, int slot, int w) {
  constexpr int TILE_BYTES = KS * 1024;
  const int64_t tt = (t < nj ? t : nj - 1) * tstep;
  Rq8Tile r;
  r.base = reinterpret_cast<const char*>(X8) + (size_t)tt * TILE_BYTES + w * 1024;
  r.m0b = lds_base + slot * TILE_BYTES + w * 1024;
  return r;
}
#define RQ8_TILE(t_, slot_) rq8_tile<KS, NW>(X8, (t_), nj, tstep, lds_base, (slot_), w)
template <int NW, int IDX>
__device__ __forceinline__ void rq8_issue_one(const Rq8Tile& r, unsigned vo /* 16 * lane */) {
  const char* p = r.base + IDX * NW * 1024;
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(p), "s"(r.m0b), "n"(IDX * NW * 1024)
               : "memory", "scc");
}
template <int NW, int DPW, int IDX = 0>
__device__ __forceinline__ void rq8_issue_all(const Rq8Tile& r, unsigned vo) {
  if constexpr (IDX < DPW) {
    rq8_issue_one<NW, IDX>(r, vo);
    rq8_issue_all<NW, DPW, IDX + 1>(r, vo);
  }
}

template <int KS, int NW, int DPW, int PL, int S>
__device__ __forceinline__ void rq8_ksteps(unsigned xa, i32x4 (&A)[4], i32x4v (&acc)[PL][2][2], const i32x4 (&Q)[PL][2][KS / 2],
                                           const Rq8Tile& refill, unsigned vo) {
  if constexpr (S < KS) {
    if constexpr (S + 3 < KS) rq_dsread<(S + 3) * 1024>(A[(S + 3) & 3], xa);
    rq_wait_lgkm<(KS - 1 - S < 3 ? KS - 1 - S : 3)>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[p][b][S & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[S & 3], Q[p][b][S >> 1], acc[p][b][S & 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S % (KS / DPW) == 1) {
      rq8_issue_one<NW, S / (KS / DPW)>(refill, vo);
      __builtin_amdgcn_sched_barrier(0);
    }
    rq8_ksteps<KS, NW, DPW, PL, S + 1>(xa, A, acc, Q, refill, vo);
  }
}

#ifdef CLIPX_ABLATE
// tools build, KNNX_RQ8_TIMER=1: shader cycles per wave in [0] the wait + barrier at the top of a tile, [1] the k-loop (fragment reads,
// MFMAs, refill DMAs), [2] the filter, [3] tiles  -> g_rq8_phase[(workgroup * 8 + wave) * 4 + i]; read back with knnx_dbg_rq8_phases()
// (tools/rq8_phases.py; profiles/r05o_rq8_phases*.log)
__device__ long long g_rq8_phase[256 * 8 * 4];
__device__ int g_rq8_timer = 0;
#define RQ8_STAMP(i) if (timer) { const long long n_ = (long long)__builtin_readcyclecounter(); tph[i] += n_ - tst; tst = n_; }
#else
#define RQ8_STAMP(i)
#endif
// KS = d / 32 pieces per 32-row tile; 8 waves x 32 queries (two blocks of 16); the structure of knn_rq_scan_kernel
// PL = 2: two query planes (knn_i8_prep_kernel), score = 128 * plane 0 + plane 1
// DOM: the index has dominant columns (I8Dom) -- bytes 0..3 of each row, zero in the query fragments; after the k-loop a lane reads those
// four bytes of its eight rows from the tile in LDS (one ds_read_b32 each) and adds dot4(x, lo) + 128 dot4(x, hi) with its queries'
// 14-bit digits (v_dot4c_i32_i8: two per score) to the MFMA sums -- the filter below then sees the full integer score
template <int KS, int NW, int NSLOT, int PL, bool DOM>
__global__ __launch_bounds__(NW * 64, NW / 4) void knn_rq8_scan_kernel(const int8_t* __restrict__ X8, int64_t N, const int8_t* __restrict__ qfrag8,
                                                                      const int8_t* __restrict__ qdom, const int* __restrict__ thr_i, unsigned* __restrict__ g_cnt, unsigned cap,
                                                                      float* __restrict__ hit_s, uint32_t* __restrict__ hit_r,
                                                                      unsigned* __restrict__ g_lost, int tstep) {
  constexpr int TILE_BYTES = KS * 1024, DPW = KS / NW, NSL = KS / 2;
  static_assert(KS % NW == 0 && KS % 2 == 0, "pieces must divide evenly among the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* stage = smem + NSLOT * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qcol = lane & 15, hb = lane >> 4;
  float* st_s = reinterpret_cast<float*>(stage + w * RQ_STAGE * 12);
  uint32_t* st_r = reinterpret_cast<uint32_t*>(st_s + RQ_STAGE);
  uint32_t* st_q = st_r + RQ_STAGE;

  i32x4 Q[PL][2][NSL];
  int tq[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int blk = w * 2 + b;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const i32x4* src = reinterpret_cast<const i32x4*>(qfrag8 + (size_t)p * 256 * (KS * 32)) + (size_t)blk * NSL * 64 + lane;
#pragma unroll
      for (int s = 0; s < NSL; ++s) Q[p][b][s] = src[s * 64];
    }
    tq[b] = thr_i[blk * 16 + qcol];
  }
  int dlo[2] = {0, 0}, dhi[2] = {0, 0};
  if constexpr (DOM) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      dlo[b] = reinterpret_cast<const int*>(qdom)[(w * 2 + b) * 16 + qcol];
      dhi[b] = reinterpret_cast<const int*>(qdom)[256 + (w * 2 + b) * 16 + qcol];
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int s = 0; s < NSL; ++s) asm volatile("" : "+v"(Q[p][b][s]));  // all landed before the DMA ring starts (see knn_rq_scan_kernel)
    asm volatile("" : "+v"(tq[b]));
    if constexpr (DOM) asm volatile("" : "+v"(dlo[b]), "+v"(dhi[b]));
  }

  const int64_t ntile = (N + 31) >> 5;
  const int64_t nj = (ntile + tstep - 1) / tstep;  // tiles this pass visits
  const unsigned vo = (unsigned)lane * 16u;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);

  int64_t t = blockIdx.x;
  const int64_t gstride = gridDim.x;
#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) rq8_issue_all<NW, DPW>(RQ8_TILE(t + (int64_t)i * gstride, i), vo);

  int nst = 0;
  int slot = 0;
#ifdef CLIPX_ABLATE
  const bool timer = g_rq8_timer != 0;
  long long tph[4] = {0, 0, 0, 0};
  long long tst = timer ? (long long)__builtin_readcyclecounter() : 0;
#endif
  for (; t < nj; t += gstride) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (NSLOT - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    RQ8_STAMP(0)
    const Rq8Tile refill = RQ8_TILE(t + (int64_t)(NSLOT - 1) * gstride, slot == 0 ? NSLOT - 1 : slot - 1);
    i32x4v acc[PL][2][2];
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[p][b][0] = acc[p][b][1] = i32x4v{0, 0, 0, 0};
    const unsigned xa = lds_base + slot * TILE_BYTES + lane * 16;
    i32x4 A[4];
    rq_dsread<0>(A[0], xa);
    rq_dsread<1024>(A[1], xa);
    rq_dsread<2048>(A[2], xa);
    __builtin_amdgcn_sched_barrier(0);
    rq8_ksteps<KS, NW, DPW, PL, 0>(xa, A, acc, Q, refill, vo);
    if constexpr (DOM) {
      // rows 16 half + 4 hb + e of the tile: piece `half` (slab 0), lane (q4 = 0, r = 4 hb + e), its first four bytes
      const unsigned da = lds_base + slot * TILE_BYTES + hb * 64;
      int x4[8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(x4[r]) : "v"(da), "n"((r >> 2) * 1024 + (r & 3) * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(x4[r]));
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int c = __builtin_amdgcn_sdot4(x4[r], dhi[b], 0, false) << 7;
          acc[0][b][r >> 2][r & 3] += __builtin_amdgcn_sdot4(x4[r], dlo[b], c, false);
        }
    }
    RQ8_STAMP(1)

    // ---- filter: lane (qcol, hb) owns rows row0 + 16 half + e of its query column in each block; integer compares
    const int64_t row0 = t * tstep * 32 + 4 * hb;
    // (two planes: |128 * plane 0| <= 128 * 127 * 127 * d < 2^31 at d <= 1024, plane 1 adds at most 127 * 127 * d)
#define RQ8_SC(b, r) (PL == 2 ? acc[0][b][(r) >> 2][(r) & 3] * 128 + acc[PL - 1][b][(r) >> 2][(r) & 3] : acc[0][b][(r) >> 2][(r) & 3])
    bool any = false;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int m = RQ8_SC(b, 0);
#pragma unroll
      for (int r = 1; r < 8; ++r) m = max(m, RQ8_SC(b, r));
      any |= m >= tq[b];
    }
    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
      bool vmem = false;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int64_t row = row0 + 16 * (r >> 2) + (r & 3);
          const bool hit = RQ8_SC(b, r) >= tq[b] && row < N;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m != 0ull) {
            const int pos = nst + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned qq = (unsigned)((w * 2 + b) * 16 + qcol);
            if (hit) {
              if (pos < RQ_STAGE) {
                st_s[pos] = (float)RQ8_SC(b, r);  // (overwritten by the exact score: knn_rq_rescore_kernel)
                st_r[pos] = (uint32_t)row;
                st_q[pos] = qq;
              } else {
                g_lost[qq] = 1u;
                vmem = true;
              }
            }
            nst += __builtin_popcountll(m);
          }
        }
      }
      if (nst > RQ_STAGE) nst = RQ_STAGE;
      if (nst >= RQ_FLUSH_AT) {
        vmem = true;
        for (int i = lane; i < nst; i += 64) {
          const unsigned qq = st_q[i];
          const unsigned pos = atomicAdd(&g_cnt[qq], 1u);
          if (pos < cap) {
            hit_s[(size_t)qq * cap + pos] = st_s[i];
            hit_r[(size_t)qq * cap + pos] = st_r[i];
          }
        }
        nst = 0;
      }
      if (__builtin_amdgcn_ballot_w64(vmem) != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
#ifdef CLIPX_ABLATE
    RQ8_STAMP(2)
    if (timer) tph[3] += 1;
#endif
  }
#ifdef CLIPX_ABLATE
  if (timer && lane == 0 && blockIdx.x < 256 && w < 8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g_rq8_phase[(blockIdx.x * 8 + w) * 4 + i] = tph[i];
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = lane; i < nst; i += 64) {
    const unsigned qq = st_q[i];
    const unsigned pos = atomicAdd(&g_cnt[qq], 1u);
    if (pos < cap) {
      hit_s[(size_t)qq * cap + pos] = st_s[i];
      hit_r[(size_t)qq * cap + pos] = st_r[i];
    }
  }
}
#undef RQ8_SC

// proof of the int8 path: the hit list is complete and its k-th exact score reaches the lower bound T -- then every row of the true
// top-k was a hit (header above).  need / gate / stats as knn_rq_proof_kernel.
__global__ __launch_bounds__(256) void knn_i8_proof_kernel(int nq, int k, const float* __restrict__ D, const float* __restrict__ thr_lb,
                                                          const unsigned* __restrict__ cnt, unsigned cap, const unsigned* __restrict__ lost,
                                                          unsigned* __restrict__ need, unsigned* __restrict__ gate,
                                                          unsigned long long* __restrict__ stats) {
  __shared__ unsigned s_need[256];
  const int qq = threadIdx.x;
  unsigned nd = 0u;
  if (qq < nq) {
    const bool complete = cnt[qq] <= cap && lost[qq] == 0u;
    const float dk = D[(size_t)qq * k + (k - 1)];
    const float t = thr_lb[qq];
    const bool proven = complete && (!(t > -INFINITY) || (dk > -FLT_MAX && dk >= t));
    nd = proven ? 0u : 1u;
    need[qq] = nd;
  }
  s_need[threadIdx.x] = nd;
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned g = 0u;
    for (int i = 0; i < 32; ++i) g += s_need[threadIdx.x * 32 + i];
    gate[threadIdx.x] = g ? 1u : 0u;
    if (stats && g) atomicAdd(&stats[1], (unsigned long long)g);
  }
  if (threadIdx.x == 0 && stats) atomicAdd(&stats[0], (unsigned long long)nq);
}
#endif  // KNNX_MFMA16

int i8_supported(int d) { return KNNX_MFMA16 && (d == 512 || d == 768 || d == 1024) ? 1 : 0; }

// quantise rows with the column scales that exist (also used for rows added later: values beyond +-127 c clamp, and A / B -- which
// are maxima over the rows as stored -- grow with them, so the bound stays a bound)
hipError_t launch_i8_quant(const _Float16* X, int64_t N, int64_t row_from, int64_t row_to, int d, const float* colscale, const I8Dom& dom,
                           int8_t* X8, int* ab_enc, hipStream_t st) {
#if KNNX_MFMA16
  // rows [row_from, row_to) of an image that covers rows [0, N) (row_to <= N); whole half tiles are written: rows below row_from that
  // share its half tile are re-quantised to the same bytes, rows at or beyond N become zeros
  const int64_t h0 = row_from >> 4, h1 = (std::max(row_to, row_from) + 15) >> 4;
  const int64_t h1p = row_to >= N ? ((h1 + 1) & ~(int64_t)1) : h1;  // the end of the index: zero the rest of its last 32-row tile
  if (h1p <= h0) return hipSuccess;
  // grid-stride: a launch's grid x block must stay below 2^32 work-items
  const unsigned grid = (unsigned)std::min<int64_t>((h1p - h0 + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(knn_i8_quant_kernel, dim3(grid), dim3(256), 0, st, X, N, h0, h1p, d, colscale, dom, X8, ab_enc);
  return hipGetLastError();
#else
  return hipErrorInvalidValue;
#endif
}

// column scales over ALL N rows (the image itself may hold fewer: launch_i8_quant); resets A and B
hipError_t launch_i8_scales(const _Float16* X, int64_t N, int d, int* colmax_enc, float* colscale, int* ab_enc, hipStream_t st) {
#if KNNX_MFMA16
  hipError_t e = hipMemsetAsync(colmax_enc, 0, (size_t)d * sizeof(int), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(ab_enc, 0, 2 * sizeof(int), st);
  if (e != hipSuccess) return e;
  const unsigned g1 = (unsigned)std::min<int64_t>(std::max<int64_t>(N, 1), 256 * 32);
  hipLaunchKernelGGL(knn_i8_colmax_kernel, dim3(g1), dim3(256), 0, st, X, N, d, colmax_enc);
  hipLaunchKernelGGL(knn_i8_colscale_kernel, dim3((d + 255) / 256), dim3(256), 0, st, colmax_enc, d, colscale);
  return hipGetLastError();
#else
  return hipErrorInvalidValue;
#endif
}

hipError_t launch_i8_prep(const float* q_dev, int nq, int d, const float* colscale, const I8Dom& dom, const int* ab_enc, const int* maxnorm,
                          const float* samp, int kw, int J, int planes, int refine, int8_t* qfrag8, int8_t* qdom, int* thr_i, float* thr_lb,
                          float* thr_rest, unsigned* cnt, unsigned* lost, hipStream_t st) {
#if KNNX_MFMA16
  if (dom.n > 0 && (planes != 1 || !qdom)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(knn_i8_prep_kernel, dim3(256), dim3(64), 0, st, q_dev, nq, d, colscale, dom, ab_enc, maxnorm, samp, kw, J, planes, refine,
                     qfrag8, qdom, thr_i, thr_lb, thr_rest, cnt, lost);
  return hipGetLastError();
#else
  return hipErrorInvalidValue;
#endif
}

#if KNNX_MFMA16
template <int KS, int NW, int NSLOT, int PL = 1, bool DOM = false>
static hipError_t launch_rq8_scan_cfg(const int8_t* X8, int64_t N, const int8_t* qfrag8, const int8_t* qdom, const int* thr_i, unsigned* cnt,
                                      unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, int grid, int tstep, hipStream_t st) {
  const size_t smem = (size_t)NSLOT * KS * 1024 + (size_t)NW * RQ_STAGE * 12;
  auto kern = knn_rq8_scan_kernel<KS, NW, NSLOT, PL, DOM>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, X8, N, qfrag8, qdom, thr_i, cnt, cap, hit_s, hit_r, lost, tstep);
  return hipGetLastError();
}
#endif

#if defined(CLIPX_ABLATE) && KNNX_MFMA16
extern "C" int knnx_dbg_rq8_phases(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rq8_phase), (size_t)n * sizeof(long long));
}
#endif
hipError_t launch_rq8_scan(const int8_t* X8, int64_t N, int d, int nq, int planes, const int8_t* qfrag8, const int8_t* qdom, const int* thr_i,
                           unsigned* cnt, unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, int grid, int tstep, hipStream_t st) {
  // qdom != nullptr: the index has dominant columns (one plane; I8Dom)
  if (tstep < 1 || (qdom && planes != 1)) return hipErrorInvalidValue;
#if defined(CLIPX_ABLATE) && KNNX_MFMA16
  {
    static const int tm = getenv("KNNX_RQ8_TIMER") ? atoi(getenv("KNNX_RQ8_TIMER")) : 0;
    static bool set = false;
    if (!set) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rq8_timer), &tm, sizeof(int)); set = true; }
  }
#endif
#if KNNX_MFMA16
#define RQ8_ARGS X8, N, qfrag8, qdom, thr_i, cnt, cap, hit_s, hit_r, lost, grid, tstep, st
  if (planes == 2) {  // two query planes: 4 waves x 32 queries (twice the fragments per query: 192 registers at d = 768)
    if (nq > 128) return hipErrorInvalidValue;
    switch (d) {
      case 512: return launch_rq8_scan_cfg<16, 4, 4, 2>(RQ8_ARGS);
      case 768: return launch_rq8_scan_cfg<24, 4, 4, 2>(RQ8_ARGS);
      case 1024: return launch_rq8_scan_cfg<32, 4, 4, 2>(RQ8_ARGS);
      default: return hipErrorInvalidValue;
    }
  }
  // up to 128 queries: four waves hold them all -- half the LDS reads and MFMAs of the 8-wave configuration, which a pass over
  // int8 rows does not hide behind HBM (8 x 32 slots: 17.4 ms per pass over 100 M x 768 whatever the batch, 76.8 GB in 12.4 ms)
  if (nq <= 128) {
    switch (d) {
      case 512: return qdom ? launch_rq8_scan_cfg<16, 4, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<16, 4, 4>(RQ8_ARGS);
      case 768: return qdom ? launch_rq8_scan_cfg<24, 4, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<24, 4, 4>(RQ8_ARGS);
      case 1024: return qdom ? launch_rq8_scan_cfg<32, 4, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<32, 4, 4>(RQ8_ARGS);
      default: return hipErrorInvalidValue;
    }
  }
  switch (d) {
    case 512: return qdom ? launch_rq8_scan_cfg<16, 8, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<16, 8, 4>(RQ8_ARGS);
    case 768: return qdom ? launch_rq8_scan_cfg<24, 8, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<24, 8, 4>(RQ8_ARGS);
    case 1024: return qdom ? launch_rq8_scan_cfg<32, 8, 4, 1, true>(RQ8_ARGS) : launch_rq8_scan_cfg<32, 8, 4>(RQ8_ARGS);
    default: return hipErrorInvalidValue;
  }
#undef RQ8_ARGS
#else
  return hipErrorInvalidValue;
#endif
}

hipError_t launch_i8_proof(int nq, int k, const float* D, const float* thr_lb, const unsigned* cnt, unsigned cap, const unsigned* lost,
                           unsigned* need, unsigned* gate, unsigned long long* stats, hipStream_t st) {
#if KNNX_MFMA16
  hipLaunchKernelGGL(knn_i8_proof_kernel, dim3(1), dim3(256), 0, st, nq, k, D, thr_lb, cnt, cap, lost, need, gate, stats);
  return hipGetLastError();
#else
  return hipErrorInvalidValue;
#endif
}

// ---------------------------------------------------------------------------------------------
// IVF build: list assignment = argmax over the centroids, for MANY points per launch (takes the place of
// the autofaiss k-means / add of clip_index.py:12-66 for this index type).  The same register-stationary structure as the
// scan with the roles of the build: a workgroup keeps ITS OWN NW x 32 points as the stationary operand (fp16 rows are
// MFMA fragments as they lie in HBM) and streams ALL centroid tiles through the LDS ring; a lane owns one point and keeps
// the running best (score, centroid) over its 16 rows of every tile.  Every workgroup reads the whole centroid matrix
// (nlist x d x 2 B, L2 / Infinity-Cache resident: all workgroups walk it in step), so the kernel is MFMA-bound:
// 2 n nlist d flops.  Scores are exact fp32 sums of fp16 x fp16 products; ties go to the smaller centroid id.
// ---------------------------------------------------------------------------------------------
template <int KS, int NW, int NSLOT>
__global__ __launch_bounds__(NW * 64, NW / 4) void knn_assign_kernel(const _Float16* __restrict__ C, int64_t nlist,
                                                                     const _Float16* __restrict__ P, int64_t n,
                                                                     int32_t* __restrict__ out) {
  constexpr int D = KS * 16;
  constexpr int TILE_BYTES = KS * 1024;
  constexpr int DPW = KS / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NQB = RQ_NQB(1), NSL = RQ_NSL(KS), QBS = KNNX_MFMA16 ? 16 : 32;
  const int qcol = lane & (QBS - 1), hb = lane / QBS;
  const _Float16* X = C;  // the streamed operand (RQ_TILE below)
  const int64_t N = nlist;

  // this wave's 32 points as B fragments: lane (qcol, hb) holds point[16 s + 8 hb .. + 8] of every k-step s
  // (KNNX_MFMA16: two blocks of 16 points, point[32 s + 8 hb .. + 8] of every slab s)
  int64_t pidx[NQB];
  half8 Q[NQB][NSL];
#pragma unroll
  for (int b = 0; b < NQB; ++b) {
    pidx[b] = (int64_t)blockIdx.x * (NW * 32) + w * 32 + b * QBS + qcol;
    const int64_t prow = pidx[b] < n ? pidx[b] : n - 1;
    const half8* src = reinterpret_cast<const half8*>(P + (size_t)prow * D) + hb;
#pragma unroll
    for (int s = 0; s < NSL; ++s) Q[b][s] = src[(KNNX_MFMA16 ? 4 : 2) * s];
  }
#pragma unroll
  for (int b = 0; b < NQB; ++b)
#pragma unroll
    for (int s = 0; s < NSL; ++s) asm volatile("" : "+v"(Q[b][s]));  // all landed before the DMA ring starts (see the scan)

  const int64_t ntile = (N + 31) >> 5;
  const int64_t last = ntile - 1;
  const RqLaneOff lane_off = rq_lane_offsets(lane, D, (int)(N - 1 - last * 32));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);

#pragma unroll
  for (int i = 0; i < NSLOT - 1; ++i) rq_issue_all<NW, DPW>(RQ_TILE((int64_t)i, i));

  float best[NQB];
  int brow[NQB];
#pragma unroll
  for (int b = 0; b < NQB; ++b) {
    best[b] = -INFINITY;
    brow[b] = 0;
  }
  int slot = 0;
  for (int64_t t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (NSLOT - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const RqTile refill = RQ_TILE(t + (NSLOT - 1), slot == 0 ? NSLOT - 1 : slot - 1);
    rq_acc_t acc[NQB];
#if KNNX_MFMA16
#pragma unroll
    for (int b = 0; b < NQB; ++b) acc[b][0] = acc[b][1] = float4v{0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#endif
    const unsigned xa = lds_base + slot * TILE_BYTES + lane * 16;
    i32x4 A[4];
    rq_dsread<0>(A[0], xa);
    rq_dsread<1024>(A[1], xa);
    rq_dsread<2048>(A[2], xa);
    __builtin_amdgcn_sched_barrier(0);
    rq_ksteps<KS, 1, NW, DPW, 0>(xa, A, acc, Q, refill);
    // running argmax of this lane's point over its 16 centroid rows of the tile (rows ascend with r: strict > keeps the
    // smallest id among equal scores); the ragged last tile re-read centroid nlist - 1 into its padding rows
    const int row0 = (int)(t * 32) + 4 * hb;
    const bool ragged = t == last && (N & 31) != 0;
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
#pragma unroll
      for (int r = 0; r < (KNNX_MFMA16 ? 8 : 16); ++r) {
        const int row = row0 + RQ_ROWOFF(r);
        const float sc = RQ_SCORE(b, r);
        const bool take = sc > best[b] && (!ragged || row < (int)N);
        best[b] = take ? sc : best[b];
        brow[b] = take ? row : brow[b];
      }
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail reloads still in flight
  // the lane groups of a point (2 half-waves; KNNX_MFMA16: 4 quarter-waves) saw different rows of every tile
#pragma unroll
  for (int b = 0; b < NQB; ++b) {
#pragma unroll
    for (int o = 32; o >= QBS; o >>= 1) {
      const float ob = __shfl_xor(best[b], o);
      const int orow = __shfl_xor(brow[b], o);
      if (ob > best[b] || (ob == best[b] && orow < brow[b])) {
        best[b] = ob;
        brow[b] = orow;
      }
    }
    if (hb == 0 && pidx[b] < n) out[pidx[b]] = brow[b];
  }
}
#undef RQ_SCORE
#undef RQ_ROWOFF

template <int KS, int NW, int NSLOT>
static hipError_t launch_assign_cfg(const _Float16* C, int64_t nlist, const _Float16* P, int64_t n, int32_t* out, hipStream_t st) {
  const size_t smem = (size_t)NSLOT * KS * 1024;
  auto kern = knn_assign_kernel<KS, NW, NSLOT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  const int64_t per = NW * 32;
  hipLaunchKernelGGL(kern, dim3((unsigned)((n + per - 1) / per)), dim3(NW * 64), smem, st, C, nlist, P, n, out);
  return hipGetLastError();
}

hipError_t launch_assign(const _Float16* C, int64_t nlist, int d, const _Float16* P, int64_t n, int32_t* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (nlist <= 0) return hipErrorInvalidValue;
  switch (d) {
    case 256: return launch_assign_cfg<16, 8, 3>(C, nlist, P, n, out, st);
    case 512: return launch_assign_cfg<32, 8, 3>(C, nlist, P, n, out, st);
    case 768: return launch_assign_cfg<48, 8, 3>(C, nlist, P, n, out, st);
    case 1024: return launch_assign_cfg<64, 4, 2>(C, nlist, P, n, out, st);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------
// exact re-scoring of the hits, in place: one wave per hit, fp32 FMA chain over the lane's columns then a butterfly --
// the arithmetic of knn_rescore_kernel, so D does not depend on which path served the query.  grid = (chunks, nq).
// Also writes cntc[q] = min(cnt[q], cap) for the selection kernel.
// ---------------------------------------------------------------------------------------------
// NE = d / 64 column groups per lane, a template parameter: with a per-lane `column < d` test around every load hipcc put each
// two-byte load in its own branch with an s_waitcnt vmcnt(0) behind it -- 12 serialised HBM round trips per hit, 1.65 ms for
// 256 x ~6 900 hits over 100 M x 768 rows (round 3 disassembly).  Unrolled and unconditional, the 2 x NE loads of two hits are
// in flight together.  Per hit the arithmetic is unchanged: FMA chain over the lane's columns in column order, then the butterfly.
template <int NE>
__global__ __launch_bounds__(256) void knn_rq_rescore_kernel(const _Float16* __restrict__ X, const float* __restrict__ q,
                                                            const unsigned* __restrict__ cnt, unsigned cap,
                                                            float* __restrict__ hit_s, const uint32_t* __restrict__ hit_r,
                                                            int* __restrict__ cntc) {
  constexpr int d = NE * 64;
  const int qq = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned n = cnt[qq] < cap ? cnt[qq] : cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) cntc[qq] = (int)n;
  const float* qv = q + (size_t)qq * d;
  float qreg[NE];  // the lane's query columns
#pragma unroll
  for (int e = 0; e < NE; ++e) qreg[e] = qv[e * 64 + lane];
  const unsigned step = gridDim.x * 4;
  const uint32_t* hr = hit_r + (size_t)qq * cap;
  // FOUR hits per wave and iteration (round 6; two before): 4 x NE two-byte loads in flight -- the pass is a random gather of
  // 1.5 KiB rows (0.8 ms for 256 x ~6 900 hits at two per iteration: 3.4 TB/s).  Per hit the arithmetic is unchanged.
  constexpr int H = 4;
  for (unsigned i0 = blockIdx.x * 4 + w; i0 < n; i0 += H * step) {
    unsigned idx[H];
    bool ok[H];
    const _Float16* xp[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      idx[h] = i0 + h * step;
      ok[h] = idx[h] < n;
      xp[h] = X + (size_t)hr[ok[h] ? idx[h] : i0] * d + lane;
    }
    _Float16 a[H][NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int h = 0; h < H; ++h) a[h][e] = xp[h][e * 64];
    float acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int h = 0; h < H; ++h) acc[h] = __builtin_fmaf((float)a[h][e], qreg[e], acc[h]);
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int h = 0; h < H; ++h) acc[h] += __shfl_xor(acc[h], o);
    if (lane == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h)
        if (ok[h]) hit_s[(size_t)qq * cap + idx[h]] = acc[h];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// proof: need[q] = 0 iff the top-k of query q is proven exact (see the file header); gate[g] = any need in queries
// 32g .. 32g+31.  D [nq, k] is the exact top-k of the hits (knn_merge_kernel).  One 256-thread workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_rq_proof_kernel(const float* __restrict__ q, int nq, int d, int k,
                                                          const float* __restrict__ D, const float* __restrict__ thr,
                                                          const unsigned* __restrict__ cnt, unsigned cap,
                                                          const unsigned* __restrict__ lost, const int* __restrict__ maxnorm,
                                                          unsigned* __restrict__ need, unsigned* __restrict__ gate,
                                                          unsigned long long* __restrict__ stats) {
  __shared__ unsigned s_need[256];
  const int qq = threadIdx.x;
  unsigned nd = 0u;
  if (qq < nq) {
    const float* qv = q + (size_t)qq * d;
    float e2 = 0.f, n2 = 0.f;
    for (int c = 0; c < d; ++c) {
      const float v = qv[c];
      const float r = v - (float)(_Float16)v;
      e2 += r * r;
      n2 += v * v;
    }
    // |approx - exact| <= eps for every row: fp16 rounding of the query + accumulation-order slack (cf. knn_rescore_kernel)
    const float eps = (sqrtf(e2) + (float)d * 1.2e-7f * sqrtf(n2)) * rq_dec_f(*maxnorm);
    const float t = thr[qq];
    const bool complete = cnt[qq] <= cap && lost[qq] == 0u;  // every row that reached the threshold was re-scored
    const float dk = D[(size_t)qq * k + (k - 1)];            // k-th exact score among the hits (-FLT_MAX: fewer than k)
    const bool all_rows = !(t > -INFINITY);                  // threshold -inf: the hits are the whole index
    const bool proven = complete && (all_rows || (dk > -FLT_MAX && dk >= t + eps));
    nd = proven ? 0u : 1u;
    need[qq] = nd;
  }
  s_need[threadIdx.x] = nd;
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned g = 0u;
    for (int i = 0; i < 32; ++i) g += s_need[threadIdx.x * 32 + i];
    gate[threadIdx.x] = g ? 1u : 0u;
    if (stats && g) atomicAdd(&stats[1], (unsigned long long)g);
  }
  if (threadIdx.x == 0 && stats) atomicAdd(&stats[0], (unsigned long long)nq);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int rq_queries_per_pass(int d) { return d == 1024 ? 128 : (d == 512 || d == 768 ? 256 : 0); }

hipError_t launch_rq_prep(const float* q_dev, int nq, int d, _Float16* qfrag, const float* samp, int kw, int J, float slack,
                          float* thr, unsigned* cnt, unsigned* lost, hipStream_t st) {
  const int nblk = rq_queries_per_pass(d) / (KNNX_MFMA16 ? 16 : 32);
  hipLaunchKernelGGL(knn_rq_prep_kernel, dim3(d / (KNNX_MFMA16 ? 32 : 16), nblk), dim3(64), 0, st, q_dev, nq, d, nblk, qfrag, samp, kw, J, slack, thr, cnt, lost);
  return hipGetLastError();
}

template <int KS, int QBW, int NW, int NSLOT>
static hipError_t launch_rq_scan_cfg(const _Float16* X, int64_t N, const _Float16* qfrag, const float* thr, unsigned* cnt,
                                     unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, const unsigned* gate, int grid,
                                     hipStream_t st, uint32_t row_off) {
  const size_t smem = (size_t)NSLOT * KS * 1024 + (size_t)NW * RQ_STAGE * 12;
  auto kern = knn_rq_scan_kernel<KS, QBW, NW, NSLOT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, row_off);
  return hipGetLastError();
}

hipError_t launch_rq_scan(const _Float16* X, int64_t N, int d, int nq, const _Float16* qfrag, const float* thr, unsigned* cnt,
                          unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, const unsigned* gate, int grid,
                          hipStream_t st, uint32_t row_off) {
  // up to 128 queries: four waves hold them all -- half the LDS reads of the X tiles and half the MFMAs of the 8-wave
  // configuration, whose upper four waves would multiply padding
  if (nq <= 128 && d == 768) return launch_rq_scan_cfg<48, 1, 4, 3>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
  if (nq <= 128 && d == 512) return launch_rq_scan_cfg<32, 1, 4, 4>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
#ifdef CLIPX_ABLATE
  {  // tools build only (A/B on one box): KNNX_RQ_4X64=1 -> 4 waves x 64 queries at d = 768 (half the LDS reads, one wave per SIMD):
     // 38.3 ms per pass against 36.4 - 36.8 for 8 x 32 on the 16x16x32 kernels (profiles/r04p_rq_8x32_vs_4x64.log)
    static const int w4 = getenv("KNNX_RQ_4X64") ? atoi(getenv("KNNX_RQ_4X64")) : 0;
    if (w4 && d == 768) return launch_rq_scan_cfg<48, 2, 4, 3>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
  }
#endif
  switch (d) {
    case 512: return launch_rq_scan_cfg<32, 1, 8, 4>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
    // d = 768: 8 waves x 32 queries, two waves per SIMD: a wave's LDS-DMA issue (~100 cycles per instruction during which it
    // issues nothing else) is covered by its SIMD partner's MFMAs.  4 waves x 64 queries (one wave per SIMD, 501 registers)
    // measured 59 % MFMA utilisation at 1.6 GHz: the 12 DMA issues per tile held the matrix pipe of their SIMD idle
    // (round 2 counter run; the file is no longer kept under profiles/).
    case 768: return launch_rq_scan_cfg<48, 1, 8, 3>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
    case 1024: return launch_rq_scan_cfg<64, 1, 4, 2>(X, N, qfrag, thr, cnt, cap, hit_s, hit_r, lost, gate, grid, st, row_off);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_rq_rescore(const _Float16* X, int d, const float* q, int nq, const unsigned* cnt, unsigned cap, float* hit_s,
                             const uint32_t* hit_r, int* cntc, hipStream_t st) {
  switch (d) {
    case 512: hipLaunchKernelGGL(knn_rq_rescore_kernel<8>, dim3(32, nq), dim3(256), 0, st, X, q, cnt, cap, hit_s, hit_r, cntc); break;
    case 768: hipLaunchKernelGGL(knn_rq_rescore_kernel<12>, dim3(32, nq), dim3(256), 0, st, X, q, cnt, cap, hit_s, hit_r, cntc); break;
    case 1024: hipLaunchKernelGGL(knn_rq_rescore_kernel<16>, dim3(32, nq), dim3(256), 0, st, X, q, cnt, cap, hit_s, hit_r, cntc); break;
    default: return hipErrorInvalidValue;  // rq_queries_per_pass(d) == 0 for every other d: the RQ path is never taken
  }
  return hipGetLastError();
}

hipError_t launch_rq_proof(const float* q, int nq, int d, int k, const float* D, const float* thr, const unsigned* cnt,
                           unsigned cap, const unsigned* lost, const int* maxnorm, unsigned* need, unsigned* gate,
                           unsigned long long* stats, hipStream_t st) {
  hipLaunchKernelGGL(knn_rq_proof_kernel, dim3(1), dim3(256), 0, st, q, nq, d, k, D, thr, cnt, cap, lost, maxnorm, need, gate, stats);
  return hipGetLastError();
}

}  // namespace knnx
