// gemm256sp.hip -- persistent 256x256x64 software-pipelined bf16 GEMM for gfx950 (GemmArgs.variant == 3).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A bf16 [M, K] activations, W bf16 [N, K] (torch Linear)
//
// Same role and operand layout as gemm256.hip (the linear layers inside `model.encode_image/encode_text`, reference
// clip_retrieval/clip_inference/mapper.py:57,65), different schedule.  Measured on MI355X the two-barriers-per-
// 8-MFMA ping-pong of gemm256.hip tops out at ~50 % of the MFMA roof with NO memory traffic at all (barrier hand-off
// latency per 256-cycle phase); this kernel synchronises ONCE per K-tile instead:
//
//   * one 512-thread workgroup per CU, persistent over output tiles as one continuous K-tile stream (as gemm256).
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4x2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator VGPRs).
//   * LDS: 2 K-tile buffers x (M operand 256 rows + N operand 256 rows) x 128 B = 128 KiB, filled by LDS-DMA
//     (global_load_lds_dwordx4, 8 per wave per K-tile), chunk-XOR swizzled through the source address;
//     + 32 KiB of per-wave scratch for the epilogue transposition = the CU's whole 160 KiB.
//   * per K-tile each wave runs 4 k-steps of {6 ds_read_b128 for the NEXT step, 8 MFMA of this step} from two
//     register sets, so LDS latency is always covered by a step's MFMAs; the single s_barrier of the K-tile sits
//     BEFORE the last step's MFMAs, with the reads of the next K-tile's first step issued right behind it:
//         step 3:  lgkmcnt(0); vmcnt(0)  [K-tile g+1 landed]; s_barrier;
//                  stage K-tile g+2 into the buffer just released; ds_read step 0 of K-tile g+1; 8 MFMA
//     so the matrix pipe only sees the barrier's own hand-off latency once per 2048 pipe-cycles.
//   * epilogue: accumulators (+bias, activation) are transposed through the wave's private LDS scratch so that
//     every global access is 16 B per lane with 8 consecutive lanes covering one full 128-B line of a row
//     (the MFMA layout itself gives 8-B pieces scattered over 32 rows per instruction).
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0 (the launcher in clip_kernels.hip peels ragged rows).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace clipx {

// LDS map: M operand of K-tile buffer b at b*32 KiB, N operand at 64 KiB + b*32 KiB (each [256 rows][128 B]), so that
// every fragment read is <per-lane base of (operand, k-step)> + a 16-bit immediate.
constexpr int S_OPB = 32768;      // bytes of one operand tile
constexpr int S_NBASE = 65536;    // N operand tiles start here
constexpr int S_SCRATCH = 131072; // per-wave 4 KiB epilogue scratch starts here

#define S_FENCE() __builtin_amdgcn_sched_barrier(0)

// Output stores are write-through and do not keep their lines in the XCD's L2 (sc1): one tile round writes
// 32 CUs x 128 KiB = the whole 4 MiB L2 of an XCD, and with plain stores that round evicts the operand panels the
// next K-tiles re-read (measured: the stores alone cost 24 % of the QKV GEMM, 215 k of 888 k cycles).  flags bit 2
// switches back to plain stores for A/B.  (s_nop 1: the data registers must not be rewritten before the store read them.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(void* p, uint4 v, bool plain) {
  if (plain) {
    *reinterpret_cast<uint4*>(p) = v;
  } else {
    const u32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
  }
}

// DBG (ablation, EPI_BIAS_BF16 only; garbage results): 1 = no staging, 3 = no staging and no ds_reads,
// 5 = normal main loop, no epilogue at all; 6 = epilogue without its global stores; 7 = XCD x starts x/8 of a tile
// period late (tests whether the tile-boundary cost is the chip-wide simultaneous store burst)
template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void gemm256sp_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ outp,
                                                          const float* __restrict__ table, int T, int N, int K, int ntm,
                                                          int ntn, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;  // waves w and w+4 share a SIMD (measured placement); rows differ, harmless here
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
  if (DBG == 7) {
    const long long t0 = __builtin_readcyclecounter();  // s_memtime: shader-clock ticks
    const long long wait = (long long)xcd * (K >> 6) * 420;  // (K/64 K-tiles) * ~3300 cycles / 8 per XCD step
    while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }

  // ---- staging: wave w fills rows [32w, 32w+32) of both operands, 8 rows (1 KiB) per instruction
  // (piece 4w + j = rows 32w + 8j .. +8).  Source chunk = LDS chunk position ^ ((row>>1)&7); rows 8 apart flip bit 2
  // of that key, so even and odd j use two per-lane offsets.
  const int srow = w * 32 + (lane >> 3);
  const int c0 = (lane & 7) ^ ((srow >> 1) & 7);
  const unsigned offE = (unsigned)((srow * K + (c0 << 3)) * 2);
  const unsigned offO = (unsigned)((srow * K + ((c0 ^ 4) << 3)) * 2);
  const size_t jstep = (size_t)8 * K * 2;
  typedef unsigned dbg_u32x4 __attribute__((ext_vector_type(4)));
  dbg_u32x4 dbg_sink = {0u, 0u, 0u, 0u};
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
  auto stage = [&](const char* baseM, const char* baseN, int buf) {
    if (DBG == 1 || DBG == 3) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = (j & 1) ? offO : offE;
      if (DBG == 11) {  // as DBG 10 but SGPR base + 32-bit VGPR offset (saddr form): is the cost the 64-bit address transfer?
        const char* bm = baseM + j * jstep;
        const char* bn = baseN + j * jstep;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dbg_sink) : "v"(off), "s"(bm) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dbg_sink) : "v"(off), "s"(bn) : "memory");
        continue;
      }
      if (DBG == 10) {  // the same loads into a (dummy) register instead of the LDS: is the per-instruction cost TA- or LDS-side?
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dbg_sink) : "v"(baseM + j * jstep + off) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dbg_sink) : "v"(baseN + j * jstep + off) : "memory");
        continue;
      }
      if (DBG == 9) {  // same instruction count, a quarter of the bytes
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseM + j * jstep + off),
                                         (lds_ptr_t)(smem + buf * S_OPB + (w * 4 + j) * 1024), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseN + j * jstep + off),
                                         (lds_ptr_t)(smem + S_NBASE + buf * S_OPB + (w * 4 + j) * 1024), 4, 0, 0);
        continue;
      }
      if (!(flags & 8)) {
        // SGPR base + 32-bit VGPR offset ("saddr") form: half the per-lane address bits to move per DMA instruction
        // (ablation DBG 11: -18 % of the per-instruction cost).  M0 (LDS destination) is written in the same statement.
        const char* bm = baseM + j * jstep;
        const char* bn = baseN + j * jstep;
        const unsigned dm = lds_base + buf * S_OPB + (w * 4 + j) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(bm), "s"(dm) : "memory");
        if (DBG == 8) continue;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(bn), "s"(dm + S_NBASE) : "memory");
        continue;
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseM + j * jstep + off),
                                       (lds_ptr_t)(smem + buf * S_OPB + (w * 4 + j) * 1024), 16, 0, 0);
      if (DBG == 8) continue;  // half the bytes and half the instructions: M operand only
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(baseN + j * jstep + off),
                                       (lds_ptr_t)(smem + S_NBASE + buf * S_OPB + (w * 4 + j) * 1024), 16, 0, 0);
    }
  };

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

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // [0..3] M fragments (mi), [4..5] N fragments (ni) of one k-step; held as 4 x b32 so that hipcc keeps each
  // fragment one 128-bit register tuple across the loop back-edge (as 8 x bf16 it re-packs them with v_perm_b32)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 F0[6], F1[6];
  if (DBG == 3) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      F0[i] = F1[i] = i32x4{0, 0, 0, 0};
      asm volatile("" : "+v"(F0[i]), "+v"(F1[i]));
    }
  }

#define S_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define S_READ(F, buf, kk)                                                                 \
  if (DBG != 3) {                                                                          \
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
#define S_WAIT_ALL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); S_FENCE();
#define S_MFMA(F)                                                                                                  \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] =  \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[4 + ni]), __builtin_bit_cast(bf16x8, F[mi]), acc[mi][ni], 0, 0, 0);

  const char* curM = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* curN = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;
  const int nk = K >> 6;  // K-tiles per output tile (even, >= 2)

  // VMEM bookkeeping (vmcnt retires in order): "the stage of the next K-tile has landed" is vmcnt(0) in steady state.
  // On the first K-tile after an epilogue the epilogue's own stores (S_EPI_ST per wave) are younger than that stage
  // and may stay in flight: vmcnt(S_EPI_ST).  (An L2 prefetch of the K-tile 4 ahead was tried and measured neutral:
  // the exposed staging time is LDS-DMA throughput, not HBM latency -- staging an L2-resident K-tile costs the same.)
  constexpr bool OUT_BF16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16;
  constexpr int S_EPI_ST = (DBG == 5 || DBG == 6) ? 0 : (OUT_BF16 ? 16 : 32);  // f32: 32 loads + 32 stores follow; 32 youngest = stores

  // ---- prologue: K-tiles 0, 1 of the first tile; K-tile 0 landed + first fragment set read
  stage(curM, curN, 0);
  stage(curM + 128, curN + 128, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  S_FENCE();
  __builtin_amdgcn_s_barrier();
  S_FENCE();
  S_READ(F0, 0, 0)

  // bias of the current tile: one LDS-DMA per wave at the tile's last K-tile drops the wave's 64 bias floats into
  // its (then idle) epilogue scratch, AHEAD of the next tile's first operand DMA, so the epilogue neither queues
  // behind those nor holds bias registers across the main loop.  (All four 16-lane groups fetch the same 256 B.)
  constexpr bool HAS_BIAS = EPI != EPI_TABLE_F32;
  const int rrow = lane >> 3, rch = lane & 7;  // epilogue read-back: row 8i + rrow, 16-B chunk rch
  auto load_bias = [&]() {
    if (!HAS_BIAS) return;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bias + n0 + wc * 64 + (lane & 15) * 4),
                                     (lds_ptr_t)(smem + S_SCRATCH + w * 4096), 16, 0, 0);
  };

  bool first = false;  // the K-tile about to run is the first one after an epilogue
  for (int j = 0;; ++j) {
    int nm0 = 0, nn0 = 0;
    const bool have_next = tile_of(j + 1, nm0, nn0);
    const char* nxtM = reinterpret_cast<const char*>(A) + (size_t)nm0 * K * 2;
    const char* nxtN = reinterpret_cast<const char*>(W) + (size_t)nn0 * K * 2;

    for (int t = 0; t < nk; t += 2) {
      // K-tile t (buffer 0) and t+1 (buffer 1).  Stage targets: K-tile t+2 -> buffer 0, t+3 -> buffer 1.
      const bool tail = t + 2 >= nk;
      const bool more = !tail || have_next;  // a K-tile t+2 / t+3 exists in this block's stream
      const char* sM = tail ? nxtM : curM + (size_t)(t + 2) * 128;
      const char* sN = tail ? nxtN : curN + (size_t)(t + 2) * 128;

#define S_KTILE(buf, is_first, bias_stmt, stage_stmt, next_ok) \
  S_READ(F1, buf, 1)                                         \
  S_WAIT_PREV()                                              \
  S_MFMA(F0)                                                 \
  S_FENCE();                                                 \
  S_READ(F0, buf, 2)                                         \
  S_WAIT_PREV()                                              \
  S_MFMA(F1)                                                 \
  S_FENCE();                                                 \
  S_READ(F1, buf, 3)                                         \
  S_WAIT_PREV()                                              \
  S_MFMA(F0)                                                 \
  S_FENCE();                                                 \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
  if ((is_first) && !(flags & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S_EPI_ST) : "memory"); \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
  S_FENCE();                                                 \
  __builtin_amdgcn_s_barrier();                              \
  S_FENCE();                                                 \
  bias_stmt;                                                 \
  stage_stmt;                                                \
  if (next_ok) { S_READ(F0, (buf) ^ 1, 0) }                  \
  S_FENCE();                                                 \
  S_MFMA(F1)                                                 \
  S_FENCE();

      S_KTILE(0, first, (void)0, if (more) stage(sM, sN, 0), true)
      first = false;
      S_KTILE(1, false, if (tail) load_bias(), if (more) stage(sM + 128, sN + 128, 1), more)
    }

    // ---- epilogue of this output tile, transposed through the wave's LDS scratch (the next tile's K-tiles 0/1 are
    // already in flight / landed and its first fragment set is in F0)
    S_WAIT_ALL()  // the next tile's first fragment set must have landed before hipcc may move/spill its registers
    // bias landed in the scratch: only the next tile's stage (8 DMA) was issued after its DMA
    if (have_next && !(flags & 2)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // last tile: no stage was issued behind the bias DMA
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
    first = true;
    m0 = nm0;
    n0 = nn0;
    curM = nxtM;
    curN = nxtN;
  }
  if (DBG == 10 || DBG == 11) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dbg_sink)::"memory");
}

template <int EPI, int DBG = 0>
static hipError_t launch_sp_epi(const GemmArgs& g, int grid, hipStream_t st) {
  const size_t smem = S_SCRATCH + 8 * 4096;  // 160 KiB: the whole LDS of the CU
  auto kern = gemm256sp_kernel<EPI, DBG>;
  const char* fl = getenv("CLIPX_GEMM_FLAGS");  // A/B switches: bit 1 = drain vmcnt(0) at every wait, bit 2 = plain (L2-resident) output stores, bit 3 = builtin (64-bit VGPR address) LDS-DMA
  const int flags = fl ? atoi(fl) : 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.N, g.K, g.M / 256,
                     g.N / 256, flags);
  return hipGetLastError();
}

hipError_t launch_gemm256sp(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (g.M <= 0 || g.M % 256 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K <= 0) return hipErrorInvalidValue;
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;  // one workgroup per CU; multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  if (g.epi == EPI_BIAS_BF16) {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) return launch_sp_epi<EPI_BIAS_BF16, 1>(g, grid, st);
    if (d == 3) return launch_sp_epi<EPI_BIAS_BF16, 3>(g, grid, st);
    if (d == 5) return launch_sp_epi<EPI_BIAS_BF16, 5>(g, grid, st);
    if (d == 6) return launch_sp_epi<EPI_BIAS_BF16, 6>(g, grid, st);
    if (d == 7) return launch_sp_epi<EPI_BIAS_BF16, 7>(g, grid, st);
    if (d == 8) return launch_sp_epi<EPI_BIAS_BF16, 8>(g, grid, st);
    if (d == 9) return launch_sp_epi<EPI_BIAS_BF16, 9>(g, grid, st);
    if (d == 10) return launch_sp_epi<EPI_BIAS_BF16, 10>(g, grid, st);
    if (d == 11) return launch_sp_epi<EPI_BIAS_BF16, 11>(g, grid, st);
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_sp_epi<EPI_BIAS_BF16>(g, grid, st);
    case EPI_BIAS_QGELU_BF16: return launch_sp_epi<EPI_BIAS_QGELU_BF16>(g, grid, st);
    case EPI_BIAS_GELU_BF16: return launch_sp_epi<EPI_BIAS_GELU_BF16>(g, grid, st);
    case EPI_BIAS_RESID_F32: return launch_sp_epi<EPI_BIAS_RESID_F32>(g, grid, st);
    case EPI_TABLE_F32: return launch_sp_epi<EPI_TABLE_F32>(g, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
