// gemm256.hip -- persistent 256x256x64 "ping-pong" bf16 GEMM for gfx950 (the dominant kernel of the encode half).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        A bf16 [M, K] activations, W bf16 [N, K] (torch Linear)
//
// replaces the torch/BLAS linear layers inside `model.encode_image` / `encode_text` that the reference calls at
// clip_retrieval/clip_inference/mapper.py:57,65.  M % 256 == 0, N % 256 == 0, K % 128 == 0 (the launcher in
// clip_kernels.hip peels ragged rows off to the 128x128 kernel).
//
// Structure (CDNA4: 4 SIMDs per CU, one matrix pipe per SIMD, 160 KiB LDS, LDS-DMA):
//   * one 512-thread workgroup per CU, persistent: block b walks output tiles b, b+grid, ... as ONE continuous
//     stream of K-tiles, so the LDS pipeline never drains at a tile boundary and the epilogue stores of tile i
//     overlap the operand loads of tile i+1.
//   * 8 waves = 2 (M) x 4 (N); a wave owns a 128(M) x 64(N) accumulator block = 4x2 v_mfma_f32_32x32x16_bf16 tiles
//     (128 accumulator VGPRs).  MFMA A operand = W rows, B operand = A rows, so a lane ends with 4 consecutive n of
//     one m (8/16-B epilogue stores).
//   * waves w and w+4 share a SIMD and belong to different wave rows.  Wave row 1 runs ONE s_barrier behind wave
//     row 0, so in every barrier interval one wave of each SIMD issues MFMAs (8 per phase = one 64x32 quadrant x
//     K=64) while its partner issues the ds_reads and LDS-DMA of its next phase.
//   * LDS: 2 K-tile buffers x {M-half0, M-half1, N-half0, N-half1} x 16 KiB.  "Half h" of an operand holds, for
//     every wave, the rows of its quadrant h, so each slot is read in exactly one of the four phases of a K-tile
//     and can be refilled two phases later.  One slot (2 x global_load_lds_dwordx4 per thread) is staged per phase;
//     s_waitcnt vmcnt(4) at phases 4 and 8 only -- two slots stay in flight across the barriers.
//   * LDS rows are 128 B (64 k); 16-B chunk c of row r is stored at chunk position c ^ ((r>>1)&7) (applied to the
//     per-lane SOURCE address, the DMA writes lane-linear): ds_read_b128 fragment reads are bank-conflict free.
//
// Phase table of one iteration (K-tiles E = 2i in buffer 0, O = 2i+1 in buffer 1):
//   phase  ds_read (slot)            stage (slot <- K-tile)        MFMA quadrant (qm, qn)
//     1    N0(E) x4, M0(E) x8        N1(O)                          (0,0)
//     2    N1(E) x4                  M1(O)                          (0,1)
//     3    M1(E) x8                  N0(E+2)                        (1,1)
//     4    -                         M0(E+2)   + vmcnt(4)           (1,0)
//     5    N0(O) x4, M0(O) x8        N1(E+2)                        (0,0)
//     6    N1(O) x4                  M1(E+2)                        (0,1)
//     7    M1(O) x8                  N0(O+2)                        (1,1)
//     8    -                         M0(O+2)   + vmcnt(4)           (1,0)
// Every slot is restaged >= 2 phases after its only read (WAR) and read >= 1 phase after the wait that retires it
// (RAW; the wait sits before the phase's first barrier, which every wave of both rows passes before the read).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gemm_common.h"

namespace clipx {

constexpr int P_HALF = 128 * 128;  // bytes of one half-tile slot (128 rows x 64 k x 2 B)
#define P_SLOT(buf, op, h) ((((buf) * 4) + (op) * 2 + (h)) * P_HALF)

#define P_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P_BARRIER()                  \
  {                                  \
    P_SCHED_FENCE();                 \
    __builtin_amdgcn_s_barrier();    \
    P_SCHED_FENCE();                 \
  }

// DBG (ablation builds, selected with CLIPX_GEMM_DBG for EPI_BIAS_BF16 only; results are garbage):
//   1 = no operand staging at all, 2 = always stage K-tile 0 (L2-resident operands), 3 = no staging and no ds_reads
template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                        const float* __restrict__ bias, void* __restrict__ outp,
                                                        const float* __restrict__ table, int T, int N, int K, int ntm,
                                                        int ntn, int wmap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> (row, column) of the 2x4 wave grid; the two waves of a SIMD must land in different rows
  const int wr = wmap ? (w & 1) : (w >> 2), wc = wmap ? (w >> 1) : (w & 3);
  const int hb = lane >> 5, l31 = lane & 31;
  const int ntiles = ntm * ntn;

  // ---- tile list of this block: round j gives XCD x (= blockIdx % 8) the contiguous run of `cpx` logical tiles
  // [(8j + x) * cpx, +cpx); logical tiles are ordered 8 m-tiles x all n-tiles per group, m fastest, so the 32 CUs
  // of an XCD work on a compact patch that shares operand rows in that XCD's L2.
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

  // ---- per-lane staging offsets (bytes, tile independent).  Piece pc = 2w + j of a slot = local rows 8pc..8pc+7.
  unsigned offM[2], offN[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lr = (w * 2 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((lr >> 1) & 7);
    offM[j] = (unsigned)((((lr >> 6) * 128 + (lr & 63)) * K + c * 8) * 2);
    offN[j] = (unsigned)((((lr >> 5) * 64 + (lr & 31)) * K + c * 8) * 2);
  }
  const size_t hM = (size_t)64 * K * 2, hN = (size_t)32 * K * 2;  // half 1 = +64 rows (M) / +32 rows (N)

  auto stage = [&](const char* base, const unsigned (&off)[2], int slot_byte) {
    if (DBG == 1 || DBG == 3) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off[j]), (lds_ptr_t)(smem + slot_byte + (w * 2 + j) * 1024), 16,
                                       0, 0);
  };

  // ---- fragment read offsets
  const int sw = (l31 >> 1) & 7;
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = ((2 * kk + hb) ^ sw) << 4;
  const unsigned char* fM = smem + (wr * 64 + l31) * 128;  // + slot + mi*4096 + xk
  const unsigned char* fN = smem + (wc * 32 + l31) * 128;  // + slot + xk

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 mf[2][4], nf0[4], nf1[4];
  if (DBG == 3) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      nf0[kk] = nf1[kk] = mf[0][kk] = mf[1][kk] = bf16x8{};
      asm volatile("" : "+v"(nf0[kk]), "+v"(nf1[kk]), "+v"(mf[0][kk]), "+v"(mf[1][kk]));
    }
  }

#define P_READ_M(buf, h)                                                                                      \
  if (DBG != 3) _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) mf[mi][kk] = \
      *reinterpret_cast<const bf16x8*>(fM + P_SLOT(buf, 0, h) + mi * 4096 + xk[kk]);
#define P_READ_N(dst, buf, h) \
  if (DBG != 3) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const bf16x8*>(fN + P_SLOT(buf, 1, h) + xk[kk]);
#define P_MFMA(QM, QN, NF)                                                                                     \
  {                                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)          \
        acc[2 * QM + mi][QN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(NF[kk], mf[mi][kk], acc[2 * QM + mi][QN], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  }

  const char* curM = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;
  const char* curN = reinterpret_cast<const char*>(W) + (size_t)n0 * K * 2;

  // ---- prologue: K-tile 0 -> buffer 0 (all four slots), K-tile 1 -> buffer 1 (N0, M0)
  stage(curN, offN, P_SLOT(0, 1, 0));
  stage(curM, offM, P_SLOT(0, 0, 0));
  stage(curN + hN, offN, P_SLOT(0, 1, 1));
  stage(curM + hM, offM, P_SLOT(0, 0, 1));
  stage(curN + 128, offN, P_SLOT(1, 1, 0));
  stage(curM + 128, offM, P_SLOT(1, 0, 0));
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  P_BARRIER();
  if (wr == 1) P_BARRIER();  // wave row 1 runs one barrier behind from here on

  const int nk2 = K >> 7;  // iterations (pairs of K-tiles) per output tile
  for (int j = 0;; ++j) {
    int nm0 = 0, nn0 = 0;
    const bool have_next = tile_of(j + 1, nm0, nn0);
    const char* nxtM = reinterpret_cast<const char*>(A) + (size_t)nm0 * K * 2;
    const char* nxtN = reinterpret_cast<const char*>(W) + (size_t)nn0 * K * 2;

    for (int it = 0; it < nk2; ++it) {
      const bool tail = it == nk2 - 1;
      const bool last = tail && !have_next;  // nothing left to stage beyond this iteration's own odd K-tile
      const char* sMo = curM + (DBG == 2 ? 0 : (size_t)(2 * it + 1) * 128);  // K-tile O of the current output tile
      const char* sNo = curN + (DBG == 2 ? 0 : (size_t)(2 * it + 1) * 128);
      const char* sMe = DBG == 2 ? curM : (tail ? nxtM : curM + (size_t)(2 * it + 2) * 128);  // K-tile E+2 (O+2 = +128 B)
      const char* sNe = DBG == 2 ? curN : (tail ? nxtN : curN + (size_t)(2 * it + 2) * 128);

      // ---------------- phase 1
      P_READ_N(nf0, 0, 0)
      P_READ_M(0, 0)
      stage(sNo + hN, offN, P_SLOT(1, 1, 1));
      P_BARRIER();
      P_MFMA(0, 0, nf0)
      P_BARRIER();
      // ---------------- phase 2
      P_READ_N(nf1, 0, 1)
      stage(sMo + hM, offM, P_SLOT(1, 0, 1));
      P_BARRIER();
      P_MFMA(0, 1, nf1)
      P_BARRIER();
      // ---------------- phase 3
      P_READ_M(0, 1)
      if (!last) stage(sNe, offN, P_SLOT(0, 1, 0));
      P_BARRIER();
      P_MFMA(1, 1, nf1)
      P_BARRIER();
      // ---------------- phase 4
      if (!last) {
        stage(sMe, offM, P_SLOT(0, 0, 0));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      P_BARRIER();
      P_MFMA(1, 0, nf0)
      P_BARRIER();
      // ---------------- phase 5
      P_READ_N(nf0, 1, 0)
      P_READ_M(1, 0)
      if (!last) stage(sNe + hN, offN, P_SLOT(0, 1, 1));
      P_BARRIER();
      P_MFMA(0, 0, nf0)
      P_BARRIER();
      // ---------------- phase 6
      P_READ_N(nf1, 1, 1)
      if (!last) stage(sMe + hM, offM, P_SLOT(0, 0, 1));
      P_BARRIER();
      P_MFMA(0, 1, nf1)
      P_BARRIER();
      // ---------------- phase 7
      P_READ_M(1, 1)
      if (!last) stage(sNe + 128, offN, P_SLOT(1, 1, 0));
      P_BARRIER();
      P_MFMA(1, 1, nf1)
      P_BARRIER();
      // ---------------- phase 8
      if (!last) {
        stage(sMe + 128, offM, P_SLOT(1, 0, 0));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      P_BARRIER();
      P_MFMA(1, 0, nf0)
      P_BARRIER();
    }

    // ---- epilogue of this output tile (no barriers: the next tile's operand loads are already in flight).
    // Loads are issued in batches ahead of their first use so that no wait drains this wave's own stores.
    {
      const int mrow = m0 + wr * 128 + l31;        // + 32*mt
      const int ncol = n0 + wc * 64 + 4 * hb;      // + 32*nt + 8*g
      float4 b4[2][4];
      if (EPI != EPI_TABLE_F32) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[nt][g] = *reinterpret_cast<const float4*>(bias + ncol + 32 * nt + 8 * g);
      }
      if (EPI == EPI_BIAS_RESID_F32) {
        float* xo = reinterpret_cast<float*>(outp);
        float4 ra[2][4], rb[2][4];
#define P_RLOAD(dst, mt)                                                                                  \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int g = 0; g < 4; ++g) dst[nt][g] = \
      *reinterpret_cast<const float4*>(xo + (size_t)(mrow + 32 * (mt)) * N + ncol + 32 * nt + 8 * g);
#define P_RSTORE(src, mt)                                                                                 \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int g = 0; g < 4; ++g) {         \
    float4 o = src[nt][g];                                                                                \
    o.x += acc[mt][nt][4 * g + 0] + b4[nt][g].x;                                                          \
    o.y += acc[mt][nt][4 * g + 1] + b4[nt][g].y;                                                          \
    o.z += acc[mt][nt][4 * g + 2] + b4[nt][g].z;                                                          \
    o.w += acc[mt][nt][4 * g + 3] + b4[nt][g].w;                                                          \
    *reinterpret_cast<float4*>(xo + (size_t)(mrow + 32 * (mt)) * N + ncol + 32 * nt + 8 * g) = o;         \
  }
        P_RLOAD(ra, 0)
        P_RLOAD(rb, 1)
        P_RSTORE(ra, 0)
        P_RLOAD(ra, 2)
        P_RSTORE(rb, 1)
        P_RLOAD(rb, 3)
        P_RSTORE(ra, 2)
        P_RSTORE(rb, 3)
#undef P_RLOAD
#undef P_RSTORE
      } else if (EPI == EPI_TABLE_F32) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                                           acc[mt][nt][4 * g + 3]);
              gemm_store_quad<EPI>(v, mrow + 32 * mt, ncol + 32 * nt + 8 * g, N, bias, outp, table, T);
            }
      } else {
        bf16* yo = reinterpret_cast<bf16*>(outp);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = acc[mt][nt][4 * g + e] + (e == 0 ? b4[nt][g].x : e == 1 ? b4[nt][g].y : e == 2 ? b4[nt][g].z : b4[nt][g].w);
                if (EPI == EPI_BIAS_QGELU_BF16) v[e] = quick_gelu(v[e]);
                if (EPI == EPI_BIAS_GELU_BF16) v[e] = gelu_erf(v[e]);
              }
              bf16x4 o;
              o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
              *reinterpret_cast<bf16x4*>(yo + (size_t)(mrow + 32 * mt) * N + ncol + 32 * nt + 8 * g) = o;
            }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    }
    if (!have_next) break;
    m0 = nm0;
    n0 = nn0;
    curM = nxtM;
    curN = nxtN;
  }
  if (wr == 0) P_BARRIER();  // pairs with wave row 1's last barrier
}

template <int EPI, int DBG = 0>
static hipError_t launch_gemm256_epi(const GemmArgs& g, int grid, hipStream_t st) {
  const size_t smem = 8 * P_HALF;  // 128 KiB
  auto kern = gemm256_kernel<EPI, DBG>;
  static const int wmap = getenv("CLIPX_GEMM_WMAP") ? atoi(getenv("CLIPX_GEMM_WMAP")) : 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.N, g.K, g.M / 256,
                     g.N / 256, wmap);
  return hipGetLastError();
}

hipError_t launch_gemm256(const GemmArgs& g, int n_cu, hipStream_t st) {
  if (g.M <= 0 || g.M % 256 != 0 || g.N % 256 != 0 || g.K % 128 != 0 || g.K <= 0) return hipErrorInvalidValue;
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;  // one workgroup per CU; multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  if (g.epi == EPI_BIAS_BF16) {
    const char* dbg = getenv("CLIPX_GEMM_DBG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) return launch_gemm256_epi<EPI_BIAS_BF16, 1>(g, grid, st);
    if (d == 2) return launch_gemm256_epi<EPI_BIAS_BF16, 2>(g, grid, st);
    if (d == 3) return launch_gemm256_epi<EPI_BIAS_BF16, 3>(g, grid, st);
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_gemm256_epi<EPI_BIAS_BF16>(g, grid, st);
    case EPI_BIAS_QGELU_BF16: return launch_gemm256_epi<EPI_BIAS_QGELU_BF16>(g, grid, st);
    case EPI_BIAS_GELU_BF16: return launch_gemm256_epi<EPI_BIAS_GELU_BF16>(g, grid, st);
    case EPI_BIAS_RESID_F32: return launch_gemm256_epi<EPI_BIAS_RESID_F32>(g, grid, st);
    case EPI_TABLE_F32: return launch_gemm256_epi<EPI_TABLE_F32>(g, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace clipx
