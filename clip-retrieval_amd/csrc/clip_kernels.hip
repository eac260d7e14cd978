// clip_kernels.hip -- gfx950 kernels of the CLIP encoder forward (encode half of the hot path).
//
// These replace the torch ops executed inside `model.encode_image` / `model.encode_text`, called by the
// reference at clip_retrieval/clip_inference/mapper.py:57,65 (and clip_back.py:230,244):
//   conv1 patch-embed, class/pos embedding, ln_pre, L x { LN, QKV, MHA, out_proj(+res), LN, fc1+GELU, fc2(+res) },
//   ln_post/ln_final, pooling, projection, L2 normalise, fp16 cast (mapper.py:58-59,66-67).
// Numerics: bf16 MFMA operands, fp32 accumulation; the residual stream, LayerNorm statistics and softmax
// stay in fp32 (SURVEY 7 "hard parts": 1e-3 cosine bar across 24-32 layers).
//
// HBM layouts
//   activations  x    f32  [B*T, d]   residual stream (row = b*T + t)
//                xn   bf16 [B*T, d]   LayerNorm output = GEMM operand
//                qkv  bf16 [B*T, 3d]  (q | k | v, each [H][64])
//                h    bf16 [B*T, mlp]
//   weights      W    bf16 [N, K]     torch nn.Linear layout ("B^T"): both GEMM operands are K-contiguous
//
// GEMM: out[m, n] = sum_k A[m,k] W[n,k].  128x128x64 tiles, 4 waves (2x2), v_mfma_f32_32x32x16_bf16 with the
// WEIGHT rows as the MFMA A operand and the activation rows as the B operand, so a lane's accumulator holds
// four consecutive n of one m: epilogue stores are 8 B (bf16) / 16 B (f32) instead of 2-byte scatters.
// LDS image per operand: [128 rows][8 chunks of 16 B], chunk position XOR ((row>>1)&7): ds_read_b128 fragment
// reads of 16 rows at one k-chunk hit 16 distinct 16-B slots of the 256-B bank row (conflict-free).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "clip_kernels.h"
#include "gemm_common.h"

namespace clipx {

// =============================================================================================
// GEMM
// =============================================================================================
constexpr int G_TM = 128, G_TN = 128, G_BK = 64;
constexpr int G_STAGE_BYTES = (G_TM + G_TN) * G_BK * 2;  // 32 KiB
constexpr int G_GROUP_M = 8;

// DEEP (GLDS only): 4-deep LDS ring (128 KiB, one workgroup per CU), the DMA of K-tile t+4 issued at K-tile t's barrier, ONE
// barrier per K-tile, software-pipelined fragment reads.  The launches this kernel serves have few workgroups (the peeled
// 257th m-tile of ViT-L/14: 16..64 workgroups; B = 1 queries), so nothing else runs on the CU to hide DMA or LDS latency.
template <int EPI, bool GLDS, bool DEEP = false, bool F16 = false>  // F16: operands are IEEE fp16 (bits travel as "bf16" pointers)
__global__ __launch_bounds__(256, (DEEP && !CLIPX_MFMA16) ? 1 : 2) void gemm_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ outp,
                                                          const float* __restrict__ table, int T, int M, int N,
                                                          int K, int row0, const float* __restrict__ rowscale,
                                                          bf16* __restrict__ out16, int kz) {
  // kz > 0 (EPI_RAW_F32, DEEP only): split-K -- workgroup column blockIdx.y multiplies K-tiles [y * kz, (y + 1) * kz) and
  // writes its raw accumulators to outp + y * M * N
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int wn = w & 1, wm = w >> 1;
  const int hb = lane >> 5, l31 = lane & 31;

  // ---- block -> tile: XCD-contiguous runs (block b executes on XCD b%8), grouped 8 m-tiles x all n-tiles
  const int ntm = (M + G_TM - 1) / G_TM, ntn = N / G_TN;
  const int nblk = ntm * ntn;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int per_group = G_GROUP_M * ntn;
  const int grp = logical / per_group, within = logical - grp * per_group;
  const int gm0 = grp * G_GROUP_M;
  const int gsz = (ntm - gm0) < G_GROUP_M ? (ntm - gm0) : G_GROUP_M;
  const int tm = gm0 + within % gsz, tn = within / gsz;
  const int m0 = tm * G_TM, n0 = tn * G_TN;

  // ---- staging map: wave w fills rows [32w, 32w+32) of both operand tiles, 8 rows (1 KiB) per instruction
  const bf16* gW[4];
  const bf16* gA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (w * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < M ? am : M - 1;
    gW[i] = W + (size_t)(n0 + row) * K + c * 8;
    gA[i] = A + (size_t)am * K + c * 8;
  }
  const int stage_off = (w * 4) * 1024;  // + i*1024 (+ lane*16)

  // ---- fragment read offsets (bytes inside an operand tile), one per k-step of the BK=64 slab
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = l31 * 128 + (((2 * kk + hb) ^ ((l31 >> 1) & 7)) << 4);
  // 16x16x32 form (CLIPX_MFMA16, gemm_common.h): row l15 of a 16-row half-block (+ 2 KiB per half), k-chunk 4 s + q4 of slab s
  const int l15 = lane & 15, q4 = lane >> 4;
  int foff16[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) foff16[sl] = l15 * 128 + (((4 * sl + q4) ^ ((l15 >> 1) & 7)) << 4);

#if CLIPX_MFMA16
  f32x4 acc[2][2][4];  // [weight block i][activation block j][quad g]: one 16 x 16 MFMA block each (gemm_common.h)
#define G_ACC(i, j, g, e) acc[i][j][g][e]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[i][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  f32x16 acc[2][2];
#define G_ACC(i, j, g, e) acc[i][j][4 * (g) + (e)]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#endif

  const int nk = kz > 0 ? kz : K / G_BK;
  const int t0 = kz > 0 ? (int)blockIdx.y * kz : 0;  // first K-tile of this workgroup
  if (EPI == EPI_RAW_F32) outp = reinterpret_cast<float*>(outp) + (size_t)blockIdx.y * M * N;

  auto compute = [&](int buf) {
    const unsigned char* sW = smem + buf * G_STAGE_BYTES + (wn * 64) * 128;
    const unsigned char* sA = smem + buf * G_STAGE_BYTES + G_TN * G_BK * 2 + (wm * 64) * 128;
#if CLIPX_MFMA16
    {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        frag_t wf[2][2], af[2][2];  // [32-row block][16-row half]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            wf[i][hf] = *reinterpret_cast<const frag_t*>(sW + (i * 32 + hf * 16) * 128 + foff16[sl]);
            af[i][hf] = *reinterpret_cast<const frag_t*>(sA + (i * 32 + hf * 16) * 128 + foff16[sl]);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mfma_block16<F16>(acc[i][j], wf[i][0], wf[i][1], af[j][0], af[j][1]);
      }
    }
#else
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 wf[2], af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sW + i * 32 * 128 + foff[kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sA + j * 32 * 128 + foff[kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_32x32x16<F16>(__builtin_bit_cast(frag_t, wf[i]), __builtin_bit_cast(frag_t, af[j]), acc[i][j]);
    }
#endif
  };

  if (GLDS) {
    auto issue = [&](int t, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned char* dW = smem + buf * G_STAGE_BYTES + stage_off + i * 1024;
        unsigned char* dA = dW + G_TN * G_BK * 2;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gW[i] + (size_t)t * G_BK), (lds_ptr_t)dW, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gA[i] + (size_t)t * G_BK), (lds_ptr_t)dA, 16, 0, 0);
      }
    };
    if (DEEP) {
      // Software-pipelined loop (same scheme as gemm256sp.hip): 4-slot LDS ring, K-tile t+4 is staged right behind the
      // barrier of K-tile t (the slot K-tile t just released), fragments of the next k-step are read (inline-asm
      // ds_read_b128, counted lgkmcnt) while this k-step's 4 MFMAs run, and the one barrier of the K-tile sits before
      // the last k-step's MFMAs with the next K-tile's first fragments read right behind it.  With one wave per SIMD
      // nothing else covers LDS latency: the previous loop (4 reads; lgkmcnt(0); 4 MFMAs, four times per K-tile) took
      // ~1 us per K-tile, this one ~0.35 us.  Needs M * K * 2 and N * K * 2 < 4 GiB (32-bit lane offsets).
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const int wu = __builtin_amdgcn_readfirstlane(w);
      const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
      const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds0);
      unsigned offW[4], offA[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        offW[i] = (unsigned)(reinterpret_cast<const char*>(gW[i]) - reinterpret_cast<const char*>(W));
        offA[i] = (unsigned)(reinterpret_cast<const char*>(gA[i]) - reinterpret_cast<const char*>(A));
      }
      // fragment read addresses of ring slot 0 (+ 32 KiB per slot): W rows wn*64 + i*32 + l31, A rows wm*64 + j*32 + l31
      unsigned fW[4], fA[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        fW[kk] = lds0 + (wn * 64) * 128 + foff[kk];
        fA[kk] = lds0 + G_TN * G_BK * 2 + (wm * 64) * 128 + foff[kk];
      }
      i32x4 F0[4], F1[4];  // [0..1] W fragments (i), [2..3] A fragments (j)
#define D_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#if CLIPX_MFMA16
      // 16x16x32 form: a K-tile is four units u = 2 * slab + (activation half): unit (sl, ha) multiplies the slab's four weight
      // fragments Wd[sl][2 i + j2] with the two activation fragments Ad[u & 1][j] of half ha -- 8 MFMAs of 16 cycles, the 128
      // matrix-pipe cycles of a 32x32x16 k-step; F0 / F1 are not used
      (void)F0; (void)F1;
      unsigned fW16[2], fA16[2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        fW16[sl] = lds0 + (wn * 64) * 128 + foff16[sl];
        fA16[sl] = lds0 + G_TN * G_BK * 2 + (wm * 64) * 128 + foff16[sl];
      }
      i32x4 Wd[2][4], Ad[2][2];
#define D_READ_W(sl, so)                                 \
  D_DSREAD(Wd[sl][0], fW16[sl] + (so), 0);               \
  D_DSREAD(Wd[sl][1], fW16[sl] + (so), 2048);            \
  D_DSREAD(Wd[sl][2], fW16[sl] + (so), 4096);            \
  D_DSREAD(Wd[sl][3], fW16[sl] + (so), 6144);
#define D_READ_A(u, so)                                                \
  D_DSREAD(Ad[(u) & 1][0], fA16[(u) >> 1] + (so), ((u) & 1) * 2048);     \
  D_DSREAD(Ad[(u) & 1][1], fA16[(u) >> 1] + (so), 4096 + ((u) & 1) * 2048);
#define D_READ_U0(so) D_READ_W(0, so) D_READ_A(0, so) __builtin_amdgcn_sched_barrier(0);
#define D_READ_U1(so) D_READ_A(1, so) __builtin_amdgcn_sched_barrier(0);
#define D_READ_U2(so) D_READ_W(1, so) D_READ_A(2, so) __builtin_amdgcn_sched_barrier(0);
#define D_READ_U3(so) D_READ_A(3, so) __builtin_amdgcn_sched_barrier(0);
#define D_WAIT_N(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0);
#define D_MFMA_U(u)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
      _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2)                                                               \
          acc[i][j][2 * ((u) & 1) + j2] =                                                                            \
              mfma_16x16x32<F16>(Wd[(u) >> 1][2 * i + j2], Ad[(u) & 1][j], acc[i][j][2 * ((u) & 1) + j2]);           \
  __builtin_amdgcn_sched_barrier(0);
#else
#define D_READ(F, so, kk)                          \
  D_DSREAD(F[0], fW[kk] + (so), 0);                \
  D_DSREAD(F[1], fW[kk] + (so), 4096);             \
  D_DSREAD(F[2], fA[kk] + (so), 0);                \
  D_DSREAD(F[3], fA[kk] + (so), 4096);             \
  __builtin_amdgcn_sched_barrier(0);
#define D_WAIT_PREV() asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
#define D_MFMA(F)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =            \
      mfma_32x32x16<F16>(F[i], F[2 + j], acc[i][j]);                                                                   \
  __builtin_amdgcn_sched_barrier(0);
#endif
      // one DMA: M0 = LDS address of the piece (SGPR), 32-bit lane offset, SGPR base of the K-tile
#define D_DMA(off, base, dst)                                                                                         \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory")
      auto stage = [&](int t, int slot) {
        const char* bw = reinterpret_cast<const char*>(W) + (size_t)(t0 + t) * (G_BK * 2);
        const char* ba = reinterpret_cast<const char*>(A) + (size_t)(t0 + t) * (G_BK * 2);
        const unsigned d = lds_base + slot * G_STAGE_BYTES + wu * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          D_DMA(offW[i], bw, d + i * 1024);
          D_DMA(offA[i], ba, d + G_TN * G_BK * 2 + i * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      const int npro = nk < 4 ? nk : 4;
      for (int t = 0; t < npro; ++t) stage(t, t);
      if (npro == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (npro == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (npro == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#if CLIPX_MFMA16
      D_READ_U0(0u)
      unsigned so = 0;  // byte offset of the current ring slot
#define D_KT_HEAD()                                   \
  D_READ_U1(so) D_WAIT_N(2) D_MFMA_U(0)                \
  D_READ_U2(so) D_WAIT_N(6) D_MFMA_U(1)                \
  D_READ_U3(so) D_WAIT_N(2) D_MFMA_U(2)                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0);
#define D_READ_FIRST(so_) D_READ_U0(so_)
#define D_MFMA_LAST() D_MFMA_U(3)
#else
      D_READ(F0, 0u, 0)
      unsigned so = 0;  // byte offset of the current ring slot
#define D_KT_HEAD()                                   \
  D_READ(F1, so, 1) D_WAIT_PREV() D_MFMA(F0)           \
  D_READ(F0, so, 2) D_WAIT_PREV() D_MFMA(F1)           \
  D_READ(F1, so, 3) D_WAIT_PREV() D_MFMA(F0)           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0);
#define D_READ_FIRST(so_) D_READ(F0, so_, 0)
#define D_MFMA_LAST() D_MFMA(F1)
#endif
#define D_SYNC(vm)                                           \
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(vm) : "memory");  \
  __builtin_amdgcn_sched_barrier(0);                         \
  __builtin_amdgcn_s_barrier();                              \
  __builtin_amdgcn_sched_barrier(0);
      int t = 0;
      // steady state: K-tiles t+1 .. t+3 are in flight at the sync (K-tile t+1 is needed: 16 younger DMA), K-tile t+4 exists
      for (; t + 4 < nk; ++t) {
        D_KT_HEAD()
        D_SYNC(16)
        stage(t + 4, t & 3);
        so = (so + G_STAGE_BYTES) & (4 * G_STAGE_BYTES - 1);
        D_READ_FIRST(so)
        D_MFMA_LAST()
      }
      // the last (up to) four K-tiles: nothing left to stage
      for (; t < nk; ++t) {
        const int rem = nk - 1 - t;  // K-tiles after this one (all issued)
        D_KT_HEAD()
        if (rem >= 3) { D_SYNC(16) } else if (rem == 2) { D_SYNC(8) } else { D_SYNC(0) }
        so = (so + G_STAGE_BYTES) & (4 * G_STAGE_BYTES - 1);
        if (rem > 0) { D_READ_FIRST(so) }
        D_MFMA_LAST()
      }
#undef D_DSREAD
#undef D_READ_FIRST
#undef D_MFMA_LAST
#undef D_DMA
#undef D_KT_HEAD
#undef D_SYNC
    } else {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
      compute(t & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    }
  } else {
    // register-staged fill (T14 split): global loads for tile t+1 are issued before the MFMAs of tile t
    // and written to LDS after them, so HBM/L2 latency hides under the compute
    uint4 rw0, rw1, rw2, rw3, ra0, ra1, ra2, ra3;
#define G_LOAD(t)                                                             \
  {                                                                           \
    rw0 = *reinterpret_cast<const uint4*>(gW[0] + (size_t)(t) * G_BK);        \
    rw1 = *reinterpret_cast<const uint4*>(gW[1] + (size_t)(t) * G_BK);        \
    rw2 = *reinterpret_cast<const uint4*>(gW[2] + (size_t)(t) * G_BK);        \
    rw3 = *reinterpret_cast<const uint4*>(gW[3] + (size_t)(t) * G_BK);        \
    ra0 = *reinterpret_cast<const uint4*>(gA[0] + (size_t)(t) * G_BK);        \
    ra1 = *reinterpret_cast<const uint4*>(gA[1] + (size_t)(t) * G_BK);        \
    ra2 = *reinterpret_cast<const uint4*>(gA[2] + (size_t)(t) * G_BK);        \
    ra3 = *reinterpret_cast<const uint4*>(gA[3] + (size_t)(t) * G_BK);        \
  }
#define G_STORE(buf)                                                                        \
  {                                                                                         \
    unsigned char* dW = smem + (buf) * G_STAGE_BYTES + stage_off + lane * 16;               \
    *reinterpret_cast<uint4*>(dW) = rw0;                                                    \
    *reinterpret_cast<uint4*>(dW + 1024) = rw1;                                             \
    *reinterpret_cast<uint4*>(dW + 2048) = rw2;                                             \
    *reinterpret_cast<uint4*>(dW + 3072) = rw3;                                             \
    *reinterpret_cast<uint4*>(dW + G_TN * G_BK * 2) = ra0;                                  \
    *reinterpret_cast<uint4*>(dW + G_TN * G_BK * 2 + 1024) = ra1;                           \
    *reinterpret_cast<uint4*>(dW + G_TN * G_BK * 2 + 2048) = ra2;                           \
    *reinterpret_cast<uint4*>(dW + G_TN * G_BK * 2 + 3072) = ra3;                           \
  }
    G_LOAD(0)
    G_STORE(0)
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) G_LOAD(t + 1)
      compute(t & 1);
      if (t + 1 < nk) G_STORE((t + 1) & 1)
      __syncthreads();
    }
#undef G_LOAD
#undef G_STORE
  }

  // ---- epilogue: quad g of each 32x32 sub-tile = four consecutive n of one m (gemm_common.h: quad_m / quad_n)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = m0 + wm * 64 + j * 32 + quad_m(g, lane);
        if (m >= M) continue;
        const int n = n0 + wn * 64 + i * 32 + quad_n(g, lane);
        const float4 v = make_float4(G_ACC(i, j, g, 0), G_ACC(i, j, g, 1), G_ACC(i, j, g, 2), G_ACC(i, j, g, 3));
        gemm_store_quad<EPI>(v, m, n, N, bias, outp, table, T, row0, rowscale, out16);
      }
    }
  }
}

template <int EPI, bool F16 = false>
static hipError_t launch_gemm_epi(const GemmArgs& g, hipStream_t st) {
  const int ntm = (g.M + G_TM - 1) / G_TM, ntn = g.N / G_TN;
  const dim3 grid(ntm * ntn), block(256);
  const size_t smem = 2 * G_STAGE_BYTES;
  const bool small_off = (size_t)g.M * g.K * 2 < ((size_t)1 << 32) && (size_t)g.N * g.K * 2 < ((size_t)1 << 32);
  if (g.variant == 1 && g.K >= 2 * G_BK && small_off) {
    auto kern = gemm_bf16_kernel<EPI, true, true, F16>;
    const size_t smem4 = 4 * G_STAGE_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, smem4, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.M, g.N, g.K, g.row0, g.rowscale, g.out16, 0);
  } else if (g.variant == 1) {
    auto kern = gemm_bf16_kernel<EPI, true, false, F16>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.M, g.N, g.K, g.row0, g.rowscale, g.out16, 0);
  } else {
    auto kern = gemm_bf16_kernel<EPI, false, false, F16>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, smem, st, g.A, g.W, g.bias, g.out, g.table, g.T, g.M, g.N, g.K, g.row0, g.rowscale, g.out16, 0);
  }
  return hipGetLastError();
}

// ---- split-K for small M (the B = 1 .. 2 query encodes of KnnService.compute_query, clip_back.py:207-255).  M = 257 rows
// give the 128x128 kernel 24 (N = 1024) .. 96 workgroups, each walking its whole K loop at ~0.5 us per K-tile with nothing
// else on the CU: fc2 (K = 4096) 36 us, out-proj 13 us.  Splitting K over S workgroup columns fills the chip; the S partial
// products (raw f32 accumulators) are summed in FIXED order 0 .. S-1 by a second kernel that applies the very epilogue code of
// the unsplit kernels (gemm_store_quad).  The result is deterministic but not bitwise equal to
// the unsplit sum, so the caller (clipx_api.hip) hands a scratch buffer only to the GEMMs of a single-sample API call: every
// call with >= 2 samples keeps the invariant "a row does not depend on the batch (or chunk) it travels in".
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N,
                                                           const float* __restrict__ bias, void* __restrict__ outp,
                                                           const float* __restrict__ rowscale, bf16* __restrict__ out16) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index
  const int nq4 = N >> 2;
  if (q >= (int64_t)M * nq4) return;
  const int m = (int)(q / nq4), n = (int)(q - (int64_t)m * nq4) * 4;
  float4 v = *reinterpret_cast<const float4*>(part + (size_t)m * N + n);
  for (int z = 1; z < S; ++z) {
    const float4 p = *reinterpret_cast<const float4*>(part + ((size_t)z * M + m) * N + n);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  gemm_store_quad<EPI>(v, m, n, N, bias, outp, nullptr, 1, 0, rowscale, out16);
}

// number of K splits for an [M, N, K] problem on n_cu compute units (1 = do not split)
static int splitk_factor(const GemmArgs& g) {
  if (!g.splitk_ws || g.M > 1024 || g.epi == EPI_TABLE_F32 || g.epi == EPI_RAW_F32) return 1;
  const int nk = g.K / G_BK;
  // Measured at M = 257 (round 2 kernel trace, no longer kept under profiles/): every launch has a floor of ~4.5 us, a split GEMM + its reduction
  // cost 9.9 + 5.0 us whatever K is, the unsplit kernel 12 us at K = 1024 and 36 us at K = 4096: only long K loops pay.
  // A single m-tile (the text query, M = 77: 6 .. 24 workgroups) gains from splitting shorter loops too (0.69 -> 0.62 ms).
  if (nk < 8 || (nk < 32 && g.M > G_TM)) return 1;
  const int cu = g.n_cu > 0 ? g.n_cu : 256;
  const int tiles = ((g.M + G_TM - 1) / G_TM) * (g.N / G_TN);
  int S = 1;
  // the largest S <= 16 that divides the K loop, leaves each workgroup >= 4 K-tiles, does not go past one workgroup per CU
  // more than needed and fits the scratch
  for (int c = 2; c <= 16; ++c) {
    if (nk % c != 0 || nk / c < 4) continue;
    if ((size_t)c * g.M * g.N * sizeof(float) > g.splitk_ws_bytes) break;
    if (tiles * (c - 1) >= cu) break;  // the previous split already filled the chip
    S = c;
  }
  return S;
}

template <int EPI>
static hipError_t launch_splitk_epi(const GemmArgs& g, int S, hipStream_t st) {
  const int ntm = (g.M + G_TM - 1) / G_TM, ntn = g.N / G_TN;
  auto kern = g.f16 ? gemm_bf16_kernel<EPI_RAW_F32, true, true, true> : gemm_bf16_kernel<EPI_RAW_F32, true, true, false>;
  const size_t smem4 = 4 * G_STAGE_BYTES;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, S), dim3(256), smem4, st, g.A, g.W, nullptr, g.splitk_ws, nullptr, 1, g.M, g.N, g.K, 0,
                     nullptr, nullptr, (g.K / G_BK) / S);
  const int64_t quads = (int64_t)g.M * (g.N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, g.splitk_ws, S, g.M, g.N,
                     g.bias, g.out, g.rowscale, g.out16);
  return hipGetLastError();
}

static hipError_t launch_gemm128(const GemmArgs& g, hipStream_t st) {
  const bool small_off = (size_t)g.M * g.K * 2 < ((size_t)1 << 32) && (size_t)g.N * g.K * 2 < ((size_t)1 << 32);
  const int S = (g.variant == 1 && small_off) ? splitk_factor(g) : 1;
  if (S > 1) {
    switch (g.epi) {
      case EPI_BIAS_BF16: return launch_splitk_epi<EPI_BIAS_BF16>(g, S, st);
      case EPI_BIAS_F16: return launch_splitk_epi<EPI_BIAS_F16>(g, S, st);
      case EPI_BIAS_QGELU_BF16: return launch_splitk_epi<EPI_BIAS_QGELU_BF16>(g, S, st);
      case EPI_BIAS_GELU_BF16: return launch_splitk_epi<EPI_BIAS_GELU_BF16>(g, S, st);
      case EPI_BIAS_RESID_F32: return launch_splitk_epi<EPI_BIAS_RESID_F32>(g, S, st);
      case EPI_BIAS_RESID_H16: return launch_splitk_epi<EPI_BIAS_RESID_H16>(g, S, st);
      default: break;
    }
  }
  if (g.f16) {  // fp16 operands exist for the bf16-output epilogues only (the LayerNorm-folded GEMMs)
    switch (g.epi) {
      case EPI_BIAS_BF16: return launch_gemm_epi<EPI_BIAS_BF16, true>(g, st);
      case EPI_BIAS_F16: return launch_gemm_epi<EPI_BIAS_F16, true>(g, st);
      case EPI_BIAS_QGELU_BF16: return launch_gemm_epi<EPI_BIAS_QGELU_BF16, true>(g, st);
      case EPI_BIAS_GELU_BF16: return launch_gemm_epi<EPI_BIAS_GELU_BF16, true>(g, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (g.epi) {
    case EPI_BIAS_BF16: return launch_gemm_epi<EPI_BIAS_BF16>(g, st);
    case EPI_BIAS_F16: return launch_gemm_epi<EPI_BIAS_F16>(g, st);
    case EPI_BIAS_QGELU_BF16: return launch_gemm_epi<EPI_BIAS_QGELU_BF16>(g, st);
    case EPI_BIAS_GELU_BF16: return launch_gemm_epi<EPI_BIAS_GELU_BF16>(g, st);
    case EPI_BIAS_RESID_F32: return launch_gemm_epi<EPI_BIAS_RESID_F32>(g, st);
    case EPI_BIAS_RESID_H16: return launch_gemm_epi<EPI_BIAS_RESID_H16>(g, st);
    case EPI_TABLE_F32: return launch_gemm_epi<EPI_TABLE_F32>(g, st);
    default: return hipErrorInvalidValue;
  }
}

// How many 256-row m-tiles go to the persistent 256x256 kernel (variant 3); the rest (ragged rows and the m-tiles
// that would only add a mostly-empty extra round over the CUs) goes to the 128x128 kernel.  ViT-L/14 at bs=256 has
// M = 257 * 256: 256 m-tiles fill the 256 CUs in whole rounds, the 257th (the class-token rows' worth) is peeled.
int gemm256_bulk_mtiles(int M, int N, int n_cu) {
  const int mt = M / 256, nt = N / 256;
  if (mt <= 0 || nt <= 0) return 0;
  const int cu = (n_cu > 0 ? n_cu : 256) & ~7;
  int best = mt;
  double best_cost = 1e30;
  for (int peel = 0; peel <= 8 && peel < mt; ++peel) {
    const int bulk = mt - peel;
    const int rounds = (bulk * nt + cu - 1) / cu;
    // cost in units of one 256x256 tile on one CU; peeled rows run as 128x128 tiles, two resident per CU, at
    // roughly 0.6x the per-CU rate of the big kernel
    const double rest = (double)(peel * 4 * nt) / (2.0 * cu);
    const double cost = rounds + (peel ? 0.25 * (rest > 1.0 ? rest : 1.0) / 0.6 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = bulk; }
  }
  return best;
}

// One ragged m-tile (256 rows) inside the persistent launch: 8 row blocks x (N / 32) column blocks of 32 x 32 are dealt to the
// workgroups as strips of NB consecutive column blocks; NB <= 4 (the strip's operands must fit the LDS ring), NB | N / 32, and
// every strip needs a workgroup.
int gemm256_tail_blocks(int N, int K, int n_cu) {
  int grid = (n_cu > 0 ? n_cu : 256) & ~7;
  if (grid < 8) grid = 8;
  if (N % 32 || K % 128 || (size_t)32 * K * 2 >= ((size_t)1 << 31)) return 0;
  const int cb = N / 32;
  for (int nb = 1; nb <= 4; ++nb)
    if (cb % nb == 0 && 8 * (cb / nb) <= grid) return nb;
  return 0;
}

hipError_t launch_gemm(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  if (g.M <= 0 || g.N % G_TN != 0 || g.K % G_BK != 0 || g.K <= 0) return hipErrorInvalidValue;
  const bool out16 = g.epi == EPI_BIAS_BF16 || g.epi == EPI_BIAS_QGELU_BF16 || g.epi == EPI_BIAS_GELU_BF16 || g.epi == EPI_BIAS_F16;
  if (out16 && !g.rowscale) return hipErrorInvalidValue;
  if (g.f16 && !out16) return hipErrorInvalidValue;
  if (g.stats_eps > 0.f) {
    // LayerNorm-folded GEMM that owns its row statistics: rowscale is the buffer a separate pass fills -- for the rows the 4-wave
    // kernel does not cover (it takes the sums from its own A fragments: gemm256w4.hip STATS).  Decided here, with the dispatch below.
    if (!g.f16 || !out16) return hipErrorInvalidValue;
    int fused_rows = 0;
    if (g.variant == 6 && g.N % 256 == 0 && g.K % 128 == 0 && g.M >= 256) {
      const int bulk = gemm256_bulk_mtiles(g.M, g.N, g.n_cu);
      const int cu = (g.n_cu > 0 ? g.n_cu : 256);
      GemmArgs b = g;
      b.M = bulk * 256;
      if (bulk > 0 && (int64_t)bulk * (g.N / 256) >= cu / 2 && gemm256w4_fuses_stats(b)) fused_rows = b.M;
    }
    if (fused_rows < g.M) {
      hipError_t e = launch_rowstats(reinterpret_cast<const char*>(g.A) + (size_t)fused_rows * g.K * 2, const_cast<float*>(g.rowscale) + fused_rows,
                                     g.M - fused_rows, g.K, g.stats_eps, st, 1, g.range_flag, 1);
      if (e != hipSuccess) return e;
    }
    if (fused_rows == 0) g.stats_eps = 0.f;  // (every row scale is in the buffer: a plain folded GEMM from here on)
  }
  if (g.variant >= 2 && g.N % 256 == 0 && g.K % 128 == 0 && g.M >= 256) {
    const int bulk = gemm256_bulk_mtiles(g.M, g.N, g.n_cu);
    // small problems (query-side B = 1: M = 257 or 77 rows) would put one 256x256 tile on each of a handful of CUs;
    // the 128x128 kernel gives them 4x the tiles.  Both kernels produce bit-identical rows.
    const int cu = (g.n_cu > 0 ? g.n_cu : 256);
    if (bulk > 0 && (int64_t)bulk * (g.N / 256) >= cu / 2) {
      GemmArgs b = g;
      b.M = bulk * 256;
      b.tail_m0 = b.tail_nb = 0;
      // exactly one m-tile left over (ViT-L/14 at bs 256: 257 m-tiles on 256 CUs): it rides in the same launch
      // (gemm256sp.hip: gemm256_tail) instead of a second, nearly empty one.  CLIPX_GEMM_VARIANT=4 keeps the separate launch (A/B).
      if (g.M - b.M == 256 && (g.variant == 3 || g.variant == 6) && g.row0 == 0) {
        b.tail_nb = gemm256_tail_blocks(g.N, g.K, g.n_cu);
        b.tail_m0 = b.M;
      }
      const bool tail_inside = b.tail_nb > 0;
      hipError_t e = (g.variant == 6 && gemm256w4_supports(b)) ? launch_gemm256w4(b, g.n_cu, st) : launch_gemm256sp(b, g.n_cu, st);
      if (e != hipSuccess) return e;
      if (b.M == g.M || tail_inside) return hipSuccess;
      GemmArgs r = g;  // remaining rows [bulk*256, M)
      r.variant = 1;
      r.stats_eps = 0.f;  // (their row scales were written above)
      r.splitk_ws = nullptr;  // rows of one large batch are computed the same way whichever kernel they land in
      r.A = g.A + (size_t)b.M * g.K;
      r.M = g.M - b.M;
      const size_t esz = (g.epi == EPI_BIAS_RESID_F32 || g.epi == EPI_TABLE_F32) ? 4 : 2;  // (EPI_BIAS_RESID_H16: fp16 rows)
      r.out = reinterpret_cast<char*>(g.out) + (size_t)b.M * g.N * esz;
      r.row0 = g.row0 + b.M;
      if (g.rowscale) r.rowscale = g.rowscale + b.M;
      if (g.out16) r.out16 = g.out16 + (size_t)b.M * g.N;
      return launch_gemm128(r, st);
    }
  }
  GemmArgs r = g;
  if (r.variant >= 2) r.variant = 1;
  r.stats_eps = 0.f;
  return launch_gemm128(r, st);
}

// =============================================================================================
// LayerNorm: one wave per row, fp32 statistics (two-pass in registers)
// =============================================================================================
template <int NV, int OUT>  // d = NV * 256; OUT 0: f32 (+ optional bf16 copy), 1: bf16, 2: IEEE fp16
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, void* __restrict__ y, int M,
                                                       float eps, bf16* __restrict__ y16) {
  constexpr int d = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    v[e] = xr[lane + 64 * e];
    s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / d);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const float a = v[e].x - mean, b = v[e].y - mean, c = v[e].z - mean, dd = v[e].w - mean;
    q += (a * a + b * b) + (c * c + dd * dd);
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.f / sqrtf(q * (1.f / d) + eps);
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int c4 = lane + 64 * e;
    const float4 g4 = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 b4 = reinterpret_cast<const float4*>(beta)[c4];
    float4 o;
    o.x = (v[e].x - mean) * rstd * g4.x + b4.x;
    o.y = (v[e].y - mean) * rstd * g4.y + b4.y;
    o.z = (v[e].z - mean) * rstd * g4.z + b4.z;
    o.w = (v[e].w - mean) * rstd * g4.w + b4.w;
    if (OUT == 1) {
      bf16x4 ob;
      ob[0] = (bf16)o.x; ob[1] = (bf16)o.y; ob[2] = (bf16)o.z; ob[3] = (bf16)o.w;
      reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(y) + (size_t)row * d)[c4] = ob;
    } else if (OUT == 2) {
      f16x4 oh;
      oh[0] = (_Float16)o.x; oh[1] = (_Float16)o.y; oh[2] = (_Float16)o.z; oh[3] = (_Float16)o.w;
      reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(y) + (size_t)row * d)[c4] = oh;
    } else {
      reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)row * d)[c4] = o;
      if (y16) {
        bf16x4 ob;
        ob[0] = (bf16)o.x; ob[1] = (bf16)o.y; ob[2] = (bf16)o.z; ob[3] = (bf16)o.w;
        reinterpret_cast<bf16x4*>(y16 + (size_t)row * d)[c4] = ob;
      }
    }
  }
}

// rstd of the bf16 shadow rows (see clip_kernels.h: launch_rowstats); d = NV * 512: a lane holds NV x 8 consecutive values
template <int NV2, bool F16>  // d = NV2 * 256 (NV2 x 4 values per lane); F16: rows are IEEE fp16, else bf16
__global__ __launch_bounds__(256) void rowstats_kernel(const void* __restrict__ x16, float* __restrict__ rstd, int M, float eps,
                                                       int* __restrict__ range_flag) {
  constexpr int d = NV2 * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const uint2* xr = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x16) + (size_t)row * d);
  float v[NV2][4];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < NV2; ++e) {
    const uint2 raw = xr[lane + 64 * e];
    if (F16) {
      const f16x4 h = __builtin_bit_cast(f16x4, raw);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[e][j] = (float)h[j];
    } else {
      const bf16x4 h = __builtin_bit_cast(bf16x4, raw);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[e][j] = (float)h[j];
    }
    s += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  // Range guard of the fp16 residual stream (clipx.h: CLIPX_E_RANGE).  The residual epilogues round f32 -> fp16 to nearest, so a
  // value beyond 65 504 is stored as inf; every state of the stream passes through this kernel (or tail_proj_kernel) before it
  // is used, and a row holding inf / NaN has a non-finite sum (d finite fp16 values cannot overflow an f32).  Free: one
  // compare on a value the kernel has anyway.
  if (range_flag && lane == 0 && !(fabsf(s) <= 3.0e38f)) atomicOr(range_flag, 1);
  const float mean = s * (1.f / d);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < NV2; ++e) {
    const float a = v[e][0] - mean, b = v[e][1] - mean, c = v[e][2] - mean, dd = v[e][3] - mean;
    q += (a * a + b * b) + (c * c + dd * dd);
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (lane == 0) rstd[row] = 1.f / sqrtf(q * (1.f / d) + eps);
}

// The fp16 stream: ONE pass in the canonical order (gemm_common.h: ln_rstd_onepass) -- four lanes per row (lane = r16 + 16 q4, the
// fragment layout of the 4-wave GEMM kernel, which computes the same sums for the rows it multiplies), lane part q4 takes the 16-B
// chunks q4, q4 + 4, ..; a wave covers 16 rows, a workgroup 64.
template <int NV2>  // d = NV2 * 256
__global__ __launch_bounds__(256) void rowstats_f16_kernel(const void* __restrict__ x16, float* __restrict__ rstd, int M, float eps,
                                                           int* __restrict__ range_flag) {
  constexpr int d = NV2 * 256, NCH = d / 8;
  const int lane = threadIdx.x & 63, q4 = lane >> 4, r16 = lane & 15;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + r16;
  const uint4* xr = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x16) + (size_t)(row < M ? row : M - 1) * d);
  const unsigned ones2 = __builtin_amdgcn_readfirstlane(0x3c003c00u);
  float s1 = 0.f, s2 = 0.f;
  constexpr int U = 8;  // chunks in flight per lane
  static_assert((NCH / 4) % U == 0, "row length");
  for (int j0 = 0; j0 < NCH / 4; j0 += U) {
    uint4 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = xr[q4 + 4 * (j0 + u)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CLIPX_DOT2_SQ(s2, c[u].x); CLIPX_DOT2_SQ(s2, c[u].y); CLIPX_DOT2_SQ(s2, c[u].z); CLIPX_DOT2_SQ(s2, c[u].w);
      CLIPX_DOT2_SUM(s1, c[u].x, ones2); CLIPX_DOT2_SUM(s1, c[u].y, ones2); CLIPX_DOT2_SUM(s1, c[u].z, ones2); CLIPX_DOT2_SUM(s1, c[u].w, ones2);
    }
  }
  s1 += __shfl_xor(s1, 16);
  s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (row >= M || q4 != 0) return;
  // range guard of the fp16 residual stream (see rowstats_kernel): a row holding inf / NaN has a non-finite sum
  if (range_flag && !(fabsf(s1) <= 3.0e38f)) atomicOr(range_flag, 1);
  rstd[row] = ln_rstd_onepass(s1, s2, 1.f / (float)d, eps);
}

hipError_t launch_rowstats(const void* x16, float* rstd, int M, int d, float eps, hipStream_t st, int f16, int* range_flag, int canonical) {
  if (M <= 0) return hipSuccess;
  if (f16 && canonical) {
    const dim3 grid((M + 63) / 64), block(256);
#define RSH_CASE(NV)                                                                                           \
  case NV * 256:                                                                                               \
    hipLaunchKernelGGL((rowstats_f16_kernel<NV>), grid, block, 0, st, x16, rstd, M, eps, range_flag);          \
    break;
    switch (d) {
      RSH_CASE(1) RSH_CASE(2) RSH_CASE(3) RSH_CASE(4) RSH_CASE(5) RSH_CASE(6) RSH_CASE(7) RSH_CASE(8)
      default: return hipErrorInvalidValue;
    }
#undef RSH_CASE
    return hipGetLastError();
  }
  const dim3 grid((M + 3) / 4), block(256);
#define RS_CASE(NV)                                                                                        \
  case NV * 256:                                                                                           \
    if (f16) hipLaunchKernelGGL((rowstats_kernel<NV, true>), grid, block, 0, st, x16, rstd, M, eps, range_flag);       \
    else hipLaunchKernelGGL((rowstats_kernel<NV, false>), grid, block, 0, st, x16, rstd, M, eps, range_flag);          \
    break;
  switch (d) {
    RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef RS_CASE
  return hipGetLastError();
}

// one workgroup per output row n of W [N, K]
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ bias,
                                                            bf16* __restrict__ Wf, float* __restrict__ cf, int K, int f16) {
  __shared__ float red[8];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* w = W + (size_t)n * K;
  float sg = 0.f, sb = 0.f;
  for (int k = tid; k < K; k += 256) {
    sg += w[k] * gamma[k];
    sb += w[k] * beta[k];
  }
  for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); }
  if ((tid & 63) == 0) { red[tid >> 6] = sg; red[4 + (tid >> 6)] = sb; }
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)K;
  for (int k = tid; k < K; k += 256) {
    const float v = w[k] * gamma[k] - mean;
    if (f16) reinterpret_cast<_Float16*>(Wf)[(size_t)n * K + k] = (_Float16)v;
    else Wf[(size_t)n * K + k] = (bf16)v;
  }
  if (tid == 0) cf[n] = bias[n] + (red[4] + red[5] + red[6] + red[7]);
}
hipError_t launch_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, bf16* Wf, float* cf,
                                 int N, int K, hipStream_t st, int f16) {
  hipLaunchKernelGGL(fold_layernorm_kernel, dim3(N), dim3(256), 0, st, W, gamma, beta, bias, Wf, cf, K, f16);
  return hipGetLastError();
}
__global__ void fill_f32_kernel(float* __restrict__ p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t launch_fill_f32(float* p, float v, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(256), dim3(256), 0, st, p, v, n);
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* y, int out_kind, int M, int d,
                            float eps, hipStream_t st, bf16* y16) {
  if (M <= 0) return hipSuccess;
  const dim3 grid((M + 3) / 4), block(256);
#define LN_CASE(NV)                                                                                           \
  case NV * 256:                                                                                              \
    if (out_kind == 1) hipLaunchKernelGGL((layernorm_kernel<NV, 1>), grid, block, 0, st, x, gamma, beta, y, M, eps, (bf16*)nullptr); \
    else if (out_kind == 2) hipLaunchKernelGGL((layernorm_kernel<NV, 2>), grid, block, 0, st, x, gamma, beta, y, M, eps, (bf16*)nullptr); \
    else hipLaunchKernelGGL((layernorm_kernel<NV, 0>), grid, block, 0, st, x, gamma, beta, y, M, eps, y16);    \
    break;
  switch (d) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef LN_CASE
  return hipGetLastError();
}

// =============================================================================================
// im2col: pixels -> bf16 patch rows (+ the all-zero class-token row), 8 k per thread (16-B stores)
// =============================================================================================
template <int FMT>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ pixels, int B, int S, int P, int Kp,
                                                    float m0, float m1, float m2, float i0, float i1, float i2,
                                                    bf16* __restrict__ out) {
  const int gdim = S / P, T = gdim * gdim + 1, PP = P * P, K = 3 * PP, nch = Kp / 8;
  const int64_t total = (int64_t)B * T * nch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % nch);
    const int64_t row = idx / nch;
    const int t = (int)(row % T), b = (int)(row / T);
    bf16x8 o;
    // the eight loads are unconditional (coordinates clamped, the value discarded afterwards): behind a per-element
    // `if (t > 0 && k < K)` hipcc put every load in its own branch with an s_waitcnt vmcnt(0) -- eight serialised round trips
    const int tp = t > 0 ? t - 1 : 0;
    const int py = tp / gdim, px = tp - py * gdim;
    float raw[8];
    int cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ch * 8 + e;
      const int kk = k < K ? k : K - 1;
      const int c = kk / PP, rem = kk - c * PP;
      const int iy = rem / P, ix = rem - iy * P;
      const int yy = py * P + iy, xx = px * P + ix;
      cc[e] = c;
      if (FMT == 0) raw[e] = reinterpret_cast<const float*>(pixels)[(((size_t)b * 3 + c) * S + yy) * S + xx];
      else raw[e] = (float)reinterpret_cast<const unsigned char*>(pixels)[(((size_t)b * S + yy) * S + xx) * 3 + c];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = raw[e];
      if (FMT != 0) {
        const int c = cc[e];
        // torchvision's arithmetic, operation for operation (ToTensor: x / 255; Normalize: (x - mean) / std, IEEE f32 divisions):
        // the f32 value is the reference reader's `image_tensor` element bit for bit (pinned against the reference-held
        // tests/test_clip_inference/test_tensors/*.pkl), so raw uint8 pixels and the reader's f32 tensor give the same patches
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? i0 : (c == 1 ? i1 : i2);
        v = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), mean), sd);
      }
      o[e] = (t > 0 && ch * 8 + e < K) ? (bf16)v : (bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(out + (size_t)row * Kp + ch * 8) = o;
  }
}

hipError_t launch_im2col(const void* pixels, int fmt, int B, int S, int P, int Kp, const float* mean,
                         const float* stdv, bf16* out, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int gdim = S / P, T = gdim * gdim + 1;
  const int64_t total = (int64_t)B * T * (Kp / 8);
  const int blocks = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  if (fmt == 0)
    hipLaunchKernelGGL(im2col_kernel<0>, dim3(blocks), dim3(256), 0, st, pixels, B, S, P, Kp, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, out);
  else
    hipLaunchKernelGGL(im2col_kernel<1>, dim3(blocks), dim3(256), 0, st, pixels, B, S, P, Kp, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2], out);
  return hipGetLastError();
}

// =============================================================================================
// Attention (head dim DH = 64 or 80): one workgroup per (batch, head); K (row-major) and V^T in LDS.
//   S^T = K Q^T   (MFMA A = K rows, B = Q rows): a lane holds 16 keys x NKB blocks of ONE query column
//   softmax over the lane's registers + one exchange with lane^32; P stays in registers as the B operand
//   O^T = V^T P^T (MFMA A = V^T rows from LDS, B = P): a lane ends with 4 consecutive d of its query -> 8-B stores
// The MFMA contraction index of the PV product is a permutation of the key index (the order in which the
// S^T accumulator registers hold keys); V^T fragments are read with the same permutation, so no lane
// exchange is needed between the two products.
// LDS images: K rows of DH*2 bytes; DH=64: 128-B rows with the 16-B chunk position XOR ((key>>1)&7); DH=80 (ViT-H/14):
// rows padded to 176 B = 11 chunks, 11 is odd so 16 consecutive rows hit 16 distinct 16-B slots without a swizzle.
// V^T has DV = DH rounded up to 32 rows (rows >= DH are zero: the third 32-row output block of DH=80 is half padding).
// RECOMP: S^T blocks are computed twice (pass 1: row max only, pass 2: exp + PV) instead of being kept in NKB*16
// registers, for configurations where one query block per wave and many waves per workgroup pay.
// =============================================================================================
// TIMER (CLIPX_ATTN_DBG=9, tools/attn_bench): per-wave shader-cycle totals of staging / S + max / exp + PV / store
// Operand type of the attention products (round 4): q, k, v arrive as IEEE fp16 (the QKV projection's EPI_BIAS_F16 epilogue) and
// the probabilities P are rounded to fp16 as well: v_mfma_f32_32x32x16_f16 issues at the rate of the bf16 form, and the three extra
// mantissa bits of q and k are worth an order of magnitude in the embedding's error where LayerNorm gains are large (the
// logits q.k are where the bf16 rounding hurt most: tools/emulate_fp16_stream.py).  The 16-bit values keep travelling
// through `bf16`-typed pointers and fragments: every load, LDS staging step and transposition moves bits.
__device__ __forceinline__ f32x16 attn_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned attn_pack_p(float p0, float p1) {  // one v_cvt_pk_f16_f32 (round to nearest even)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){p0, p1}, f16x2_t));
}

__device__ long long g_attn_phase[8192 * 4];
#ifdef CLIPX_ABLATE
__device__ int g_attn_pk_timer = 0;  // set by launch_attention_pk9 from CLIPX_ATTN_PK_TIMER (tools build)
#endif
template <int DH, int NKB, int NW, int QPW, bool CAUSAL, bool RECOMP, bool TIMER = false>
__global__ __launch_bounds__(NW * 64, (2 * NW + 3) / 4) void attention_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int Tin,
                                                              int H, float scale_log2e, int dbg, int q_blocks,
                                                              const int* __restrict__ offs, const int* __restrict__ lens) {
  constexpr int TP = NKB * 32;
  constexpr int CH = DH / 8;                      // 16-B chunks per key row
  constexpr int KS = DH / 16;                     // MFMA k-steps of the QK^T product
  constexpr int KROW = DH == 64 ? 128 : 176;      // bytes per K row in LDS
  constexpr int DV = (DH + 31) / 32 * 32, NB = DV / 32;
  constexpr int VT_STRIDE = TP * 2 + 8;  // bytes per V^T row: odd multiple of 8 -> conflict-free ds_read_b64
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                 // [TP][KROW]
  unsigned char* sVt = smem + TP * KROW;    // [DV][VT_STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hb = lane >> 5, l31 = lane & 31;
  long long tph[4] = {0, 0, 0, 0};
  long long tst = TIMER ? (long long)__builtin_readcyclecounter() : 0;
#define A_STAMP(i) if (TIMER) { const long long n_ = (long long)__builtin_readcyclecounter(); tph[i] += n_ - tst; tst = n_; }
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ld = 3 * H * DH;  // qkv row stride (elements)
  // ragged batches (the causal text tower: rows past a caption's EOT are never read): sample b owns rows offs[b] .. + lens[b]
  const int T = lens ? lens[b] : Tin;
  const size_t row0 = offs ? (size_t)offs[b] : (size_t)b * Tin;
  const bf16* qbase = qkv + row0 * ld + h * DH;
  const bf16* kbase = qbase + H * DH;
  const bf16* vbase = qbase + 2 * H * DH;
  auto kchunk = [](int key, int c) -> int { return DH == 64 ? (c ^ ((key >> 1) & 7)) : c; };

  // ---- Q fragments of every query block of this wave, requested before the K/V staging so that their HBM
  // latency hides under it (B operand: lane (q = l31, hb) holds Q[q][16s + 8hb .. +8])
  bf16x8 qf_all[QPW][KS];
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    const int qpos = (qi * NW + w) * 32 + l31;  // query blocks are dealt round-robin to the waves
    const int qrow = qpos < T ? qpos : T - 1;  // padded query rows compute on a valid row and are never stored
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 v = *reinterpret_cast<const uint4*>(qbase + (size_t)qrow * ld + 16 * s + 8 * hb);
      qf_all[qi][s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  }

  if (dbg != 2) {  // ablation 2: no staging (garbage results)
  // ---- stage K: CH lanes cover one key's row.  Loads are unconditional (row clamped, zeroed after) and issued as
  // one batch: a per-element `if (key < T) load` makes hipcc branch around every load and drain vmcnt(0) each time.
  {
    constexpr int KIT = (TP * CH + NW * 64 - 1) / (NW * 64);
    uint4 kv[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int i = tid + it * NW * 64;
      int key = i / CH;
      const int c = i - key * CH;
      key = key < T ? key : T - 1;
      kv[it] = *reinterpret_cast<const uint4*>(kbase + (size_t)key * ld + c * 8);
    }
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int i = tid + it * NW * 64;
      const int key = i / CH, c = i - key * CH;
      if (i < TP * CH) {
        const uint4 v = key < T ? kv[it] : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(sK + key * KROW + (kchunk(key, c) << 4)) = v;
      }
    }
  }
  // ---- stage V transposed: a thread takes keys (2kp, 2kp+1) x 8 d and writes 8 packed key-pairs
  {
    constexpr int VIT = ((TP / 2) * CH + NW * 64 - 1) / (NW * 64);
    uint4 v0s[VIT], v1s[VIT];
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int i = tid + it * NW * 64;
      const int kp = i / CH, c = i - kp * CH;
      const int k0 = 2 * kp < T ? 2 * kp : T - 1, k1 = 2 * kp + 1 < T ? 2 * kp + 1 : T - 1;
      v0s[it] = *reinterpret_cast<const uint4*>(vbase + (size_t)k0 * ld + c * 8);
      v1s[it] = *reinterpret_cast<const uint4*>(vbase + (size_t)k1 * ld + c * 8);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int i = tid + it * NW * 64;
      const int kp = i / CH, c = i - kp * CH;
      if (i < (TP / 2) * CH) {
        const uint4 v0 = 2 * kp < T ? v0s[it] : make_uint4(0u, 0u, 0u, 0u);
        const uint4 v1 = 2 * kp + 1 < T ? v1s[it] : make_uint4(0u, 0u, 0u, 0u);
        const unsigned a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const unsigned lo = (a0[jj] & 0xffffu) | (a1[jj] << 16);         // d = 8c + 2jj
          const unsigned hi = (a0[jj] >> 16) | (a1[jj] & 0xffff0000u);     // d = 8c + 2jj + 1
          *reinterpret_cast<unsigned*>(sVt + (8 * c + 2 * jj) * VT_STRIDE + kp * 4) = lo;
          *reinterpret_cast<unsigned*>(sVt + (8 * c + 2 * jj + 1) * VT_STRIDE + kp * 4) = hi;
        }
      }
    }
    if (DV > DH) {  // zero rows DH .. DV-1 of V^T
      for (int i = tid; i < (DV - DH) * (TP / 2); i += NW * 64) {
        const int r = i / (TP / 2), kp = i - r * (TP / 2);
        *reinterpret_cast<unsigned*>(sVt + (DH + r) * VT_STRIDE + kp * 4) = 0u;
      }
    }
  }
  }
  __syncthreads();
  A_STAMP(0)
  if (dbg == 1) return;  // ablation (CLIPX_ATTN_DBG=1): staging only

  const int ksw = (l31 >> 1) & 7;
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    const int qb = qi * NW + w;
    if (qb >= q_blocks) break;  // q_blocks = NKB, or 1 when only the rows of query block 0 are read afterwards (last block, token 0)
    const int qpos = qb * 32 + l31;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qf_all[qi][s];
    // ---- S^T blocks; causal rows never look right of the diagonal block (wave-uniform skip)
    auto s_block = [&](int kb) -> f32x16 {
      f32x16 sb;
#pragma unroll
      for (int r = 0; r < 16; ++r) sb[r] = 0.f;
      if (!CAUSAL || kb <= qb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int c = 2 * s + hb;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kb * 32 + l31) * KROW + ((DH == 64 ? (c ^ ksw) : c) << 4));
          sb = attn_mfma(kf, qf[s], sb);
        }
      }
      // masking is needed only in the last key block (padding past T) and, for causal, on/after the diagonal
      if (kb == NKB - 1 || CAUSAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
          const bool ok = key < T && (!CAUSAL || key <= qpos);
          sb[r] = ok ? sb[r] : -INFINITY;
        }
      }
      return sb;
    };
    f32x16 sacc[RECOMP ? 1 : NKB];
    float mx = -INFINITY;
#pragma unroll(RECOMP ? 1 : NKB)
    for (int kb = 0; kb < NKB; ++kb) {
      const f32x16 sb = s_block(kb);
      if (!RECOMP) sacc[kb] = sb;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sb[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (mx == -INFINITY) mx = 0.f;  // padded query rows
    const float nmx = -mx * scale_log2e;
    if (TIMER) { asm volatile("" : "+v"(mx)); A_STAMP(1) }

    // ---- per key block: P = exp2(S*c - m*c) -> bf16 (stays in registers as the B operand), then O^T += V^T P^T
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    f32x2_t sum2 = {0.f, 0.f};
    const int tail_keys = T - (NKB - 1) * 32;  // keys of the last key block that exist
    f32x16 oacc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[nb][r] = 0.f;
#pragma unroll(RECOMP ? 1 : NKB)
    for (int kb = 0; kb < NKB; ++kb) {
      if (CAUSAL && kb > qb) continue;
      bf16x8 pf[2];
      const f32x16 sb = RECOMP ? s_block(kb) : sacc[RECOMP ? 0 : kb];
      // pairs go through one v_cvt_pk_bf16_f32 (element-wise casts make hipcc convert singly and re-pack with v_perm)
      unsigned pw[8];
      // the scale-and-shift and the row sum run two values per instruction (v_pk_fma_f32 / v_pk_add_f32); in the last key
      // block only the first `tail_keys` keys exist (ViT-L/14: 1 of 32): when they all sit in the first register quad
      // the other 12 exponentials of the block are skipped (their P is exactly 0)
      const bool short_tail = !CAUSAL && kb == NKB - 1 && tail_keys <= 4;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if (short_tail && r >= 4) {
          pw[r >> 1] = 0u;
          continue;
        }
        const f32x2_t e = (f32x2_t){sb[r], sb[r + 1]} * (f32x2_t){scale_log2e, scale_log2e} + (f32x2_t){nmx, nmx};
        const f32x2_t pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        sum2 += pp;
        pw[r >> 1] = attn_pack_p(pp[0], pp[1]);
      }
      {
        const uint4 w0 = make_uint4(pw[0], pw[1], pw[2], pw[3]), w1 = make_uint4(pw[4], pw[5], pw[6], pw[7]);
        pf[0] = *reinterpret_cast<const bf16x8*>(&w0);
        pf[1] = *reinterpret_cast<const bf16x8*>(&w1);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          // lane (d = 32nb + l31, hb): keys kb*32 + 16*s2 + 4hb + {0..3} and + 8 + {0..3}
          const unsigned char* vp = sVt + (32 * nb + l31) * VT_STRIDE + (kb * 32 + 16 * s2 + 4 * hb) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(vp);
          const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
          uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
          oacc[nb] = attn_mfma(*reinterpret_cast<bf16x8*>(&vv), pf[s2], oacc[nb]);
        }
    }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    if (TIMER) { asm volatile("" : "+v"(oacc[0]), "+v"(oacc[NB - 1])); A_STAMP(2) }
    // ---- store: lane owns query qpos, d = 32nb + 8g + 4hb + {0..3}
    if (qpos < T) {
      bf16* orow = out + (row0 + qpos) * (H * DH) + h * DH;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (32 * nb + 8 * g >= DH) continue;  // padded d of the last block (compile time)
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(oacc[nb][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(orow + 32 * nb + 8 * g + 4 * hb) = o;
        }
    }
    A_STAMP(3)
  }
  if (TIMER && lane == 0 && blockIdx.x * NW + w < 8192) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g_attn_phase[(blockIdx.x * NW + w) * 4 + i] = tph[i];
  }
#undef A_STAMP
}

// =============================================================================================
// Persistent attention for the long image sequences (head dim 64, not causal, NKB = 9: ViT-L/14's T = 257).
// One workgroup of 6 waves per CU walks the (batch, head) pairs; K and V of the NEXT pair arrive by LDS-DMA into the other
// half of the LDS (2 x 72 KiB) while the current pair is computed, so the staging -- 39 % of a wave's time per head in the
// one-head-per-workgroup kernel above (24 register loads, 12 ds_write_b128 and 96 transposing ds_write_b32 per thread), which
// can only overlap it across the two workgroups a CU holds -- needs no registers, no LDS stores and no transposition pass:
//   * K image as above (128-B rows, 16-B chunk position XOR ((key >> 1) & 7)), written by global_load_lds_dwordx4 with the
//     swizzle folded into each lane's SOURCE address (the DMA writes lane-linear);
//   * V stays row-major: two [TP keys][32 d] images (64-B rows), and the A fragments of O^T = V^T P^T -- 4 consecutive keys of
//     ONE column per lane -- come out of ds_read_b64_tr_b16, the gfx950 transposing LDS read (a 16-lane group reads a
//     [4 keys][16 d] tile: lane i supplies the address of row i >> 2, piece 4 (i & 3), and receives column i;
//     tools/tr_probe.hip).  With 64-B rows the 32 lanes of a read group touch 32 distinct 8-B slots of the 256-B bank row.
// Query blocks are dealt w, w + 6: waves 0 - 2 take two, waves 3 - 5 one (9 blocks on 6 waves) -- and waves 3 - 5 issue all of
// the DMAs (an LDS-DMA issue holds its wave for ~100 cycles; spread over all six waves the kernel was no faster than the one
// above, issued by the waves with time to spare it is 6 % faster: 174 vs 186 us at B H = 4096, same box, alternating).  The
// S^T / softmax / PV arithmetic is the kernel's above, operation for operation: the outputs are the same bits.
// (Also built and measured: two 3-wave workgroups per CU, one LDS image each, K of the next pair requested as soon as the last
// S phase is over and V at the top of its own pair: every wave issues 24 DMAs per pair -- 212 us.
// the patch is in the repository's history, round 3)
// =============================================================================================
#ifndef CLIPX_ATTN_SPIPE
#define CLIPX_ATTN_SPIPE 1
#endif
#ifndef CLIPX_ATTN_ROLES
#define CLIPX_ATTN_ROLES 1
#endif
// NWT = 8 (round 4; T = 32 (NKB - 1) + 1 only, ViT-L/14's 257 = 8 x 32 + 1): EIGHT waves, two on every SIMD, one full query block
// each -- the 6-wave dealing leaves one SIMD with three blocks -- and the single query of the ragged last block, which costs a
// wave a whole block pass above, split along the KEYS: wave w attends it to key block w (wave 0 also to the one-key block 8), leaves
// (max, sum, 64 partial outputs) in a small LDS table, and wave 0 merges the nine partials (rescaled by 2^((m_j - m) c)) behind the
// next pair's barrier.  All eight waves issue DMAs (9 of the pair's 72 pieces each).  The 256 full-block rows are computed exactly as
// by the 6-wave kernel (same bits); the last token's row sums its softmax in nine pieces (f32) and may differ in the last bit before
// the bf16 rounding.  MEASURED AND NOT USED: 188 us against 162 for six waves on the same box (profiles/r04x_attention_8wave.log;
// round 3's version of the idea, before the phases were pipelined: 190 against 186) -- a SIMD's VALU is what two waves share, the six-
// wave dealing already keeps it ~85 % busy on the SIMDs that hold two, and this form adds the tail's and eight DMA issuers' VALU work.
// Instantiated in the tools build only (CLIPX_ATTN_CFG=13).
template <int NKB, int NWT = 6>
__global__ __launch_bounds__(NWT * 64, 1) void attention_pk_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int T, int H,
                                                                  int nheads, float scale_log2e, int q_blocks) {
  constexpr bool EIGHT = NWT == 8;
  constexpr int DH = 64, KS = 4, NW = NWT, NDW = EIGHT ? 8 : (CLIPX_ATTN_ROLES ? 2 : 3), QPW = (NKB + NW - 1) / NW, TP = NKB * 32,
                KROW = 128, NB = 2;
  static_assert(NWT == 6 || NWT == 8, "6 waves (any T of NKB blocks) or 8 (T = 32 (NKB - 1) + 1)");
  static_assert(!CLIPX_ATTN_ROLES || NKB == 9, "the role table below is the one of 9 query blocks");
  static_assert(!EIGHT || (NKB == NW + 1 && QPW == 2), "eight full query blocks + the one-query block");
  constexpr int KBYTES = TP * KROW, VHALF = TP * 64, BUF = KBYTES + 2 * VHALF;
  constexpr int KDMA = TP * 8 / 64 / NDW, VDMA = 2 * TP * 4 / 64 / NDW;
  static_assert(EIGHT || TP * 8 % (64 * NDW) == 0, "the DMA pieces must divide evenly among the issuing waves");
  constexpr int NPIECE = BUF / 1024, PPW = NPIECE / 8, KPIECE = KBYTES / 1024;  // EIGHT: 72 DMA pieces of 1 KiB per pair, 9 per wave
  static_assert(!EIGHT || NPIECE % 8 == 0, "the DMA pieces must divide evenly among the eight waves");
  constexpr int MRG = 68;  // EIGHT: floats per partial of the last query: max, sum, 64 outputs (+ pad); two tables (pair parity)
  const bool tail_on = EIGHT && q_blocks >= NKB;  // (the pooled last block asks for query block 0 only)
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hb = lane >> 5, l31 = lane & 31;
  const int ld = 3 * H * DH;  // qkv row stride (elements)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
#ifdef CLIPX_ABLATE
  // tools build, CLIPX_ATTN_PK_TIMER=1 (tools/attn_bench): shader cycles per wave in [0] the wait + barrier at the top of a pair,
  // [1] the DMA issue, [2] the S phases, [3] exp + P V + output of its blocks -> g_attn_phase[(workgroup * 6 + wave) * 4 + i]
  const bool timer = g_attn_pk_timer != 0;
  long long tph[4] = {0, 0, 0, 0};
  long long tst = timer ? (long long)__builtin_readcyclecounter() : 0;
#define P_STAMP(i) if (timer) { const long long n_ = (long long)__builtin_readcyclecounter(); tph[i] += n_ - tst; tst = n_; }
#else
#define P_STAMP(i)
#endif

  // ---- who does what (CLIPX_ATTN_ROLES, round 4).  Waves w and w + 4 share a SIMD (0 & 4, 1 & 5); waves 2 and 3 have theirs alone.
  // The phase timer (tools build, profiles/r04t_attention_phases.log) showed the original dealing -- blocks w and w + 6, the DMAs
  // on waves 3 - 5 -- as 3 blocks + a DMA share on each shared SIMD (saturated: a block costs 7.7 - 12 k cycles there against 6.8 k
  // alone), 2 blocks on wave 2's and 1 block + a DMA share on wave 3's (idle half of the pair).  New dealing: the lone waves take
  // two blocks each (2: 2, 7; 3: 3, 8), wave 0 two (0, 6) beside wave 4's one, and waves 1 and 5 one block each plus all the DMAs.
  auto blk_of = [&](int qi) -> int {  // query block qi of this wave (-1: none)
    if (EIGHT) return qi == 0 ? w : -1;
#if CLIPX_ATTN_ROLES
    if (qi == 0) return w;
    return w == 0 ? 6 : (w == 2 ? 7 : (w == 3 ? 8 : -1));
#else
    return qi * NW + w < NKB ? qi * NW + w : -1;
#endif
  };
#if CLIPX_ATTN_ROLES
  const bool dma_wave = EIGHT || w == 1 || w == 5;
  const int dma_idx = EIGHT ? w : (w == 1 ? 0 : 1);
#else
  const bool dma_wave = EIGHT || w >= NW - NDW;
  const int dma_idx = EIGHT ? w : w - (NW - NDW);
#endif
  float* mrg = reinterpret_cast<float*>(smem + 2 * BUF);  // EIGHT: [2][NKB][MRG]

  // ---- DMA of one pair into LDS half `buf`, by waves NW - NDW .. NW - 1.  Per-lane source offsets (bytes from the pair's q
  // base) are recomputed per piece -- a handful of VALU on waves that have the time.  The lane id goes through an opaque asm per
  // call: the offsets are loop-invariant, and hoisted out of the pair loop they would occupy 24 registers for the whole kernel.
  auto issue = [&](int hd, int buf) {
    if (!dma_wave) return;  // wave-uniform
    const int ww = dma_idx;
    const int b = hd / H, h = hd - b * H;
    const char* base = reinterpret_cast<const char*>(qkv + (size_t)b * T * ld + h * DH);
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    if constexpr (EIGHT) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int piece = ww * PPW + j;  // wave-uniform: pieces 0 .. KPIECE - 1 are the K image, the rest the two V images
        unsigned off;
        if (piece < KPIECE) {
          const int c = piece * 64 + lane_v;
          const int key = c >> 3, pos = c & 7;
          const int kk = key < T ? key : T - 1;
          off = (unsigned)((kk * ld + H * DH + ((pos ^ ((key >> 1) & 7)) << 3)) * 2);
        } else {
          const int c = (piece - KPIECE) * 64 + lane_v;
          const int nbh = c / (TP * 4), rem = c - nbh * (TP * 4);
          const int key = rem >> 2, pos = rem & 3;
          const int kk = key < T ? key : T - 1;
          off = (unsigned)((kk * ld + 2 * H * DH + 32 * nbh + 8 * pos) * 2);
        }
        const unsigned m0 = lds_base + buf * BUF + piece * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(m0) : "memory");
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < KDMA; ++j) {
      const int c = (ww * KDMA + j) * 64 + lane_v;  // 16-B chunk of the K image, lane-linear
      const int key = c >> 3, pos = c & 7;
      const int kk = key < T ? key : T - 1;  // rows past T repeat the last one (masked in S, finite in V)
      const unsigned off = (unsigned)((kk * ld + H * DH + ((pos ^ ((key >> 1) & 7)) << 3)) * 2);
      const unsigned m0 = lds_base + buf * BUF + (ww * KDMA + j) * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(m0) : "memory");
    }
#pragma unroll
    for (int j = 0; j < VDMA; ++j) {
      const int c = (ww * VDMA + j) * 64 + lane_v;  // chunk of the two V images [nb][key][4 chunks]
      const int nbh = c / (TP * 4), rem = c - nbh * (TP * 4);
      const int key = rem >> 2, pos = rem & 3;
      const int kk = key < T ? key : T - 1;
      const unsigned off = (unsigned)((kk * ld + 2 * H * DH + 32 * nbh + 8 * pos) * 2);
      const unsigned m0 = lds_base + buf * BUF + KBYTES + (ww * VDMA + j) * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(m0) : "memory");
    }
  };
  // Q fragments (B operand: lane (q = l31, hb) holds Q[q][16s + 8hb .. +8]).  A block's fragments are dead once its S phase is
  // over: the NEXT pair's are requested into the same registers right there, behind the next pair's DMA (hipcc's own vmcnt for
  // them then covers the DMAs it cannot see, never the other way round).
  bf16x8 qn[QPW][KS];
  auto load_q = [&](int hd, int qi) {
    const int b = hd / H, h = hd - b * H;
    const bf16* qbase = qkv + (size_t)b * T * ld + h * DH;
    // (the lane id goes through an opaque asm per call, as in issue(): hoisted out of the pair loop the per-lane row offsets are
    // 64-bit values that live -- or, once the hand-pipelined phases below took their registers, spill -- for the whole kernel)
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));
    if (blk_of(qi) < 0) return;  // wave-uniform
    const int qpos = blk_of(qi) * 32 + (lane_q & 31);
    const int qrow = qpos < T ? qpos : T - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 v = *reinterpret_cast<const uint4*>(qbase + (size_t)qrow * ld + 16 * s + 8 * (lane_q >> 5));
      qn[qi][s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  };

  bf16x8 qt[KS];  // EIGHT: the last token's row as a B fragment (the same query in every column)
  auto load_qt = [&](int hd) {
    const int b = hd / H, h = hd - b * H;
    const bf16* qbase = qkv + (size_t)b * T * ld + h * DH;
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 v = *reinterpret_cast<const uint4*>(qbase + (size_t)(T - 1) * ld + 16 * s + 8 * (lane_q >> 5));
      qt[s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  };
  // EIGHT: merge of the nine partials of pair `phd`'s last query (table `par`), by wave 0: lane = d
  auto merge_tail = [&](int phd, int par) {
    const float* mt = mrg + par * (NKB * MRG);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKB; ++j) m = fmaxf(m, mt[j * MRG]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const float wj = __builtin_amdgcn_exp2f((mt[j * MRG] - m) * scale_log2e);
      l += wj * mt[j * MRG + 1];
      o += wj * mt[j * MRG + 2 + lane];
    }
    const int pb = phd / H, ph = phd - pb * H;
    out[((size_t)pb * T + (T - 1)) * (H * DH) + ph * DH + lane] = (bf16)(l > 0.f ? o / l : 0.f);
  };

  int hd = blockIdx.x;
  if (hd >= nheads) return;
  issue(hd, 0);
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) load_q(hd, qi);
  if (tail_on) {
    load_qt(hd);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qt[s]));
  }
  // Where this kernel waits for its own VMEM operations (round 4).  vmcnt counts loads, stores and LDS-DMAs in issue order, and
  // hipcc, which cannot count across the loop's back edge, writes `s_waitcnt vmcnt(0)` in front of the first MFMA that reads Q
  // fragments loaded one pair earlier: that also waited for the eight output stores the wave had issued a moment before (block
  // 1's S phase for block 0's stores, the next pair's block 0 for block 1's) and, on the DMA waves, for the whole next pair's
  // K / V -- they could not start their block before their DMAs had landed.  So every Q fragment is "used" (an empty asm) at a
  // point where its load is old: here for the first pair, and in front of each block's output stores for the next pair's.  hipcc
  // puts its wait there, where it costs nothing, and none in front of the S phases.
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qn[qi][s]));
  const int ksw = (l31 >> 1) & 7;
  const unsigned vlane = (unsigned)(((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1)) * 2 + (lane & 3) * 8);
  int buf = 0;
  for (; hd < nheads; hd += gridDim.x, buf ^= 1) {
    // the issuing waves' shares of the pair's K / V have landed, then the barrier tells everyone; every wave is also done
    // reading the other half, which the next pair's DMA is about to overwrite.  (Waves 0 - 2 issue no DMA: they do not wait
    // here for their own output stores, the critical path of the pair.)
    // (a DMA wave that computed a block in the previous pair has already seen its DMAs land -- the vmcnt(0) in front of that
    // block's output stores, below -- and does not wait for those stores here)
    if (dma_wave && !(hd != (int)blockIdx.x && w < q_blocks)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    P_STAMP(0)
    const int nxt = hd + gridDim.x;
    if (nxt < nheads) issue(nxt, buf ^ 1);
    P_STAMP(1)
    // EIGHT: the previous pair's partials are complete (the barrier above); this pair's go to the other table
    if (tail_on && w == 0 && hd != (int)blockIdx.x) merge_tail(hd - (int)gridDim.x, buf ^ 1);
    const unsigned char* sK = smem + buf * BUF;
    const unsigned char* sV = sK + KBYTES;
    const int b = hd / H, h = hd - b * H;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      const int qb = blk_of(qi);
      if (qb < 0 || qb >= q_blocks) continue;  // wave-uniform
      const int qpos = qb * 32 + l31;
      bf16x8 qf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) qf[s] = qn[qi][s];
      f32x16 sacc[NKB];
      float mx = -INFINITY;
#if CLIPX_ATTN_SPIPE
      // S^T = K Q^T, software-pipelined by hand (round 4).  hipcc's own schedule of the plain loop below is one chain per query
      // block -- `ds_read_b128 v[0:3]; s_waitcnt lgkmcnt(0); v_mfma` 36 times, every fragment through the SAME four registers --
      // i.e. a full LDS round trip in front of every MFMA (~100 cycles per MFMA in the phase timer, 3.6 k of a block's ~10 k
      // cycles).  Here the four K fragments of key block kb + 1 are requested (inline asm: this file places the wait) before the
      // four MFMAs of block kb issue back to back on their accumulator -- nothing between two MFMAs of a chain, which would cost
      // the dependent-issue cliff of ~43 cycles each -- and the running maximum of block kb - 1 is taken while they run.  Same
      // MFMAs, same order per accumulator: the same bits.
      {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        unsigned kaddr[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) kaddr[s] = lds_base + buf * BUF + l31 * KROW + (((2 * s + hb) ^ ksw) << 4);
        i32x4 KF[2][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(KF[0][s]) : "v"(kaddr[s]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(KF[kb & 1][s]));  // (the values the MFMAs read exist from here on)
          if (kb + 1 < NKB) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(KF[(kb + 1) & 1][s]) : "v"(kaddr[s]), "n"((kb + 1) * 32 * KROW));
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x16 sb;
#pragma unroll
          for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s) sb = attn_mfma(__builtin_bit_cast(bf16x8, KF[kb & 1][s]), qf[s], sb);
          __builtin_amdgcn_sched_barrier(0);
          if (kb > 0) {  // (the asm pins these eight v_max3 here, under the last MFMA of the chain: hipcc would sink all 72 behind the phase)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb - 1][r]);
            asm volatile("" : "+v"(mx));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (kb == NKB - 1) {  // keys past T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
              sb[r] = key < T ? sb[r] : -INFINITY;
            }
          }
          sacc[kb] = sb;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[NKB - 1][r]);
      }
#else
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        f32x16 sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int c = 2 * s + hb;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kb * 32 + l31) * KROW + ((c ^ ksw) << 4));
          sb = attn_mfma(kf, qf[s], sb);
        }
        if (kb == NKB - 1) {  // keys past T
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
            sb[r] = key < T ? sb[r] : -INFINITY;
          }
        }
        sacc[kb] = sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sb[r]);
      }
#endif
#ifdef CLIPX_ABLATE
      if (timer) { asm volatile("" : "+v"(mx)); P_STAMP(2) }
#endif
#if !CLIPX_ATTN_SPIPE
      if (nxt < nheads) load_q(nxt, qi);  // this block's Q is dead: the next pair's lands under exp + PV
#endif
      // (CLIPX_ATTN_SPIPE: requested in the middle of the exp + P V loop instead -- the start of that loop, with all nine score
      // blocks, both Q sets and the output accumulators live, is the kernel's register peak)
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (mx == -INFINITY) mx = 0.f;
      const float nmx = -mx * scale_log2e;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      f32x2_t sum2 = {0.f, 0.f};
      const int tail_keys = T - (NKB - 1) * 32;
      f32x16 oacc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nb][r] = 0.f;
#if CLIPX_ATTN_SPIPE
      // exp + P V, software-pipelined by hand (round 4): hipcc's schedule of the plain loop below puts the two transposing reads
      // of a V^T fragment right in front of the MFMA that takes it (`ds_read_b64_tr_b16 x2; s_waitcnt lgkmcnt(0); v_mfma`, four
      // times per key block) and all of a block's exponentials in front of its MFMAs.  Here iteration kb computes P of block kb in
      // four chunks of four values and issues ONE MFMA of block kb - 1 behind each chunk (an in-order wave that issues two MFMAs
      // back to back sits out the first one's 32 cycles): M0 | P chunk | M1 | read the V^T fragments of M2, M3 | P chunk | M2 |
      // P chunk | M3 | read the fragments of the next block's M0, M1 | P chunk.  A chunk's new P values go to the registers whose
      // old values the MFMA just issued was the last to read (M1 frees P columns 0 - 15, M3 columns 16 - 31), the first chunk of a
      // block to a spare pair: 10 P registers and two fragment sets are live.
      // The sums (r ascending) and each accumulator's MFMAs (kb, then s2 ascending) keep their order: the same bits.
      {
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        unsigned pw[8];     // P of the block the MFMAs are taking: columns 4 j .. 4 j + 3 of the lane's 16 in pw[2 j], pw[2 j + 1]
        unsigned pnew[8];   // P of the block being computed (hipcc keeps only what is live: see above)
        s16x8 vf[2][NB];    // [s2][nb]
        auto read_v = [&](int kb, int s2) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            // lane (d = 32nb + l31, hb): keys kb*32 + 16*s2 + 4hb + {0..3} and + 8 + {0..3}, each quad one transposing read
            const unsigned char* vp = sV + nb * VHALF + (kb * 32 + 16 * s2 + 4 * hb) * 64 + vlane;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp + 8 * 64));
            vf[s2][nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        };
        auto p_chunk = [&](int kb, int c, unsigned* dst) {  // P values r = 4 c .. 4 c + 3 of block kb -> dst[2 c], dst[2 c + 1]
          const bool short_tail = kb == NKB - 1 && tail_keys <= 4;
          // (the two empty asm statements keep the chunk where it is written: pure VALU work has no order against the scheduling
          // barriers until it depends on something that has -- hipcc merged chunks and issued the MFMAs in pairs without them)
          float nmx_c = nmx;
          asm volatile("" : "+v"(nmx_c));
#pragma unroll
          for (int r = 4 * c; r < 4 * c + 4; r += 2) {
            if (short_tail && r >= 4) {
              dst[r >> 1] = 0u;
              continue;
            }
            const f32x2_t e = (f32x2_t){sacc[kb][r], sacc[kb][r + 1]} * (f32x2_t){scale_log2e, scale_log2e} + (f32x2_t){nmx_c, nmx_c};
            const f32x2_t pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            sum2 += pp;
            dst[r >> 1] = attn_pack_p(pp[0], pp[1]);
          }
          asm volatile("" : "+v"(dst[2 * c]), "+v"(dst[2 * c + 1]));
        };
        auto pv_mfma = [&](int m) {  // MFMA m = 2 * s2 + nb: P columns 16 s2 .. + 16 = pw[4 s2 .. + 3]
          const int s2 = m >> 1, nb = m & 1;
          const uint4 w4 = make_uint4(pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]);
          oacc[nb] = attn_mfma(__builtin_bit_cast(bf16x8, vf[s2][nb]), *reinterpret_cast<const bf16x8*>(&w4), oacc[nb]);
        };
        read_v(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) p_chunk(0, c, pw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 1; kb < NKB; ++kb) {
          pv_mfma(0);  // block kb - 1
          __builtin_amdgcn_sched_barrier(0);
          p_chunk(kb, 0, pnew);
          __builtin_amdgcn_sched_barrier(0);
          pv_mfma(1);
          __builtin_amdgcn_sched_barrier(0);
          read_v(kb - 1, 1);
          p_chunk(kb, 1, pnew);
          __builtin_amdgcn_sched_barrier(0);
          pv_mfma(2);
          __builtin_amdgcn_sched_barrier(0);
          p_chunk(kb, 2, pnew);
          __builtin_amdgcn_sched_barrier(0);
          pv_mfma(3);
          __builtin_amdgcn_sched_barrier(0);
          read_v(kb, 0);
          p_chunk(kb, 3, pnew);
#pragma unroll
          for (int e = 0; e < 8; ++e) pw[e] = pnew[e];
          if (kb == NKB / 2 && nxt < nheads) load_q(nxt, qi);  // this block's Q is long dead: the next pair's lands under the rest
          __builtin_amdgcn_sched_barrier(0);
        }
        pv_mfma(0);
        pv_mfma(1);
        read_v(NKB - 1, 1);
        pv_mfma(2);
        pv_mfma(3);
      }
#else
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        bf16x8 pf[2];
        const f32x16 sb = sacc[kb];
        unsigned pw[8];
        const bool short_tail = kb == NKB - 1 && tail_keys <= 4;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if (short_tail && r >= 4) {
            pw[r >> 1] = 0u;
            continue;
          }
          const f32x2_t e = (f32x2_t){sb[r], sb[r + 1]} * (f32x2_t){scale_log2e, scale_log2e} + (f32x2_t){nmx, nmx};
          const f32x2_t pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          sum2 += pp;
          pw[r >> 1] = attn_pack_p(pp[0], pp[1]);
        }
        {
          const uint4 w0 = make_uint4(pw[0], pw[1], pw[2], pw[3]), w1 = make_uint4(pw[4], pw[5], pw[6], pw[7]);
          pf[0] = *reinterpret_cast<const bf16x8*>(&w0);
          pf[1] = *reinterpret_cast<const bf16x8*>(&w1);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            // lane (d = 32nb + l31, hb): keys kb*32 + 16*s2 + 4hb + {0..3} and + 8 + {0..3}, each quad one transposing read
            const unsigned char* vp = sV + nb * VHALF + (kb * 32 + 16 * s2 + 4 * hb) * 64 + vlane;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp + 8 * 64));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[nb] = attn_mfma(__builtin_bit_cast(bf16x8, vv), pf[s2], oacc[nb]);
          }
      }
#endif
      float sum = sum2[0] + sum2[1];
      sum += __shfl_xor(sum, 32);
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qn[qi][s]));  // the next pair's Q has landed (see the prologue)
      if (tail_on) {  // ... and the last token's (requested in the previous pair's tail section)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qt[s]));
      }
      // ... and, on a DMA wave, the next pair's K / V (requested a whole block ago): nothing of this wave is in flight when its
      // stores go out, so the top of the next pair has nothing to wait for but the barrier
      if (dma_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // (not hoisted: see load_q)
      const int opos = qb * 32 + (lane_o & 31);
      if (opos < T) {
        bf16* orow = out + ((size_t)b * T + opos) * (H * DH) + h * DH;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(oacc[nb][4 * g + e] * inv);
            *reinterpret_cast<bf16x4*>(orow + 32 * nb + 8 * g + 4 * (lane_o >> 5)) = o;
          }
      }
      P_STAMP(3)
    }
    if (tail_on) {
      // ---- the last token's row against key block w (wave 0: also the one-key block NKB - 1): S, softmax pieces, P V of ONE
      // 32-key block; every column of the tile is that one query, lanes 0 and 32 hold the column that is kept
      bf16x8 qtl[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) qtl[s] = qt[s];
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1 && w != 0) break;
        const int kb = rep == 0 ? w : NKB - 1;
        f32x16 sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int c = 2 * s + hb;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kb * 32 + l31) * KROW + ((c ^ ksw) << 4));
          sb = attn_mfma(kf, qtl[s], sb);
        }
        if (rep == 0 && nxt < nheads && w != 0) load_qt(nxt);  // (wave 0: behind its second block)
        if (rep == 1 && nxt < nheads) load_qt(nxt);
        if (kb == NKB - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
            sb[r] = key < T ? sb[r] : -INFINITY;
          }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sb[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float nmx = -mx * scale_log2e;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t sum2 = {0.f, 0.f};
        unsigned pw[8];
        const bool first_quad_only = kb == NKB - 1 && T - (NKB - 1) * 32 <= 4;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if (first_quad_only && r >= 4) {
            pw[r >> 1] = 0u;
            continue;
          }
          const f32x2_t e = (f32x2_t){sb[r], sb[r + 1]} * (f32x2_t){scale_log2e, scale_log2e} + (f32x2_t){nmx, nmx};
          const f32x2_t pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          sum2 += pp;
          pw[r >> 1] = attn_pack_p(pp[0], pp[1]);
        }
        f32x16 oacc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[nb][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 w4 = make_uint4(pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const unsigned char* vp = sV + nb * VHALF + (kb * 32 + 16 * s2 + 4 * hb) * 64 + vlane;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vp + 8 * 64));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            oacc[nb] = attn_mfma(__builtin_bit_cast(bf16x8, vv), *reinterpret_cast<const bf16x8*>(&w4), oacc[nb]);
          }
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 32);
        if (l31 == 0) {  // lanes 0 and 32: the d = 32 nb + 8 g + 4 hb + {0..3} of column 0
          float* m = mrg + buf * (NKB * MRG) + kb * MRG;
          if (hb == 0) {
            m[0] = mx;
            m[1] = sum;
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) m[2 + 32 * nb + 8 * g + 4 * hb + e] = oacc[nb][4 * g + e];
        }
      }
    }
  }
  if (tail_on) {  // the last pair's partials (buf has flipped once more since they were written)
    __syncthreads();
    if (w == 0) merge_tail(hd - (int)gridDim.x, buf ^ 1);
  }
#ifdef CLIPX_ABLATE
  if (timer && lane == 0 && blockIdx.x * NW + w < 8192) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g_attn_phase[(blockIdx.x * NW + w) * 4 + i] = tph[i];
  }
#endif
#undef P_STAMP
}

static int attn_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

static hipError_t launch_attention_pk9(const bf16* qkv, bf16* out, int B, int T, int H, hipStream_t st, int q_blocks) {
  constexpr int NKB = 9, TP = NKB * 32;
  const size_t smem = (size_t)2 * (TP * 128 + 2 * TP * 64);
  const int nheads = B * H, grid = std::min(nheads, attn_cu_count());
#ifdef CLIPX_ABLATE
  // tools build, CLIPX_ATTN_CFG=13: the 8-wave form (NWT = 8 above) -- measured 188 us against 162 for the 6-wave kernel
  // (profiles/r04x_attention_8wave.log), so the product does not instantiate it
  static const int cfg8 = getenv("CLIPX_ATTN_CFG") ? atoi(getenv("CLIPX_ATTN_CFG")) : 0;
  if (cfg8 == 13 && T == (NKB - 1) * 32 + 1) {
    const size_t smem8 = smem + (size_t)2 * NKB * 68 * sizeof(float);
    auto k8 = attention_pk_kernel<NKB, 8>;
    hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem8);
    if (e8 != hipSuccess) return e8;
    hipLaunchKernelGGL(k8, dim3(grid), dim3(512), smem8, st, qkv, out, T, H, nheads, (1.f / 8.f) * 1.4426950408889634f,
                       q_blocks > 0 && q_blocks < NKB ? q_blocks : NKB);
    return hipGetLastError();
  }
#endif
  auto kern = attention_pk_kernel<NKB>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
#ifdef CLIPX_ABLATE
  // tools build only: CLIPX_ATTN_QBLOCKS=n computes the first n query blocks of every pair (6: one block on every wave, 3: one
  // block on waves 0 - 2) -- how the time of a pair depends on the blocks per wave (profiles/r04r_attention_qblocks.log)
  static const int qb_env = getenv("CLIPX_ATTN_QBLOCKS") ? atoi(getenv("CLIPX_ATTN_QBLOCKS")) : 0;
  if (qb_env > 0) q_blocks = qb_env;
  static const int timer_env = getenv("CLIPX_ATTN_PK_TIMER") ? atoi(getenv("CLIPX_ATTN_PK_TIMER")) : 0;
  static bool timer_set = false;
  if (!timer_set) {
    timer_set = true;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_pk_timer), &timer_env, sizeof(int));
  }
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(384), smem, st, qkv, out, T, H, nheads, (1.f / 8.f) * 1.4426950408889634f,
                     q_blocks > 0 && q_blocks < NKB ? q_blocks : NKB);
  return hipGetLastError();
}

#ifdef CLIPX_ABLATE
extern "C" int clipx_dbg_attn_phase(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_phase), (size_t)n * sizeof(long long));
}
#endif

template <int DH, int NKB, int NW, int QPW, bool RECOMP = false>
static hipError_t launch_attention_cfg(const bf16* qkv, bf16* out, int B, int T, int H, int causal, hipStream_t st, int q_blocks,
                                       const int* offs = nullptr, const int* lens = nullptr) {
  q_blocks = q_blocks > 0 && q_blocks < NKB ? q_blocks : NKB;
  constexpr int KROW = DH == 64 ? 128 : 176, DV = (DH + 31) / 32 * 32;
  const size_t smem = (size_t)NKB * 32 * KROW + (size_t)DV * (NKB * 64 + 8);
  const float scale_log2e = (1.f / sqrtf((float)DH)) * 1.4426950408889634f;
  const dim3 grid(B * H), block(NW * 64);
#ifdef CLIPX_ABLATE
  static const int dbg = getenv("CLIPX_ATTN_DBG") ? atoi(getenv("CLIPX_ATTN_DBG")) : 0;  // ablations (tools build only), see the kernel
#else
  constexpr int dbg = 0;
#endif
  if (causal) {
    auto kern = attention_kernel<DH, NKB, NW, QPW, true, RECOMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, smem, st, qkv, out, T, H, scale_log2e, dbg, q_blocks, offs, lens);
  } else {
    auto kern = attention_kernel<DH, NKB, NW, QPW, false, RECOMP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, smem, st, qkv, out, T, H, scale_log2e, dbg, q_blocks, offs, lens);
  }
  return hipGetLastError();
}

hipError_t launch_attention(const bf16* qkv, bf16* out, int B, int T, int H, int dh, int causal, hipStream_t st, int q_blocks,
                            const int* offs, const int* lens) {
  if (B <= 0) return hipSuccess;
  if ((offs != nullptr) != (lens != nullptr)) return hipErrorInvalidValue;
  if (offs && (dh != 64 || (T + 31) / 32 > 4)) return hipErrorInvalidValue;  // ragged batches: the short-sequence configurations only
  const int nkb = (T + 31) / 32;
  if (dh == 80) {  // ViT-H/14 image tower (T = 257), ViT-bigG is dh 104: not built
    switch (nkb) {
      case 1: return launch_attention_cfg<80, 1, 1, 1>(qkv, out, B, T, H, causal, st, q_blocks);
      case 2: return launch_attention_cfg<80, 2, 2, 1>(qkv, out, B, T, H, causal, st, q_blocks);
      case 3: return launch_attention_cfg<80, 3, 3, 1>(qkv, out, B, T, H, causal, st, q_blocks);
      case 9: return launch_attention_cfg<80, 9, 9, 1, true>(qkv, out, B, T, H, causal, st, q_blocks);
      default: return hipErrorInvalidValue;
    }
  }
  if (dh != 64) return hipErrorInvalidValue;
  switch (nkb) {
    case 1: return launch_attention_cfg<64, 1, 1, 1>(qkv, out, B, T, H, causal, st, q_blocks, offs, lens);
    case 2: return launch_attention_cfg<64, 2, 2, 1>(qkv, out, B, T, H, causal, st, q_blocks, offs, lens);   // ViT-B/32 image (T=50)
    case 3: return launch_attention_cfg<64, 3, 3, 1>(qkv, out, B, T, H, causal, st, q_blocks, offs, lens);   // text (T=77)
    case 4: return launch_attention_cfg<64, 4, 4, 1>(qkv, out, B, T, H, causal, st, q_blocks, offs, lens);
    case 5: return launch_attention_cfg<64, 5, 3, 2>(qkv, out, B, T, H, causal, st, q_blocks);
    case 6: return launch_attention_cfg<64, 6, 3, 2>(qkv, out, B, T, H, causal, st, q_blocks);
    case 7: return launch_attention_cfg<64, 7, 4, 2>(qkv, out, B, T, H, causal, st, q_blocks);   // ViT-B/16 image (T=197)
    case 8: return launch_attention_cfg<64, 8, 4, 2>(qkv, out, B, T, H, causal, st, q_blocks);
    case 9: {  // ViT-L/14 image (T=257)
#ifdef CLIPX_ABLATE
      static const int cfg = getenv("CLIPX_ATTN_CFG") ? atoi(getenv("CLIPX_ATTN_CFG")) : 0;
#else
      constexpr int cfg = 0;
#endif
      if (cfg == 4) return launch_attention_cfg<64, 9, 4, 3>(qkv, out, B, T, H, causal, st, q_blocks);
#ifdef CLIPX_ABLATE
      if (cfg == 5) return launch_attention_cfg<64, 9, 9, 1, true>(qkv, out, B, T, H, causal, st, q_blocks);   // 9 waves, S recomputed
      if (cfg == 6) return launch_attention_cfg<64, 9, 5, 2>(qkv, out, B, T, H, causal, st, q_blocks);
      if (cfg == 7) return launch_attention_cfg<64, 9, 9, 1>(qkv, out, B, T, H, causal, st, q_blocks);         // 9 waves, S kept (144 regs)
      if (cfg == 8) return launch_attention_cfg<64, 9, 5, 2, true>(qkv, out, B, T, H, causal, st, q_blocks);
#endif
      if (cfg == 9 && !causal) {  // phase timer
        constexpr int KROW = 128, DV = 64;
        const size_t smem = (size_t)9 * 32 * KROW + (size_t)DV * (9 * 64 + 8);
        auto kern = attention_kernel<64, 9, 3, 3, false, false, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(B * H), dim3(192), smem, st, qkv, out, T, H, (1.f / 8.f) * 1.4426950408889634f, 0, 9, (const int*)nullptr, (const int*)nullptr);
        return hipGetLastError();
      }
#ifdef CLIPX_ABLATE
      if (cfg == 10 || causal) return launch_attention_cfg<64, 9, 3, 3>(qkv, out, B, T, H, causal, st, q_blocks);  // the one-head-per-workgroup kernel (A/B)
#else
      if (causal) return launch_attention_cfg<64, 9, 3, 3>(qkv, out, B, T, H, causal, st, q_blocks);
#endif
      return launch_attention_pk9(qkv, out, B, T, H, st, q_blocks);
    }
    default: return hipErrorInvalidValue;
  }
}

// =============================================================================================
// text embedding gather
// =============================================================================================
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ ids, const float* __restrict__ tok,
                                                        const float* __restrict__ pos, float* __restrict__ x, int BT,
                                                        int T, int d, int vocab, void* __restrict__ x16, int x16_f16,
                                                        const int* __restrict__ rowmap) {
  const int d4 = d / 4;
  const int64_t total = (int64_t)BT * d4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % d4);
    const int row = (int)(idx / d4);
    const int src = rowmap ? rowmap[row] : row;  // ragged: output row `row` is token src = b T + t of the rectangular batch
    int id = ids[src];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float4 a = reinterpret_cast<const float4*>(tok + (size_t)id * d)[c];
    const float4 p = reinterpret_cast<const float4*>(pos + (size_t)(src % T) * d)[c];
    const float4 o = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    if (x) reinterpret_cast<float4*>(x + (size_t)row * d)[c] = o;
    if (x16 && x16_f16) {
      f16x4 h;
      h[0] = (_Float16)o.x; h[1] = (_Float16)o.y; h[2] = (_Float16)o.z; h[3] = (_Float16)o.w;
      reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(x16) + (size_t)row * d)[c] = h;
    } else if (x16) {
      bf16x4 h;
      h[0] = (bf16)o.x; h[1] = (bf16)o.y; h[2] = (bf16)o.z; h[3] = (bf16)o.w;
      reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(x16) + (size_t)row * d)[c] = h;
    }
  }
}

hipError_t launch_text_embed(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int T, int d,
                             int vocab, hipStream_t st, void* x16, int x16_f16, const int* rowmap, int nrows) {
  if (B <= 0) return hipSuccess;
  const int rows = rowmap ? nrows : B * T;
  if (rows <= 0) return hipSuccess;
  const int64_t total = (int64_t)rows * (d / 4);
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(text_embed_kernel, dim3(blocks), dim3(256), 0, st, ids, tok_emb, pos_emb, x, rows, T, d, vocab, x16, x16_f16, rowmap);
  return hipGetLastError();
}

// =============================================================================================
// tail: pool -> LayerNorm -> projection -> L2 normalise -> fp16   (mapper.py:58-59, 66-67)
// =============================================================================================
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Two launches: (E / 64, B) workgroups each LayerNorm the sample's pooled row (recomputed per workgroup: d floats) and
// produce 64 projection outputs with 4 threads per output; a second kernel normalises.  (One workgroup per sample doing the
// whole d x E projection took 79 us -- 3 % of a B = 1 image query, 8 % of a text query.)
__global__ __launch_bounds__(256) void tail_proj_kernel(const void* __restrict__ x, const int32_t* __restrict__ ids,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const bf16* __restrict__ proj, float* __restrict__ o_raw, int T, int d,
                                                       int E, float eps, int x_f16, int* __restrict__ range_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* y = reinterpret_cast<float*>(smem);  // [d]
  float* red = y + d;                         // [4]   (all LDS in the one dynamic region: guide G17)
  int* s_posp = reinterpret_cast<int*>(red + 4);
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) {
    int best = 0;
    if (ids) {  // EOT token = highest id; first occurrence (torch.argmax semantics of the reference model)
      int bv = ids[(size_t)b * T];
      for (int t = 1; t < T; ++t) {
        const int v = ids[(size_t)b * T + t];
        if (v > bv) { bv = v; best = t; }
      }
    }
    *s_posp = best;
  }
  __syncthreads();
  const size_t xrow = ((size_t)b * T + *s_posp) * d;
  float s = 0.f;
  for (int c = tid; c < d; c += 256) {
    y[c] = x_f16 ? (float)reinterpret_cast<const _Float16*>(x)[xrow + c] : reinterpret_cast<const float*>(x)[xrow + c];
    s += y[c];
  }
  const float tot = block_sum_256(s, red);
  if (range_flag && tid == 0 && !(fabsf(tot) <= 3.0e38f)) atomicOr(range_flag, 1);  // the last state of the fp16 stream (rowstats_kernel)
  const float mean = tot / d;
  float q = 0.f;
  for (int c = tid; c < d; c += 256) { const float a = y[c] - mean; q += a * a; }
  const float rstd = 1.f / sqrtf(block_sum_256(q, red) / d + eps);
  for (int c = tid; c < d; c += 256) y[c] = (y[c] - mean) * rstd * gamma[c] + beta[c];
  __syncthreads();
  const int e = blockIdx.x * 64 + (tid >> 2), part = tid & 3;
  const int seg = d >> 2;  // d % 32 == 0
  float acc = 0.f;
  if (e < E) {
    const bf16x8* pr = reinterpret_cast<const bf16x8*>(proj + (size_t)e * d + part * seg);
    const float* yp = y + part * seg;
    for (int c8 = 0; c8 < seg / 8; ++c8) {
      const bf16x8 pw = pr[c8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += yp[c8 * 8 + j] * (float)pw[j];
    }
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  if (part == 0 && e < E) o_raw[(size_t)b * E + e] = acc;
}

__global__ __launch_bounds__(256) void tail_norm_kernel(const float* __restrict__ o_raw, uint16_t* __restrict__ out_f16,
                                                       float* __restrict__ out_f32, int E) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float ss = 0.f;
  for (int e = tid; e < E; e += 256) { const float v = o_raw[(size_t)b * E + e]; ss += v * v; }
  const float nrm = sqrtf(block_sum_256(ss, red));
  for (int e = tid; e < E; e += 256) {
    const float v = o_raw[(size_t)b * E + e] / nrm;  // no epsilon: mapper.py:58 divides by the raw norm
    const _Float16 hv = (_Float16)v;
    out_f16[(size_t)b * E + e] = *reinterpret_cast<const uint16_t*>(&hv);
    if (out_f32) out_f32[(size_t)b * E + e] = v;
  }
}

// The rows the embedding is read from -- token 0 of every image, the EOT token (highest id, first occurrence) of every caption --
// copied out of the [B * T, d] attention output and residual stream into two compact [B, d] buffers: the last block's
// out-proj / MLP then run on B rows instead of B * T (nothing else of that block's output is ever read).
__global__ __launch_bounds__(256) void gather_pooled_kernel(const bf16* __restrict__ att, const _Float16* __restrict__ x,
                                                           const int32_t* __restrict__ ids, bf16* __restrict__ attc,
                                                           _Float16* __restrict__ xc, int T, int d, const int* __restrict__ rows) {
  __shared__ int s_pos;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int best = 0;
    if (ids && !rows) {
      int bv = ids[(size_t)b * T];
      for (int t = 1; t < T; ++t) {
        const int v = ids[(size_t)b * T + t];
        if (v > bv) { bv = v; best = t; }
      }
    }
    s_pos = best;
  }
  __syncthreads();
  const size_t row = rows ? (size_t)rows[b] : (size_t)b * T + s_pos;  // ragged batches bring the pooled row of every sample
  const uint4* a = reinterpret_cast<const uint4*>(att + row * d);
  const uint4* xs = reinterpret_cast<const uint4*>(x + row * d);
  uint4* ao = reinterpret_cast<uint4*>(attc + (size_t)b * d);
  uint4* xo = reinterpret_cast<uint4*>(xc + (size_t)b * d);
  for (int c = tid; c < d / 8; c += 256) {
    ao[c] = a[c];
    xo[c] = xs[c];
  }
}

hipError_t launch_gather_pooled(const bf16* att, const void* x16, const int32_t* ids_or_null, bf16* attc, void* xc, int B, int T,
                                int d, hipStream_t st, const int* rows_or_null) {
  if (B <= 0) return hipSuccess;
  if (d % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_pooled_kernel, dim3(B), dim3(256), 0, st, att, reinterpret_cast<const _Float16*>(x16), ids_or_null, attc,
                     reinterpret_cast<_Float16*>(xc), T, d, rows_or_null);
  return hipGetLastError();
}

hipError_t launch_tail(const void* x, const int32_t* ids_or_null, const float* gamma, const float* beta, const bf16* proj,
                       uint16_t* out_f16, float* out_f32_or_null, float* scratch, int B, int T, int d, int E, float eps,
                       hipStream_t st, int x_f16, int* range_flag) {
  if (B <= 0) return hipSuccess;
  if (d % 32 != 0 || !scratch) return hipErrorInvalidValue;
  const size_t smem = (size_t)(d + 8) * sizeof(float);
  hipLaunchKernelGGL(tail_proj_kernel, dim3((E + 63) / 64, B), dim3(256), smem, st, x, ids_or_null, gamma, beta, proj, scratch,
                     T, d, E, eps, x_f16, range_flag);
  hipLaunchKernelGGL(tail_norm_kernel, dim3(B), dim3(256), 0, st, scratch, out_f16, out_f32_or_null, E);
  return hipGetLastError();
}

// =============================================================================================
// weight conversion
// =============================================================================================
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (bf16)in[i];
}
hipError_t launch_f32_to_bf16(const float* in, bf16* out, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(1024), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}
__global__ void pad_rows_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, int rows, int k, int kp) {
  const int64_t total = (int64_t)rows * kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / kp), c = (int)(i % kp);
    out[i] = c < k ? (bf16)in[(size_t)r * k + c] : (bf16)0.f;
  }
}
hipError_t launch_pad_rows_bf16(const float* in, bf16* out, int rows, int k, int kp, hipStream_t st) {
  hipLaunchKernelGGL(pad_rows_bf16_kernel, dim3(1024), dim3(256), 0, st, in, out, rows, k, kp);
  return hipGetLastError();
}

}  // namespace clipx
