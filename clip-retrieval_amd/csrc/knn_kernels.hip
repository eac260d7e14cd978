// knn_kernels.hip -- gfx950 kernels of the flat inner-product kNN (search half of the hot path).
//
// Stands in for the arithmetic inside faiss `IndexFlatIP.search*` that the reference calls at
// clip_retrieval/clip_back.py:362 (search_and_reconstruct), clip_filter.py:52,55 (range_search,
// search).  Written for CDNA4 directly: 64-wide waves, v_mfma_f32_32x32x16_f16, LDS queues.
//
// Data layout in HBM
//   X        fp16 [N, d] row-major, d % 256 == 0 (rows are 16-B aligned, whole 128-B lines)
//   qfrag    fp16 MFMA B-fragments of the <=32 queries of one scan:
//            [d/16 k-steps][2 parts: hi, lo*2048][64 lanes][8 halves]; lane (n=l&31, h=l>>5)
//            holds q_n[16s + 8h + j], j=0..7.  q = hi + lo/2048 captures 22 mantissa bits and
//            keeps `lo` out of the fp16 subnormal range.
//   part_*   per-workgroup sorted top-k lists, merged by knn_merge_kernel.
//
// Scan kernel (HBM-bound: algorithmic bytes = N*d*2 per launch, independent of the number of
// queries <= 32).  One 512-thread workgroup per CU, grid-strided over groups of 8 row tiles;
// wave w owns a 32-row tile: its lanes load the A-fragments (16 B each) straight from HBM into
// VGPRs -- an X byte is used by exactly one wave, so an LDS round trip would be pure overhead
// (guide: "GEMV / operand streamed once per block -> load straight to VGPRs").  The 32 queries
// are the stationary B operand, read from LDS as conflict-free ds_read_b128.  A 32x32 fp32 score
// tile lands with column = query, so every lane owns ONE query and filters its 16 scores against
// that query's running k-th-best threshold; survivors (rare after warm-up) are appended to a
// per-query LDS queue, pruned by a rank-select when a queue overflows.  Thresholds are shared
// between workgroups through a global atomicMax word per query (a lower bound only: staleness
// costs time, never correctness).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <algorithm>
#include "knn_kernels.h"

namespace knnx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// order-preserving float <-> int map (involution), so atomicMax on ints orders floats
__device__ __forceinline__ int enc_f(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float dec_f(int e) { return __int_as_float(e >= 0 ? e : (e ^ 0x7fffffff)); }

// strict total order of results: score descending, then id ascending
__device__ __forceinline__ bool better(float sa, uint32_t ia, float sb, uint32_t ib) {
  return (sa > sb) || (sa == sb && ia < ib);
}

// ---------------------------------------------------------------------------------------------
// query preparation: f32 [nq, d] -> hi/lo fp16 MFMA fragments; resets the per-scan global state
// ---------------------------------------------------------------------------------------------
// wide = 1: slot 1 holds the fp16 hi part of query n + 32 instead of the lo part of query n (QB = 2 scan)
// gridDim.y > 1 (the multi-block IVF pass; the sample scans of a register-stationary batch): block b = blockIdx.y prepares queries
// [32 b, 32 b + 32) (wide: [64 b, 64 b + 64)) into the b-th fragment image and resets their thresholds in thr_g and in thr_g2 (the
// coarse and the fine scan of one IVF pass: two arrays).
__global__ void knn_prep_queries_kernel(const float* __restrict__ q, int nq, int d,
                                        _Float16* __restrict__ qfrag, int* __restrict__ thr_g,
                                        unsigned* __restrict__ range_cnt, int wide, const unsigned* __restrict__ gate,
                                        int* __restrict__ thr_g2) {
  if (gate && *gate == 0) return;
  const int s = blockIdx.x;  // k-step
  const int lane = threadIdx.x;
  const int qpb = wide ? KNN_NQ_MAX : KNN_NQ;  // queries per block (a wide block: the hi parts of 64 queries)
  if (gridDim.y > 1) {
    const int b = blockIdx.y;
    q += (size_t)b * qpb * d;
    nq -= qpb * b;
    qfrag += (size_t)b * d * 64;
    thr_g += qpb * b;
    if (thr_g2) thr_g2 += qpb * b;
  }
  const int n = lane & 31, h = lane >> 5;
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = 16 * s + 8 * h + j;
    const float v = (n < nq) ? q[(size_t)n * d + kk] : 0.f;
    const _Float16 vh = (_Float16)v;
    hi[j] = vh;
    if (wide) lo[j] = (n + 32 < nq) ? (_Float16)q[(size_t)(n + 32) * d + kk] : (_Float16)0.f;
    else lo[j] = (_Float16)((v - (float)vh) * KNN_LO_SCALE);
  }
  half8* out = reinterpret_cast<half8*>(qfrag);
  out[(size_t)(s * 2 + 0) * 64 + lane] = hi;
  out[(size_t)(s * 2 + 1) * 64 + lane] = lo;
  if (s == 0 && lane < (gridDim.y > 1 ? qpb : KNN_NQ_MAX)) {
    thr_g[lane] = enc_f(-INFINITY);
    if (thr_g2) thr_g2[lane] = enc_f(-INFINITY);
    if (range_cnt && lane < KNN_NQ) range_cnt[lane] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// the scan
// ---------------------------------------------------------------------------------------------
struct ScanSmem {
  // carved from dynamic LDS in this order (all offsets multiples of 16 B)
  half8* qf;         // [KS*2*64]
  float* cand_s;     // [NQ*cap]
  uint32_t* cand_i;  // [NQ*cap]
  int* cnt;          // [NQ]
  int* thr;          // [NQ]  (encoded)
  int* flag;         // [4]
};

__device__ __forceinline__ ScanSmem carve(unsigned char* base, int d, int cap, int nqs) {
  ScanSmem s;
  size_t off = 0;
  s.qf = reinterpret_cast<half8*>(base + off);
  off += (size_t)d * 128;  // (d/16) * 2 * 64 * 16 B
  s.cand_s = reinterpret_cast<float*>(base + off);
  off += (size_t)nqs * cap * 4;
  s.cand_i = reinterpret_cast<uint32_t*>(base + off);
  off += (size_t)nqs * cap * 4;
  s.cnt = reinterpret_cast<int*>(base + off);
  off += nqs * 4;
  s.thr = reinterpret_cast<int*>(base + off);
  off += nqs * 4;
  s.flag = reinterpret_cast<int*>(base + off);
  return s;
}

// One wave sorts/prunes the queue of query `qq`: keeps the best min(n, k) entries, sorted, and
// raises the threshold to the k-th best.  n <= cap <= 128 (two entries per lane).
__device__ __forceinline__ void prune_query(const ScanSmem& sm, int qq, int cap, int k, int lane,
                                            int* __restrict__ thr_g) {
  int n = sm.cnt[qq];
  n = n < cap ? n : cap;
  float* cs = sm.cand_s + (size_t)qq * cap;
  uint32_t* ci = sm.cand_i + (size_t)qq * cap;
  const int e0 = lane, e1 = lane + 64;
  const bool v0 = e0 < n, v1 = e1 < n;
  const float s0 = v0 ? cs[e0] : 0.f, s1 = v1 ? cs[e1] : 0.f;
  const uint32_t i0 = v0 ? ci[e0] : 0u, i1 = v1 ? ci[e1] : 0u;
  int r0 = 0, r1 = 0;
  for (int j = 0; j < n; ++j) {
    const float sj = cs[j];
    const uint32_t ij = ci[j];
    r0 += better(sj, ij, s0, i0) ? 1 : 0;
    r1 += better(sj, ij, s1, i1) ? 1 : 0;
  }
  // all reads above are complete (in-order LDS queue of this wave) before the writes below issue
  __builtin_amdgcn_wave_barrier();
  if (v0 && r0 < k) { cs[r0] = s0; ci[r0] = i0; }
  if (v1 && r1 < k) { cs[r1] = s1; ci[r1] = i1; }
  if (n >= k) {
    if (v0 && r0 == k - 1) { atomicMax(&sm.thr[qq], enc_f(s0)); atomicMax(&thr_g[qq], enc_f(s0)); }
    if (v1 && r1 == k - 1) { atomicMax(&sm.thr[qq], enc_f(s1)); atomicMax(&thr_g[qq], enc_f(s1)); }
  }
  if (lane == 0) sm.cnt[qq] = n < k ? n : k;
}

// Multi-block IVF pass: the workgroups [s, e) of a G-workgroup launch that serve query block b, given the blocks' work-list
// lengths wk[0 .. nblk): every block owns one workgroup plus a share of the other G - nblk proportional to its tiles (blocks of a
// batch differ by +-10 % in tiles; an equal split leaves the chip waiting for the longest).  The list scan and the merge of its
// partial lists compute the same ranges from the same array.
__device__ __forceinline__ void ivfm_range(const unsigned* __restrict__ wk, int nblk, int G, int b, int& s, int& e) {
  unsigned long long tot = 0, cum = 0, upto = 0;
  for (int i = 0; i < nblk; ++i) {
    const unsigned v = wk[i];
    tot += v;
    if (i < b) cum += v;
    if (i <= b) upto += v;
  }
  const unsigned long long Gf = (unsigned long long)(G - nblk);
  s = b + (tot ? (int)(Gf * cum / tot) : 0);
  e = b + 1 + (tot ? (int)(Gf * upto / tot) : 0);
}

// IVF = true: instead of all row tiles 0..N/32, the waves walk a WORK LIST of 32-row tiles (the tiles of the inverted
// lists probed by at least one query of this scan, built by ivf_expand_kernel); item = {tile, query mask, valid rows}:
// a lane (= query column) only admits a score when its query probes that tile's list -- exactly the candidate set
// faiss IndexIVFFlat scans for that query -- and row ids are positions in the list-sorted arena (mapped back by idmap).
// QB = 2 ("wide" scan): the two MFMA B operands are the fp16 HI parts of queries 0..31 and 32..63 instead of the hi / lo
// parts of queries 0..31: the same instruction stream and the same HBM bytes serve 64 queries, with scores that are off
// by at most |q - fp16(q)| * |x|.  The caller asks for the best kw > k rows by that approximate score, re-scores them
// exactly (knn_rescore_kernel) and PROVES the top-k exact from the gap between the kw-th approximate and the k-th exact
// score, falling back to the QB = 1 scan for a query whose gap is too small.  `gate`: run only if *gate != 0.
template <int NCH, int MODE, bool NT, bool IVF, int QB>  // NCH = d/128 (even); MODE 0 = top-k, 1 = range; NT = nontemporal loads
__global__ __launch_bounds__(KNN_WG, 2) void knn_scan_kernel(
    const _Float16* __restrict__ X, int64_t N, const _Float16* __restrict__ qfrag, int nq, int k, int cap,
    int* __restrict__ thr_g, float* __restrict__ part_s, uint32_t* __restrict__ part_i, int* __restrict__ part_n,
    float range_thr, unsigned* __restrict__ range_cnt, unsigned range_cap, float* __restrict__ range_s,
    uint32_t* __restrict__ range_i, const uint4* __restrict__ work, const unsigned* __restrict__ nwork_ptr,
    const unsigned* __restrict__ gate, int tstride, int nblk, unsigned work_stride) {
  constexpr int D = NCH * 128;
  constexpr int KS = D / 16;
  constexpr int NQ = 32 * QB;  // queries (= queues, thresholds) of this scan
  static_assert(QB == 1 || (MODE == 0 && !IVF), "the wide scan exists for the flat top-k mode only");
  if (gate && *gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const ScanSmem sm = carve(smem_raw, D, cap, NQ);

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int q = lane & 31, hb = lane >> 5;

  // nblk > 1: SEVERAL query blocks (32 queries; QB = 2: 64) in one launch -- the IVF pass of a batch of up to 32 nblk queries, the
  // four 64-query sample scans of a 256-query register-stationary batch.  Workgroup g
  // serves query block g % nblk (its own fragment image, thresholds, work list, result slots) as member g / nblk of that block's
  // gridDim.x / nblk workgroups: nblk independent scans side by side, one launch and one set of small kernels around it.
  // Which block: the list scan (IVF) splits the workgroups in proportion to the blocks' work (ivfm_range); the flat scan (the
  // coarse quantiser: equal work) puts the nblk workgroups that visit the same row tiles on ONE XCD (workgroup g runs on XCD g % 8)
  // when the grid allows it, so that a centroid tile is read from HBM once and from that XCD's L2 by the other blocks.
  int blk = 0, bidx = (int)blockIdx.x, bgrid = (int)gridDim.x;
  size_t slot = blockIdx.x;  // where this workgroup's partial lists go
  if (nblk > 1) {
    const int g = (int)blockIdx.x, G = (int)gridDim.x;
    if (IVF && MODE != 2) {
      bgrid = 0;
      for (int b = 0; b < nblk; ++b) {
        int s_, e_;
        ivfm_range(nwork_ptr, nblk, G, b, s_, e_);
        if (g >= s_ && g < e_) { blk = b; bidx = g - s_; bgrid = e_ - s_; }
      }
      if (bgrid == 0) return;  // (no work at all: only the first workgroup of every block stays, to publish empty lists)
    } else {
      bgrid = G / nblk;
      if (G % (8 * nblk) == 0) { blk = (g >> 3) % nblk; bidx = (g / (8 * nblk)) * 8 + (g & 7); }
      else { blk = g % nblk; bidx = g / nblk; }
      slot = (size_t)blk * bgrid + bidx;
    }
  }
  if (nblk > 1) {
    qfrag += (size_t)blk * D * 64;
    nq = nq - NQ * blk < NQ ? nq - NQ * blk : NQ;
    thr_g += NQ * blk;
    if (IVF) {
      work += (size_t)blk * work_stride;
      nwork_ptr += blk;
    }
  }

  // stage the query fragments (already in fragment order) and reset the queues
  {
    const uint4* src = reinterpret_cast<const uint4*>(qfrag);
    uint4* dst = reinterpret_cast<uint4*>(sm.qf);
    for (int i = tid; i < KS * 2 * 64; i += KNN_WG) dst[i] = src[i];
    if (tid < NQ) { sm.cnt[tid] = 0; sm.thr[tid] = enc_f(-INFINITY); }
    if (tid < 4) sm.flag[tid] = 0;
  }
  __syncthreads();

  // tstride > 1 (flat scans): only every tstride-th 32-row tile is visited -- the strided SAMPLE of the index from which
  // the register-stationary scan (knn_rq_kernels.hip) takes its per-query thresholds
  const int64_t ntile = IVF ? (int64_t)*nwork_ptr : (((N + 31) >> 5) + tstride - 1) / tstride;  // IVF: number of work items
  const int64_t ngroup = (ntile + KNN_WAVES - 1) / KNN_WAVES;

  half8 a0[8], a1[8];
  auto item_of = [&](int64_t grp) -> uint4 {  // IVF work item of this wave in group `grp` (wave-uniform address)
    const int64_t idx = grp * KNN_WAVES + w;
    return (grp < ngroup && idx < ntile) ? work[idx] : make_uint4(0u, 0u, 0u, 0u);
  };
  uint4 it_cur = make_uint4(0u, 0u, 0u, 0u), it_nxt = it_cur, it_n2 = it_cur;  // items are fetched two groups ahead
  auto row_ptr = [&](int64_t grp) -> const half8* {
    int64_t row;
    if (IVF) {
      row = (int64_t)it_nxt.x * 32 + q;  // the arena is padded to whole tiles: always in range
    } else {
      row = (grp * KNN_WAVES + w) * (int64_t)tstride * 32 + q;
      row = row < N ? row : N - 1;
    }
    return reinterpret_cast<const half8*>(X + (size_t)row * D) + hb;
  };
  auto load_chunk = [&](half8 (&buf)[8], const half8* xp, int c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) buf[j] = NT ? __builtin_nontemporal_load(xp + 2 * (c * 8 + j)) : xp[2 * (c * 8 + j)];
    // pin the burst: hipcc otherwise sinks each load next to its MFMA (2 loads in flight, not 8-16)
    __builtin_amdgcn_sched_barrier(0);
  };

  int64_t grp = bidx;
  if (IVF) {  // row_ptr() addresses the tile of it_nxt
    it_nxt = item_of(grp);
    it_n2 = item_of(grp + bgrid);
  }
  const half8* xp = row_ptr(grp < ngroup ? grp : 0);
  if (grp < ngroup) load_chunk(a0, xp, 0);

  for (int rnd = 0; grp < ngroup; grp += bgrid, ++rnd) {
    const int64_t gnext = grp + bgrid;
    if (IVF) {
      it_cur = it_nxt;
      it_nxt = it_n2;
      it_n2 = item_of(gnext + bgrid);
    }
    const half8* xnext = row_ptr(gnext < ngroup ? gnext : grp);
    float16v acc_h, acc_l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_h[r] = 0.f; acc_l[r] = 0.f; }

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      half8(&cur)[8] = (c & 1) ? a1 : a0;
      half8(&nxt)[8] = (c & 1) ? a0 : a1;
      if (c + 1 < NCH) load_chunk(nxt, xp, c + 1);
      else load_chunk(nxt, xnext, 0);  // next group's first chunk flies across the append phase
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = c * 8 + j;
        const half8 bh = sm.qf[(s * 2 + 0) * 64 + lane];
        const half8 bl = sm.qf[(s * 2 + 1) * 64 + lane];
        acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[j], bh, acc_h, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[j], bl, acc_l, 0, 0, 0);
      }
    }
    xp = xnext;

    // ---- filter: lane (q, hb) owns rows row0 + (r&3) + 8*(r>>2) + 4*hb of query q
    const int64_t row0 = (IVF ? (int64_t)it_cur.x : (grp * KNN_WAVES + w) * (int64_t)tstride) * 32 + 4 * hb;
    // IVF: rows of this tile beyond the list's size are padding; the query must probe the tile's list
    const int64_t row_lim = IVF ? (int64_t)it_cur.x * 32 + (int64_t)it_cur.z : N;
    const bool q_ok = q < nq && (!IVF || ((it_cur.y >> q) & 1u));
    float sc[16 * QB];
    unsigned pend = 0;  // bit r + 16*blk: score r of query q + 32*blk still has to be queued
    if (MODE == 0) {
#pragma unroll
      for (int blk = 0; blk < QB; ++blk) {
        const int qq = q + 32 * blk;
        const float thr = dec_f(sm.thr[qq]);
        const bool ok_q = QB == 1 ? q_ok : qq < nq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = QB == 1 ? acc_h[r] + acc_l[r] * KNN_LO_INV : (blk == 0 ? acc_h[r] : acc_l[r]);
          sc[16 * blk + r] = v;
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
          if (v >= thr && row < row_lim && ok_q) pend |= 1u << (16 * blk + r);
        }
      }
      const int par = rnd & 1;
      for (;;) {
        bool ovf = false;
#pragma unroll
        for (int b = 0; b < 16 * QB; ++b) {
          if (pend & (1u << b)) {
            const int qq = q + 32 * (b >> 4), r = b & 15;
            const int pos = atomicAdd(&sm.cnt[qq], 1);
            if (pos < cap) {
              sm.cand_s[(size_t)qq * cap + pos] = sc[b];
              sm.cand_i[(size_t)qq * cap + pos] = (uint32_t)(row0 + (r & 3) + 8 * (r >> 2));
              pend &= ~(1u << b);
            } else {
              ovf = true;
            }
          }
        }
        if (ovf) sm.flag[par] = 1;
        __syncthreads();  // (A) every append of this attempt has landed
        if (sm.flag[par] == 0) break;
        for (int qq = w; qq < NQ; qq += KNN_WAVES) prune_query(sm, qq, cap, k, lane, thr_g);
        __syncthreads();  // (B) queues pruned, everyone has read flag[par]
        if (tid == 0) sm.flag[par] = 0;
        __syncthreads();  // (C) flag cleared before anyone appends again
#pragma unroll
        for (int blk = 0; blk < QB; ++blk) {
          const float thr2 = dec_f(sm.thr[q + 32 * blk]);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if ((pend & (1u << (16 * blk + r))) && !(sc[16 * blk + r] >= thr2)) pend &= ~(1u << (16 * blk + r));
        }
      }
      // every 4th round pull the other workgroups' thresholds (lower bounds, monotone)
      if ((rnd & 3) == 3 && w == 0 && lane < NQ) {
        const int g = __hip_atomic_load(&thr_g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(&sm.thr[lane], g);
      }
    } else if (MODE == 2) {
      // every score goes to range_s [nq, N] (N = range_cap): the IVF coarse quantiser for nprobe > 64, where the
      // top-nprobe centroids of a query no longer fit the LDS queues and are selected from the score matrix instead
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
        if (row < row_lim && q_ok) range_s[(size_t)(32 * blk + q) * range_cap + row] = acc_h[r] + acc_l[r] * KNN_LO_INV;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = acc_h[r] + acc_l[r] * KNN_LO_INV;
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
        if (s > range_thr && row < row_lim && q_ok) {
          const unsigned pos = atomicAdd(&range_cnt[q], 1u);
          if (pos < range_cap) {
            range_s[(size_t)q * range_cap + pos] = s;
            range_i[(size_t)q * range_cap + pos] = (uint32_t)row;
          }
        }
      }
    }
  }

  if (MODE == 0) {
    __syncthreads();
    for (int qq = w; qq < NQ; qq += KNN_WAVES) prune_query(sm, qq, cap, k, lane, thr_g);
    __syncthreads();
    // publish this workgroup's sorted lists (nblk > 1, flat: block-major -- the lists of block b are slots [b bgrid, (b + 1) bgrid);
    // IVF: slot = workgroup, block b owns the slots of ivfm_range)
    for (int i = tid; i < NQ * k; i += KNN_WG) {
      const int qq = i / k, j = i - qq * k;
      const int n = sm.cnt[qq];
      const size_t o = (slot * NQ + qq) * k + j;
      if (j < n) {
        part_s[o] = sm.cand_s[(size_t)qq * cap + j];
        part_i[o] = sm.cand_i[(size_t)qq * cap + j];
      }
    }
    if (tid < NQ) part_n[slot * NQ + tid] = sm.cnt[tid];
  }
}

// ---------------------------------------------------------------------------------------------
// merge of P partial top-k lists per query -> final top-k (also used after the all-gather)
//   in:  ps [P, nq_stride, kin] scores, pi ids (u32 local or i64 global), pn [P, nq_stride] counts
//        (pn == nullptr: an entry is valid iff its id >= 0)
//   out: D [nq, k], I [nq, k] (id_base added for u32 inputs); padding -FLT_MAX / -1.   k <= 64.
// Selection is exact and data-independent in cost: the order-encoded scores of all P*kin candidates sit in
// LDS; a 32-step bisection finds the k-th largest value V, a second bisection over ids resolves ties at V
// (ascending id), the exactly-k selected entries are compacted and rank-sorted.
// ---------------------------------------------------------------------------------------------
// RADIX select (round 6): the need-th largest of n order-encoded keys by three levels (11 + 11 + 10 bits) of an LDS histogram over the
// keys that share the prefix found so far -> that value V and how many of the entries equal to V belong to the selection.  Takes
// the place of the 32-step bisections (two barriers and a pass over the keys per step) of the merge and of the coarse quantiser.
// hist: 2048 words of LDS, sh: 2 words; every thread of the NT-thread workgroup calls it; returns with V / need_out valid in every thread and
// hist holding the last level's histogram (bin = key & 1023 among the keys that share V's upper 22 bits)
// SKIP0: key 0 marks an empty slot that is never counted (the caller guarantees need <= the number of non-zero keys): the hit lists the
// merge selects from are mostly empty slots, and 25 000 LDS atomics on ONE bin cost more than the rest of the kernel
template <int NT, class KeyFn, bool SKIP0 = false>
__device__ __forceinline__ void radix_select_block(KeyFn key, int n, unsigned need, unsigned* hist, unsigned* sh, unsigned& V,
                                                  unsigned& need_out) {
  const int tid = threadIdx.x;
  unsigned prefix = 0;  // keys with this prefix: `need` of them (the largest) are still to be taken
  for (int lvl = 0; lvl < 3; ++lvl) {
    const int shift = lvl == 0 ? 21 : (lvl == 1 ? 10 : 0);
    const int nb = lvl == 2 ? 1024 : 2048;
    const unsigned himask = lvl == 0 ? 0u : (lvl == 1 ? 0xffe00000u : 0xfffffc00u);
    for (int b = tid; b < 2048; b += NT) hist[b] = 0u;
    __syncthreads();
    for (int l = tid; l < n; l += NT) {
      const unsigned u = key(l);
      const bool part = (u & himask) == prefix && !(SKIP0 && u == 0u);
      const unsigned bin = (u >> shift) & (unsigned)(nb - 1);
      // scores crowd into a few bins (every hit of a list lies within a narrow band above its threshold): same-address LDS atomics
      // serialise -- 6 900 of them were 90 us of the merge.  The lanes that share the first participating lane's bin add ONE count.
      const unsigned long long m = __ballot(part);
      if (m != 0ull) {
        const int leader = __ffsll((long long)m) - 1;
        const unsigned b0 = (unsigned)__shfl((int)bin, leader);
        const unsigned long long same = __ballot(part && bin == b0);
        if (part && bin == b0) {
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
        } else if (part) {
          atomicAdd(&hist[bin], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {  // the bin b with count(bins > b) < need <= count(bins >= b): lane L owns bins [L per, (L + 1) per)
      const int per = nb / 64;
      unsigned mine = 0;
      for (int i = 0; i < per; ++i) mine += hist[tid * per + i];
      unsigned suf = mine;  // inclusive suffix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(suf, o);
        if (tid + o < 64) suf += t;
      }
      const unsigned above = suf - mine;
      if (above < need && need <= suf) {  // exactly one lane
        unsigned acc = above;
        int b = per - 1;
        for (; b > 0; --b) {
          const unsigned c = hist[tid * per + b];
          if (acc + c >= need) break;
          acc += c;
        }
        sh[0] = prefix | ((unsigned)(tid * per + b) << shift);
        sh[1] = need - acc;
      }
    }
    __syncthreads();
    prefix = sh[0];
    need = sh[1];
    __syncthreads();  // (hist is cleared / sh rewritten only after everyone has read them)
  }
  V = prefix;
  need_out = need;
}
__device__ __forceinline__ int block_count_1024(int v, int* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ int block_count_256(int v, int* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// 1024 threads per query (round 6; 256 before): the kernel is a chain of short passes over a few thousand candidates with a barrier
// between them, and a launch is one workgroup per query -- 64 workgroups of 256 threads left most of the chip idle for 43 us.
constexpr int MERGE_NT = 1024;
template <typename IdT>
__global__ __launch_bounds__(MERGE_NT) void knn_merge_kernel(const float* __restrict__ ps, const IdT* __restrict__ pi,
                                                       const int* __restrict__ pn, int P, int nq_stride, int kin,
                                                       int k, int64_t id_base, const int64_t* __restrict__ idmap,
                                                       float* __restrict__ D, int64_t* __restrict__ I,
                                                       const unsigned* __restrict__ gate, int blk_q,
                                                       const unsigned* __restrict__ blk_work, int nblk, int G) {
  if (gate && *gate == 0) return;
  // blk_q > 0 (multi-block scans): query blockIdx.x is query blockIdx.x % blk_q of block blockIdx.x / blk_q, whose P partial lists
  // start P * nq_stride lists into the arrays per block -- or, after a list scan (blk_work = the blocks' work-list lengths), are the
  // lists of the workgroups ivfm_range gave that block (P = the launch's upper bound, for the LDS size only)
  if (blk_q > 0) {
    const int b = (int)(blockIdx.x / (unsigned)blk_q);
    size_t first = (size_t)b * P;
    if (blk_work) {
      int s_, e_;
      ivfm_range(blk_work, nblk, G, b, s_, e_);
      first = (size_t)s_;
      P = e_ - s_;
    }
    ps += first * nq_stride * kin;
    pi += first * nq_stride * kin;
    if (pn) pn += first * nq_stride;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
  unsigned* s_u = reinterpret_cast<unsigned*>(merge_smem);             // [P*kin] order-encoded score, 0 = empty
  const int ncap = P * kin;
  long long* sel_i = reinterpret_cast<long long*>(s_u + ((ncap + 3) & ~3));  // [64], 16-B aligned
  float* sel_s = reinterpret_cast<float*>(sel_i + 64);                        // [64]
  int* red = reinterpret_cast<int*>(sel_s + 64);                              // [16] + counter
  const int qout = blockIdx.x, qq = blk_q > 0 ? (int)(blockIdx.x % (unsigned)blk_q) : (int)blockIdx.x, tid = threadIdx.x;

  auto gidx = [&](int c) -> size_t { return ((size_t)(c / kin) * nq_stride + qq) * kin + (c % kin); };
  // IVF: candidate ids are positions in the list-sorted arena; idmap gives the id the row was added with
  auto get_id = [&](int c) -> long long {
    return idmap ? (long long)idmap[(size_t)pi[gidx(c)]] : (long long)pi[gidx(c)] + id_base;
  };

  // ONE list of up to kin entries (the hit lists of the register-stationary scans: kin = 32 768, a few thousand filled): only the
  // filled part is walked -- by this loop and by every pass of the selection below.  (Walking all 32 768 slots with one dependent
  // load per iteration was 0.1 ms of a 0.15 ms kernel: profiles/r06x_knn_b256_timeline.log.)
  int ncand = ncap;
  if (P == 1 && pn) {
    const int nv = pn[qq];
    ncand = nv < ncap ? (nv > 0 ? nv : 0) : ncap;
  }
  int mine = 0;
  if (P == 1 && pn) {
    // every slot below ncand is filled: eight independent loads per thread in flight (the general loop below has two DEPENDENT global
    // round trips per iteration -- the list's length, then the score -- and was ~4 us per iteration)
    const float* pq = ps + (size_t)qq * kin;
    for (int c0 = tid; c0 < ncand; c0 += 8 * MERGE_NT) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = pq[c0 + MERGE_NT * i < ncand ? c0 + MERGE_NT * i : c0];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (c0 + MERGE_NT * i < ncand) {
          unsigned u = (unsigned)enc_f(v[i]) ^ 0x80000000u;
          if (u == 0) u = 1;
          s_u[c0 + MERGE_NT * i] = u;
          ++mine;
        }
      }
    }
  } else {
  // the P list lengths once, into the LDS (they were a global load in front of every score load: two dependent round trips per
  // iteration, 48 us for 128 lists of 40)
  __shared__ int s_pn[256];
  const bool pn_lds = pn && P <= 256;
  if (pn_lds) {
    if (tid < P) s_pn[tid] = pn[tid * nq_stride + qq];
    __syncthreads();
  }
#pragma unroll 4
  for (int c = tid; c < ncand; c += MERGE_NT) {
    const int p = c / kin, j = c - p * kin;
    const size_t g = ((size_t)p * nq_stride + qq) * kin + j;
    bool ok;
    if (pn_lds) ok = j < s_pn[p];
    else if (pn) ok = j < pn[p * nq_stride + qq];
    else ok = (long long)pi[g] >= 0;
    const float sc = ps[ok ? g : 0];             // (unconditional: the loads of the unrolled iterations fly together)
    unsigned u = 0;
    if (ok) {
      u = (unsigned)enc_f(sc) ^ 0x80000000u;     // unsigned order == float order
      if (u == 0) u = 1;                         // (only a -NaN payload; keeps 0 = "empty")
    }
    s_u[c] = u;
    mine += ok ? 1 : 0;
  }
  }
  if (tid == 0) red[16] = 0;
  const int ntot = block_count_1024(mine, red);
  const int kk = ntot < k ? ntot : k;
  if (kk > 0) {
    // V = kk-th largest encoded score (an empty slot is key 0 and kk <= the number of filled ones: V is a filled slot's key)
    __shared__ unsigned m_hist[2048];
    __shared__ unsigned m_sh[2];
    unsigned V, need;
    auto key_of = [&](int e) -> unsigned { return s_u[e]; };
    radix_select_block<MERGE_NT, decltype(key_of), true>(key_of, ncand, (unsigned)kk, m_hist, m_sh, V, need);
    const int ceq = (int)m_hist[V & 1023u];  // entries equal to V (the last level's histogram)
    const int t = (int)need;                 // entries to take among the ties at V, smallest ids first
    long long X = 0x7fffffffffffffffll;
    if (ceq > t) {
      X = 0;
      for (int bit = 62; bit >= 0; --bit) {
        const long long hi = X | ((1ll << bit) - 1);  // largest id with this prefix and the bit clear
        int c = 0;
        for (int e = tid; e < ncand; e += MERGE_NT)
          if (s_u[e] == V) c += get_id(e) <= hi ? 1 : 0;
        if (block_count_1024(c, red) < t) X |= (1ll << bit);
      }
    }
    // compact the exactly-kk selected entries
    for (int e = tid; e < ncand; e += MERGE_NT) {
      const unsigned u = s_u[e];
      bool take = u > V;
      long long id = 0;
      if (u >= V && u != 0) {
        id = get_id(e);
        if (u == V) take = id <= X;
      }
      if (take) {
        const int pos = atomicAdd(&red[16], 1);
        if (pos < 64) { sel_s[pos] = ps[gidx(e)]; sel_i[pos] = id; }
      }
    }
    __syncthreads();
    if (tid < kk) {
      const float se = sel_s[tid];
      const long long ie = sel_i[tid];
      int r = 0;
      for (int j = 0; j < kk; ++j) {
        const float sj = sel_s[j];
        const long long ij = sel_i[j];
        r += ((sj > se) || (sj == se && ij < ie)) ? 1 : 0;
      }
      D[(size_t)qout * k + r] = se;
      I[(size_t)qout * k + r] = ie;
    }
  }
  for (int j = kk + tid; j < k; j += MERGE_NT) { D[(size_t)qout * k + j] = -FLT_MAX; I[(size_t)qout * k + j] = -1; }
}

// ---------------------------------------------------------------------------------------------
// reconstruct: out[i, :] = f32(X[ids[i] - id_base, :]); id < 0 (or out of range) -> 0xFF bytes
// ---------------------------------------------------------------------------------------------
__global__ void knn_gather_rows_kernel(const _Float16* __restrict__ X, int64_t N, int d, int64_t id_base,
                                       const int64_t* __restrict__ ids, int64_t n, float* __restrict__ out) {
  const int64_t i = blockIdx.x;
  if (i >= n) return;
  const int64_t r = ids[i] - id_base;
  const bool ok = ids[i] >= 0 && r >= 0 && r < N;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    out[(size_t)i * d + c] = ok ? (float)X[(size_t)r * d + c] : __int_as_float(-1);
  }
}

__global__ void knn_f32_to_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (_Float16)in[i];
}

// ---------------------------------------------------------------------------------------------
// range_search post-pass: sort each query's hits by ascending id into the caller's CSR arrays
// (rank-by-counting; hit lists are short in every reference use: k<=3000 vectors, thresh 0.94)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_range_sort_kernel(const float* __restrict__ rs, const uint32_t* __restrict__ ri,
                                                            const unsigned* __restrict__ cnt, unsigned cap,
                                                            const int64_t* __restrict__ lims, int64_t id_base,
                                                            const int64_t* __restrict__ idmap, float* __restrict__ D,
                                                            int64_t* __restrict__ I, unsigned max_n) {
  // idmap (IVF: arena row -> id) or id_base + row; the hits of a query leave in ascending id order
  const int qq = blockIdx.x;
  const unsigned n = cnt[qq] < cap ? cnt[qq] : cap;
  if (n > max_n) return;  // long lists go through the radix sort below (rank-by-counting is O(n^2))
  const float* s = rs + (size_t)qq * cap;
  const uint32_t* id = ri + (size_t)qq * cap;
  const int64_t o = lims[qq];
  for (unsigned e = threadIdx.x; e < n; e += 256) {
    const int64_t ie = idmap ? idmap[id[e]] : (int64_t)id[e] + id_base;
    unsigned r = 0;
    for (unsigned j = 0; j < n; ++j) r += (idmap ? idmap[id[j]] : (int64_t)id[j] + id_base) < ie ? 1u : 0u;
    D[o + r] = s[e];
    I[o + r] = ie;
  }
}

// ---------------------------------------------------------------------------------------------
// range_search post-pass for LONG hit lists (> RANGE_SORT_SMALL hits of one query: a large-k search that descended to a low
// threshold, a range query over a dense neighbourhood): stable LSD radix sort by id, 4 bits per pass, one query at a time.
// Per pass: digit histogram of every 2048-element block -> exclusive scan over [digit][block] -> stable scatter (ranks inside
// a block by wave ballots: one ballot per digit value, waves ordered through an LDS table).  O(n) per pass instead of the
// O(n^2) rank-by-counting above (1e12 comparisons at 1 M hits).
// ---------------------------------------------------------------------------------------------
constexpr int RS_BLOCK = 2048;  // elements per 256-thread block
__global__ __launch_bounds__(256) void rs_keys_kernel(const uint32_t* __restrict__ ri, const int64_t* __restrict__ idmap,
                                                     int64_t id_base, unsigned n, uint32_t* __restrict__ keys) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = idmap ? (uint32_t)(idmap[ri[i]] - id_base) : ri[i];
}
__global__ __launch_bounds__(256) void rs_hist_kernel(const uint32_t* __restrict__ keys, unsigned n, int shift, unsigned nblk,
                                                     unsigned* __restrict__ hist) {
  __shared__ unsigned c[16];
  if (threadIdx.x < 16) c[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned base = blockIdx.x * RS_BLOCK;
  for (int r = 0; r < RS_BLOCK / 256; ++r) {
    const unsigned i = base + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&c[(keys[i] >> shift) & 15u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 16) hist[threadIdx.x * nblk + blockIdx.x] = c[threadIdx.x];
}
// single workgroup: in-place exclusive prefix sum over L entries
__global__ __launch_bounds__(1024) void rs_scan_kernel(unsigned* __restrict__ h, unsigned L) {
  __shared__ unsigned red[1024];
  __shared__ unsigned carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < L; base += 1024) {
    const unsigned l = base + tid;
    const unsigned v = l < L ? h[l] : 0u;
    red[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned t = tid >= o ? red[tid - o] : 0u;
      __syncthreads();
      red[tid] += t;
      __syncthreads();
    }
    if (l < L) h[l] = carry + red[tid] - v;
    __syncthreads();
    if (tid == 1023) carry += red[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint32_t* __restrict__ kin, const float* __restrict__ vin, unsigned n,
                                                        int shift, unsigned nblk, const unsigned* __restrict__ offs,
                                                        uint32_t* __restrict__ kout, float* __restrict__ vout) {
  __shared__ unsigned run[16];    // next output slot of every digit for this block
  __shared__ unsigned wc[4][16];  // this round's per-wave digit counts
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 16) run[tid] = offs[tid * nblk + blockIdx.x];
  __syncthreads();
  const unsigned base = blockIdx.x * RS_BLOCK;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  for (int r = 0; r < RS_BLOCK / 256; ++r) {
    const unsigned i = base + r * 256 + tid;
    const bool ok = i < n;
    const uint32_t key = ok ? kin[i] : 0u;
    const float val = ok ? vin[i] : 0.f;
    const int dg = ok ? (int)((key >> shift) & 15u) : -1;
    unsigned rank = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const unsigned long long m = __ballot(dg == v);
      if (dg == v) rank = (unsigned)__builtin_popcountll(m & lt);
      if (lane == 0) wc[w][v] = (unsigned)__builtin_popcountll(m);
    }
    __syncthreads();
    if (ok) {
      unsigned o = run[dg] + rank;
      for (int ww = 0; ww < w; ++ww) o += wc[ww][dg];
      kout[o] = key;
      vout[o] = val;
    }
    __syncthreads();
    if (tid < 16) run[tid] += wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void rs_emit_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ vals, unsigned n,
                                                     int64_t id_base, float* __restrict__ D, int64_t* __restrict__ I) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    D[i] = vals[i];
    I[i] = (int64_t)keys[i] + id_base;
  }
}
// one query's n hits (scores vs, arena rows / local ids ri) -> D, I sorted by ascending id.  Scratch: 2 key + 2 value buffers
// of n entries and 16 * ceil(n / 2048) counters, owned by the caller.  key_bits = bits of the largest local id.
hipError_t launch_range_sort_long(const float* vs, const uint32_t* ri, unsigned n, int64_t id_base, const int64_t* idmap, int key_bits,
                                  uint32_t* k0, uint32_t* k1, float* v0, float* v1, unsigned* hist, float* D, int64_t* I,
                                  hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned nb256 = (n + 255) / 256, nblk = (n + RS_BLOCK - 1) / RS_BLOCK;
  hipLaunchKernelGGL(rs_keys_kernel, dim3(nb256), dim3(256), 0, st, ri, idmap, id_base, n, k0);
  uint32_t* kbuf[2] = {k0, k1};
  float* vbuf[2] = {v0, v1};
  int cur = 0;
  const float* vin = vs;  // the scores are only read: the first pass moves them into the ping-pong buffers
  for (int shift = 0; shift < key_bits; shift += 4) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(256), 0, st, kbuf[cur], n, shift, nblk, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, hist, 16u * nblk);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblk), dim3(256), 0, st, kbuf[cur], vin, n, shift, nblk, hist, kbuf[cur ^ 1],
                       vbuf[cur ^ 1]);
    vin = vbuf[cur ^ 1];
    cur ^= 1;
  }
  hipLaunchKernelGGL(rs_emit_kernel, dim3(nb256), dim3(256), 0, st, kbuf[cur], vin, n, id_base, D, I);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// synthetic corpus (bench + parity at full scale): exactly re-derivable on the CPU.
//   v(r,c)   = sum of the four 16-bit fields of mix64(seed ^ ((r*d + c) * GOLD)) - 2*65535   (Irwin-Hall ~ normal)
//   x(r,c)   = fp16( fp32( v / sqrt(sum_c v^2) ) )   with the sqrt/divide in fp64 (IEEE exact on both sides)
// one wave per row
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ int synth_v(uint64_t seed, uint64_t idx) {
  const uint64_t h = mix64(seed ^ (idx * 0x9e3779b97f4a7c15ull));
  return (int)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48)) - 131070;
}

// dominant != 0 (corpus kind 2: "CLIP-like" anisotropy): columns 0 .. 2 carry 6 v + 113 511 (three standard deviations of v: a common
// offset plus a six-fold spread, as a few dimensions of real CLIP embeddings do) before the row is normalised -- the corpus on which
// one plain int8 plane fails; the first stage keeps those columns as 14-bit query digits (the 1-plane dominant-column form, DESIGN
// 4.3; two planes before round 5); oracle/knn_oracle.py: synth_rows(..., dominant=True).
constexpr int SYNTH_DOM_COLS = 3, SYNTH_DOM_GAIN = 6, SYNTH_DOM_OFFSET = 113511;
__global__ __launch_bounds__(256) void knn_synth_kernel(_Float16* __restrict__ X, int64_t row_begin, int64_t n, int d,
                                                       uint64_t seed, int dominant) {
  const int lane = threadIdx.x & 63;
  const int64_t r = row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= row_begin + n) return;
  constexpr int MAXE = 16;  // d <= 1024
  int v[MAXE];
  long long ss = 0;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = e * 64 + lane;
    v[e] = 0;
    if (c < d) {
      v[e] = synth_v(seed, (uint64_t)r * (uint64_t)d + (uint64_t)c);
      if (dominant && c < SYNTH_DOM_COLS) v[e] = SYNTH_DOM_GAIN * v[e] + SYNTH_DOM_OFFSET;
      ss += (long long)v[e] * (long long)v[e];
    }
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const double scale = 1.0 / sqrt((double)ss);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int c = e * 64 + lane;
    if (c < d) {
      float f = (float)((double)v[e] * scale);
      asm volatile("" : "+v"(f));  // keep the two roundings (f64->f32, f32->f16) separate: hipcc otherwise folds them
      X[(size_t)r * d + c] = (_Float16)f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// mixture corpus (BASELINE config 5: an index where IVF is meaningful), exactly re-derivable on the CPU
// (oracle/knn_oracle.py: synth_mixture_rows).  A mixture of `n_clusters` Gaussians in a MIX_M-dimensional latent space,
// embedded in R^d by a fixed random matrix, plus a little isotropic noise; integer arithmetic throughout:
//   s8(salt, i)  = sum of the four low bytes of mix64((seed ^ salt) ^ (i * GOLD)) - 510         (~ N(0, 148^2))
//   cluster(r)   = (((u1 * u2) >> 32) * n_clusters) >> 32,  u1 | u2 = the two halves of mix64((seed ^ SALT_C) ^ (r * GOLD)):
//                  the product of two uniforms -- component weights are skewed (the first cluster is ~ 1 + ln(n_clusters)
//                  times the average), like real embedding collections
//   l_j          = s8(MU, cluster * M + j) + (1 + cluster % 3) * s8(Z, r * M + j)              j < M: centre + spread x noise
//   v_c          = sum_j l_j * P[j][c] + 1024 * s8(N, r * d + c),  P[j][c] = s8(P, j * d + c)
//   x(r, c)      = fp16(fp32(v_c / sqrt(sum_c v_c^2)))   (sqrt / divide in fp64, as in knn_synth_kernel)
// The components overlap heavily (centre and within-component spread have the same scale), so the nearest neighbours of a
// point straddle several k-means cells: recall@40 of an IVF index over it rises with nprobe instead of being 1 at once
// (measured by tools/config5.py; a scaled-down numpy run of the same recipe gives 0.83 / 0.975 / 1.0 at nprobe 16 / 64 / 256
// of 256 lists).  One wave per row; the lane owns column pairs (2 lane, 2 lane + 1) + 128 e.
// ---------------------------------------------------------------------------------------------
constexpr int MIX_M = 32;
constexpr uint64_t MIX_SALT_P = 0x50u, MIX_SALT_C = 0xC1u, MIX_SALT_MU = 0x3Du, MIX_SALT_Z = 0x2Au, MIX_SALT_N = 0x4Eu;
__device__ __forceinline__ int synth_s8(uint64_t seed, uint64_t idx) {
  const uint64_t h = mix64(seed ^ (idx * 0x9e3779b97f4a7c15ull));
  return (int)((h & 0xff) + ((h >> 8) & 0xff) + ((h >> 16) & 0xff) + ((h >> 24) & 0xff)) - 510;
}
__global__ __launch_bounds__(256) void knn_mix_table_kernel(short* __restrict__ P, int d, uint64_t seed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < MIX_M * d) P[i] = (short)synth_s8(seed ^ MIX_SALT_P, (uint64_t)i);
}
__global__ __launch_bounds__(256) void knn_synth_mix_kernel(_Float16* __restrict__ X, int64_t row_begin, int64_t row_stride, int64_t n,
                                                           int d, uint64_t seed, uint64_t n_clusters, const short* __restrict__ P) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // destination row
  if (i >= n) return;
  const uint64_t r = (uint64_t)(row_begin + i * row_stride);       // corpus row
  const uint64_t hc = mix64((seed ^ MIX_SALT_C) ^ (r * 0x9e3779b97f4a7c15ull));
  const uint64_t t = ((hc & 0xffffffffull) * (hc >> 32)) >> 32;
  const uint64_t c = (t * n_clusters) >> 32;
  const int spread = 1 + (int)(c % 3);
  const int j = lane & (MIX_M - 1);
  const int lj = synth_s8(seed ^ MIX_SALT_MU, c * MIX_M + j) + spread * synth_s8(seed ^ MIX_SALT_Z, r * MIX_M + j);
  constexpr int MAXE = 8;  // d <= 1024: 8 column pairs per lane
  int v0[MAXE], v1[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { v0[e] = 0; v1[e] = 0; }
  for (int jj = 0; jj < MIX_M; ++jj) {
    const int l = __shfl(lj, jj);
    const int* prow = reinterpret_cast<const int*>(P + (size_t)jj * d);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int cp = e * 64 + lane;  // column pair index
      if (2 * cp < d) {
        const int pv = prow[cp];
        v0[e] += l * (int)(short)(pv & 0xffff);
        v1[e] += l * (pv >> 16);
      }
    }
  }
  long long ss = 0;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int col = 2 * (e * 64 + lane);
    if (col < d) {
      v0[e] += 1024 * synth_s8(seed ^ MIX_SALT_N, r * (uint64_t)d + (uint64_t)col);
      v1[e] += 1024 * synth_s8(seed ^ MIX_SALT_N, r * (uint64_t)d + (uint64_t)col + 1);
      ss += (long long)v0[e] * v0[e] + (long long)v1[e] * v1[e];
    }
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const double scale = 1.0 / sqrt((double)ss);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int col = 2 * (e * 64 + lane);
    if (col < d) {
      float f0 = (float)((double)v0[e] * scale), f1 = (float)((double)v1[e] * scale);
      asm volatile("" : "+v"(f0), "+v"(f1));  // keep the two roundings (f64->f32, f32->f16) separate
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      half2v o;
      o[0] = (_Float16)f0;
      o[1] = (_Float16)f1;
      *reinterpret_cast<half2v*>(X + (size_t)i * d + col) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wide-scan support: largest row norm of the index, exact re-scoring + proof, gates and selection
// ---------------------------------------------------------------------------------------------
// maxnorm (order-encoded float, atomicMax) = max over rows of |x|_2: the bound |<x, q_lo>| <= |q_lo| * maxnorm
__global__ __launch_bounds__(256) void knn_maxnorm_kernel(const _Float16* __restrict__ X, int64_t n, int d, int* __restrict__ maxnorm) {
  const int lane = threadIdx.x & 63;
  float best = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
    const half8* xr = reinterpret_cast<const half8*>(X + (size_t)r * d);
    float ss = 0.f;
    for (int c = lane; c < d / 8; c += 64) {
      const half8 v = xr[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    best = fmaxf(best, ss);
  }
  if (lane == 0) atomicMax(maxnorm, enc_f(sqrtf(best) * 1.0001f));
}

// One workgroup per query.  cand [nq, kw]: rows (id_base NOT added) of the kw best APPROXIMATE scores, approx [nq, kw]
// those scores (descending; -1 / -FLT_MAX padding).  Re-scores every candidate exactly (fp32 FMA of fp32(x) and the
// fp32 query), writes the top-k by (score desc, id asc) and need[q] = 1 when exactness cannot be proven:
//   every row outside the candidates has hi-score <= approx[kw-1], hence exact score <= approx[kw-1] + eps with
//   eps = |q - fp16(q)|_2 * maxnorm; if that is < the k-th exact score among the candidates, no outsider belongs to
//   the exact top-k.  (Fewer than kw candidates = the whole index was a candidate.)
__global__ __launch_bounds__(256) void knn_rescore_kernel(const _Float16* __restrict__ X, int d, const float* __restrict__ q,
                                                         const int64_t* __restrict__ cand, const float* __restrict__ approx,
                                                         int kw, int k, int64_t id_base, const int* __restrict__ maxnorm,
                                                         float* __restrict__ D, int64_t* __restrict__ I,
                                                         unsigned* __restrict__ need) {
  __shared__ float s_sc[64];
  __shared__ long long s_id[64];
  __shared__ float s_red[4], s_qn[4];
  const int qq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* qv = q + (size_t)qq * d;
  // |q - fp16(q)|^2 and |q|^2
  float e2 = 0.f, n2 = 0.f;
  for (int c = tid; c < d; c += 256) {
    const float v = qv[c];
    const float r = v - (float)(_Float16)v;
    e2 += r * r;
    n2 += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) { e2 += __shfl_xor(e2, o); n2 += __shfl_xor(n2, o); }
  if (lane == 0) { s_red[w] = e2; s_qn[w] = n2; }
  for (int j = w; j < kw; j += 4) {
    const int64_t row = cand[(size_t)qq * kw + j];
    float acc = 0.f;
    if (row >= 0) {
      const _Float16* xr = X + (size_t)row * d;
      for (int c = lane; c < d; c += 64) acc = __builtin_fmaf((float)xr[c], qv[c], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      s_sc[j] = row >= 0 ? acc : -FLT_MAX;
      s_id[j] = row >= 0 ? row + id_base : -1;
    }
  }
  __syncthreads();
  if (tid < kw) {
    const float se = s_sc[tid];
    const long long ie = s_id[tid];
    int r = 0, nvalid = 0;
    for (int j = 0; j < kw; ++j) {
      const float sj = s_sc[j];
      const long long ij = s_id[j];
      nvalid += ij >= 0 ? 1 : 0;
      if (ij >= 0 && ie >= 0) r += ((sj > se) || (sj == se && ij < ie)) ? 1 : 0;
      else if (ij >= 0 && ie < 0) r += 1;  // invalid entries sort last
      else if (ij < 0 && ie < 0) r += j < tid ? 1 : 0;
    }
    if (r < k) {
      D[(size_t)qq * k + r] = ie >= 0 ? se : -FLT_MAX;
      I[(size_t)qq * k + r] = ie;
    }
    if (r == (k < nvalid ? k : nvalid) - 1 || (nvalid == 0 && tid == 0)) {
      // this thread holds the k-th best exact score (or the last valid one)
      // eps bounds |approx - exact| for every row: the fp16 rounding of the query (|q - fp16(q)| * |x|) plus the
      // accumulation-order difference between the MFMA score and the FMA re-score (<= d * 2^-24 * |q| * |x| each)
      const float eps = (sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]) + (float)d * 1.2e-7f * sqrtf(s_qn[0] + s_qn[1] + s_qn[2] + s_qn[3])) * dec_f(*maxnorm);
      const bool all_in = nvalid < kw;  // the index has fewer than kw rows for this query: nothing is outside
      const float a_last = approx[(size_t)qq * kw + kw - 1];
      const bool proven = all_in || (nvalid >= k && a_last + eps < se);
      need[qq] = proven ? 0u : 1u;
    }
  }
}

// gate[b] = any(need[32b .. 32b+31]) for the two 32-query halves of a wide scan
// stats (may be null): [0] += queries served, [1] += proofs that failed (read by knnx_get_stats)
__global__ void knn_gates_kernel(const unsigned* __restrict__ need, int nq, unsigned* __restrict__ gate,
                                 unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x;  // 64 threads
  const unsigned v = lane < nq ? need[lane] : 0u;
  const unsigned long long m = __ballot(v != 0);
  if (lane == 0) {
    gate[0] = (unsigned)(m & 0xffffffffull) ? 1u : 0u;
    gate[1] = (unsigned)(m >> 32) ? 1u : 0u;
    if (stats) {
      atomicAdd(&stats[0], (unsigned long long)nq);
      atomicAdd(&stats[1], (unsigned long long)__builtin_popcountll(m));
    }
  }
}

// rows of queries with need[q] != 0 are replaced by the exact-scan result of their half
__global__ void knn_select_kernel(const unsigned* __restrict__ need, int q0, int nq, int k, const float* __restrict__ Dfb,
                                  const int64_t* __restrict__ Ifb, float* __restrict__ D, int64_t* __restrict__ I) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  const int qq = i / k;
  if (need[q0 + qq]) {
    D[(size_t)(q0 + qq) * k + (i - qq * k)] = Dfb[i];
    I[(size_t)(q0 + qq) * k + (i - qq * k)] = Ifb[i];
  }
}

// ---------------------------------------------------------------------------------------------
// IVF-Flat (faiss IndexIVFFlat semantics, inner product): coarse quantiser = the same flat scan over the centroid
// matrix; the kernels below turn its [nq, nprobe] list ids into the work list the IVF scan walks.
//   list l occupies tiles [tile0[l], tile0[l] + ntile[l]) of the list-sorted, tile-padded arena; size[l] rows are real.
// ---------------------------------------------------------------------------------------------
// (more than 32 queries -- the multi-block pass: query q marks bit q % 32 of masks[q / 32][l])
__global__ void ivf_mark_kernel(const int64_t* __restrict__ Ic, int nq, int nprobe, int nlist, unsigned* __restrict__ masks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * nprobe) return;
  const int64_t l = Ic[i];
  const int qi = i / nprobe;
  if (l >= 0 && l < nlist) atomicOr(&masks[(size_t)(qi >> 5) * nlist + l], 1u << (qi & 31));
}

// Work list of a query block b (blockIdx.y), SORTED (list order = arena order: the workgroups of a scan then walk neighbouring tiles at
// the same time -- 3 - 5 % on the long scans against an arbitrary order), in two small kernels and no scan over all lists:
//   ivf_wgsum_kernel   workgroup w of block b: the tiles of the probed lists among its 256 -> wgsum[b][w]
//   ivf_expand_kernel  workgroup w: base = sum of wgsum[b][0 .. w) (every workgroup adds them up itself: a few hundred words), position
//                      of a list = base + the exclusive prefix inside the workgroup (shuffles + the four wave totals); then the WAVE
//                      writes the tiles of each probed list among its 64 (lane t -> tile t, t + 64, ..: whole lines of the work list);
//                      the last workgroup leaves the length of the list in nwork[b].
// (Rounds 3 - 5: a single-workgroup prefix sum over all 65 536 lists, 57 - 180 us in four variants, + one workgroup per list, 30 us;
// one atomic per probed list on the block's tile counter: 15 us at 512 probed lists, 50 us at 2 048, and an arbitrary order.)
__global__ __launch_bounds__(256) void ivf_wgsum_kernel(const unsigned* __restrict__ masks, const unsigned* __restrict__ ntile, int nlist,
                                                       unsigned* __restrict__ wgsum) {
  __shared__ unsigned ws[4];
  const int l = blockIdx.x * 256 + threadIdx.x;
  const size_t b = blockIdx.y;
  unsigned v = (l < nlist && masks[b * nlist + l]) ? ntile[l] : 0u;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) wgsum[b * gridDim.x + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void ivf_expand_kernel(const unsigned* __restrict__ masks, const unsigned* __restrict__ tile0,
                                                        const unsigned* __restrict__ ntile, const unsigned* __restrict__ size,
                                                        unsigned* __restrict__ nwork, uint4* __restrict__ work,
                                                        unsigned work_stride, int nlist, const unsigned* __restrict__ wgsum) {
  __shared__ unsigned red[4], wtot[4];
  const int l = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t b = blockIdx.y;
  const unsigned m = l < nlist ? masks[b * nlist + l] : 0u;
  const unsigned nt = m ? ntile[l] : 0u, t0 = m ? tile0[l] : 0u, sz = m ? size[l] : 0u;
  // tiles in front of this workgroup
  unsigned part = 0;
  for (unsigned i = threadIdx.x; i < blockIdx.x; i += 256) part += wgsum[b * gridDim.x + i];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  // exclusive prefix of the tiles inside the workgroup
  unsigned inc = nt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 0) red[wv] = part;
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  const unsigned base = red[0] + red[1] + red[2] + red[3];
  unsigned before = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) before += i < wv ? wtot[i] : 0u;
  const unsigned o = base + before + inc - nt;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) nwork[b] = o + nt;
  unsigned long long probed = __ballot(m != 0u);
  work += b * work_stride;
  while (probed) {
    const int src = __ffsll((long long)probed) - 1;
    probed &= probed - 1;
    const unsigned mm = (unsigned)__shfl((int)m, src), nn = (unsigned)__shfl((int)nt, src), tt = (unsigned)__shfl((int)t0, src),
                   ss = (unsigned)__shfl((int)sz, src), oo = (unsigned)__shfl((int)o, src);
    for (unsigned t = lane; t < nn; t += 64) {
      const unsigned valid = (t + 1 < nn) ? 32u : ss - 32u * (nn - 1);
      work[oo + t] = make_uint4(tt + t, mm, valid, 0u);
    }
  }
}

// one workgroup per list: copy its rows from the unpadded (list-grouped) arena to the tile-padded one, zero the pad
// rows, and lay down idmap (-1 on pad rows) and the inverse map id -> padded row
__global__ __launch_bounds__(256) void ivf_relayout_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int d,
                                                          const int64_t* __restrict__ src0, const unsigned* __restrict__ tile0,
                                                          const unsigned* __restrict__ ntile, const unsigned* __restrict__ size,
                                                          const int64_t* __restrict__ ids, int64_t id_lo, int64_t n_ids,
                                                          int64_t* __restrict__ idmap, uint32_t* __restrict__ inv) {
  const int l = blockIdx.x;
  const int64_t s0 = src0[l];
  const size_t r0 = (size_t)tile0[l] * 32, nrow = (size_t)ntile[l] * 32, sz = size[l];
  const int d8 = d / 8;
  const uint4* sp = reinterpret_cast<const uint4*>(src + (size_t)s0 * d);
  uint4* dp = reinterpret_cast<uint4*>(dst + r0 * d);
  for (size_t i = threadIdx.x; i < nrow * d8; i += 256) dp[i] = (i / d8) < sz ? sp[i] : make_uint4(0u, 0u, 0u, 0u);
  for (size_t r = threadIdx.x; r < nrow; r += 256) {
    const int64_t id = r < sz ? ids[s0 + r] : -1;
    idmap[r0 + r] = id;
    if (id >= id_lo && id - id_lo < n_ids) inv[id - id_lo] = (uint32_t)(r0 + r);
  }
}

// reconstruct for IVF: ids -> padded arena rows through the inverse map
__global__ void knn_gather_rows_inv_kernel(const _Float16* __restrict__ X, int d, int64_t id_lo, int64_t n_ids,
                                           const uint32_t* __restrict__ inv, const int64_t* __restrict__ ids, int64_t n,
                                           float* __restrict__ out) {
  const int64_t i = blockIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  const bool ok = id >= id_lo && id - id_lo < n_ids;
  const size_t r = ok ? inv[id - id_lo] : 0;
  for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)i * d + c] = ok ? (float)X[r * d + c] : __int_as_float(-1);
}

// Coarse quantiser from a score matrix: scores [nq, nlist] (dumped by the MODE 2 scan over the centroids) -> bit q % 32 of
// masks[q / 32][l] for the nprobe best lists of query q, (score desc, list id asc) -- the selection rule of the top-k path.  One
// 1024-thread workgroup per query.
//   * radix_select_1024: the need-th largest of n order-encoded keys by three levels (11 + 11 + 10 bits) of an LDS histogram over the
//     keys that share the prefix found so far -> that value V and how many of the entries equal to V belong to the selection.
//   * fast path (nprobe <= 512): T0 = the nprobe-th largest of the 1024 per-thread maxima (a radix select over 1024 keys: one LDS
//     atomic per thread and level) is a lower bound of V -- nprobe distinct entries are >= T0 --, so only the entries >= T0 can be
//     selected: a few hundred at most for random scores (~1.1 nprobe at nprobe 16, ~1.4 nprobe at 512).  They are collected in the
//     LDS and ranked against each other.  Two reads of the score row (256 KiB at nlist 65 536, the second from the L2).
//   * general path (more than SV_CAP entries >= T0: duplicate / zero centroids; nprobe > 512): the radix select over the whole row
//     (65 536 LDS atomics per level into a handful of hot bins: ~17 us per level), ties at V resolved by a bisection over list
//     ids.  The 32-step bisection over scores this replaces read the row 33 times (the nprobe > 64 path of rounds 3-5).
__global__ __launch_bounds__(1024) void ivf_select_mark_kernel(const float* __restrict__ scores, int nlist, int nprobe,
                                                              unsigned* __restrict__ masks) {
  constexpr int SV_CAP = 2048;
  __shared__ unsigned hist[2048];
  __shared__ unsigned sv_key[SV_CAP];
  __shared__ int sv_id[SV_CAP];
  __shared__ unsigned sh[2];
  __shared__ int sh_cnt;
  __shared__ int red[16];
  const int qq = blockIdx.x, tid = threadIdx.x;
  const float* s = scores + (size_t)qq * nlist;
  unsigned* mk = masks + (size_t)(qq >> 5) * nlist;
  const unsigned bit = 1u << (qq & 31);
  auto enc = [](float f) -> unsigned { return (unsigned)enc_f(f) ^ 0x80000000u; };  // unsigned order == float order
  const int np = nprobe < nlist ? nprobe : nlist;
  if (np <= 512) {
    unsigned mx = 0;
    for (int l = tid; l < nlist; l += 1024) {
      const unsigned u = enc(s[l]);
      mx = u > mx ? u : mx;
    }
    sv_key[tid] = mx;  // (the survivor arrays are free until T0 is known)
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    unsigned T0, unused;
    radix_select_block<1024>([&](int i) -> unsigned { return sv_key[i]; }, 1024, (unsigned)np, hist, sh, T0, unused);
    for (int l = tid; l < nlist; l += 1024) {
      const unsigned u = enc(s[l]);
      if (u >= T0) {
        const int pos = atomicAdd(&sh_cnt, 1);
        if (pos < SV_CAP) { sv_key[pos] = u; sv_id[pos] = l; }
      }
    }
    __syncthreads();
    const int S = sh_cnt;
    if (S <= SV_CAP) {
      for (int i = tid; i < S; i += 1024) {
        const unsigned ki = sv_key[i];
        const int li = sv_id[i];
        int r = 0;
        for (int j = 0; j < S; ++j) {
          const unsigned kj = sv_key[j];
          r += (kj > ki || (kj == ki && sv_id[j] < li)) ? 1 : 0;
        }
        if (r < np) atomicOr(&mk[li], bit);
      }
      return;
    }
    __syncthreads();  // (everyone has read sh_cnt; the general path reuses the LDS)
  }
  unsigned V, need;
  radix_select_block<1024>([&](int l) -> unsigned { return enc(s[l]); }, nlist, (unsigned)np, hist, sh, V, need);
  const int t = (int)need;                   // entries to take among the ties at V, smallest list ids first
  const int ceq = (int)hist[V & 1023u];      // entries equal to V (the last level's histogram)
  int X = 0x7fffffff;
  if (ceq > t) {
    X = 0;
    for (int b = 30; b >= 0; --b) {
      const int hi = X | ((1 << b) - 1);
      int c = 0;
      for (int l = tid; l < nlist; l += 1024) c += (enc(s[l]) == V && l <= hi) ? 1 : 0;
      if (block_count_1024(c, red) < t) X |= (1 << b);
    }
  }
  for (int l = tid; l < nlist; l += 1024) {
    const unsigned u = enc(s[l]);
    if (u > V || (u == V && l <= X)) atomicOr(&mk[l], bit);
  }
}

// Lloyd update of the IVF k-means: one workgroup per list sums its member rows (order[] = the sample rows sorted by list,
// ascending row id inside a list: a fixed summation order) in fp32 and writes the mean, SCALED TO UNIT NORM, as the new fp16
// centroid (spherical k-means: with inner-product assignment an un-normalised mean of spread-out members is short and loses
// its points to any long centroid -- measured on config 5's corpus: median list 17 rows of an average of 1 907, a few lists of
// 45 k; normalised centroids make the argmax a cosine assignment and the lists balanced).
__global__ __launch_bounds__(256) void kmeans_update_kernel(const _Float16* __restrict__ X, int d, const int64_t* __restrict__ order,
                                                           const int64_t* __restrict__ off, _Float16* __restrict__ cent) {
  __shared__ float red[4];
  const int l = blockIdx.x;
  const int64_t a = off[l], b = off[l + 1];
  if (b <= a) return;  // empty list: the caller re-seeds it
  float m[4] = {0.f, 0.f, 0.f, 0.f};  // d <= 1024: up to 4 columns per thread
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = threadIdx.x + 256 * e;
    if (c < d) {
      float s = 0.f;
      for (int64_t i = a; i < b; ++i) s += (float)X[(size_t)order[i] * d + c];
      m[e] = s;
      ss += s * s;
    }
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = threadIdx.x + 256 * e;
    if (c < d) cent[(size_t)l * d + c] = (_Float16)(m[e] * inv);
  }
}
hipError_t launch_kmeans_update(const _Float16* X, int d, const int64_t* order, const int64_t* off, int nlist, _Float16* cent,
                                hipStream_t st) {
  if (d > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kmeans_update_kernel, dim3(nlist), dim3(256), 0, st, X, d, order, off, cent);
  return hipGetLastError();
}

// one wave per row: copy into its slot of the list-sorted arena
__global__ __launch_bounds__(256) void ivf_scatter_kernel(const _Float16* __restrict__ src, int64_t n, int d,
                                                         const int32_t* __restrict__ lists, const int32_t* __restrict__ pos,
                                                         const int64_t* __restrict__ ids, int64_t id0,
                                                         const unsigned* __restrict__ tile0, int64_t id_lo, int64_t n_ids,
                                                         _Float16* __restrict__ dst,
                                                         int64_t* __restrict__ idmap, uint32_t* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const size_t drow = (size_t)tile0[lists[r]] * 32 + (size_t)pos[r];
  const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)r * d);
  uint4* o = reinterpret_cast<uint4*>(dst + drow * d);
  for (int c = lane; c < d / 8; c += 64) o[c] = s[c];
  if (lane == 0) {
    const int64_t id = ids ? ids[r] : id0 + r;  // ids == null: consecutive ids id0, id0 + 1, ... (a device-resident chunk of the corpus)
    idmap[drow] = id;
    if (id >= id_lo && id - id_lo < n_ids) inv[id - id_lo] = (uint32_t)drow;
  }
}
hipError_t launch_ivf_scatter(const _Float16* src, int64_t n, int d, const int32_t* lists, const int32_t* pos, const int64_t* ids,
                              int64_t id0, const unsigned* tile0, int64_t id_lo, int64_t n_ids, _Float16* dst, int64_t* idmap,
                              uint32_t* inv, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ivf_scatter_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src, n, d, lists, pos, ids, id0, tile0, id_lo,
                     n_ids, dst, idmap, inv);
  return hipGetLastError();
}

// dst[dst_rows[i], :] = src[src_rows[i], :] (k-means seeding: centroid <- sample row); one wave per pair
__global__ __launch_bounds__(256) void copy_rows_kernel(const _Float16* __restrict__ src, int d, const int64_t* __restrict__ src_rows,
                                                       const int32_t* __restrict__ dst_rows, int64_t n, _Float16* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)src_rows[i] * d);
  uint4* o = reinterpret_cast<uint4*>(dst + (size_t)dst_rows[i] * d);
  for (int c = lane; c < d / 8; c += 64) o[c] = s[c];
}
hipError_t launch_copy_rows(const _Float16* src, int d, const int64_t* src_rows, const int32_t* dst_rows, int64_t n, _Float16* dst,
                            hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src, d, src_rows, dst_rows, n, dst);
  return hipGetLastError();
}

// histogram of list ids (the list sizes knnx_ivf_begin needs), accumulated over the chunks of an assignment pass
__global__ __launch_bounds__(256) void ivf_hist_kernel(const int32_t* __restrict__ lists, int64_t n, int nlist,
                                                      unsigned long long* __restrict__ hist) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int l = lists[i];
    if (l >= 0 && l < nlist) atomicAdd(&hist[l], 1ull);
  }
}
hipError_t launch_ivf_hist(const int32_t* lists, int64_t n, int nlist, unsigned long long* hist, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ivf_hist_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, lists, n, nlist, hist);
  return hipGetLastError();
}

// nq > 32 (the multi-block pass): masks / off are [nblk, nlist], nwork [nblk], work [nblk, work_stride], nblk = ceil(nq / 32)
hipError_t launch_ivf_worklist_from_scores(const float* scores, int nq, int nprobe, int nlist, unsigned* masks, const unsigned* tile0,
                                           const unsigned* ntile, const unsigned* size, unsigned* off, uint4* work, unsigned* nwork,
                                           hipStream_t st, unsigned work_stride) {
  const int nblk = (nq + 31) / 32;
  hipError_t e = hipMemsetAsync(masks, 0, (size_t)nblk * nlist * sizeof(unsigned), st);  // (nwork is written by ivf_expand_kernel)
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ivf_select_mark_kernel, dim3(nq), dim3(1024), 0, st, scores, nlist, nprobe, masks);
  // (`off` [nblk][nlist] holds the per-workgroup tile sums: [nblk][ceil(nlist / 256)] of it)
  hipLaunchKernelGGL(ivf_wgsum_kernel, dim3((nlist + 255) / 256, nblk), dim3(256), 0, st, masks, ntile, nlist, off);
  hipLaunchKernelGGL(ivf_expand_kernel, dim3((nlist + 255) / 256, nblk), dim3(256), 0, st, masks, tile0, ntile, size, nwork, work, work_stride, nlist, off);
  return hipGetLastError();
}

hipError_t launch_ivf_worklist(const int64_t* Ic, int nq, int nprobe, int nlist, unsigned* masks, const unsigned* tile0,
                               const unsigned* ntile, const unsigned* size, unsigned* off, uint4* work, unsigned* nwork,
                               hipStream_t st, unsigned work_stride) {
  const int nblk = (nq + 31) / 32;
  hipError_t e = hipMemsetAsync(masks, 0, (size_t)nblk * nlist * sizeof(unsigned), st);  // (nwork is written by ivf_expand_kernel)
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ivf_mark_kernel, dim3((nq * nprobe + 255) / 256), dim3(256), 0, st, Ic, nq, nprobe, nlist, masks);
  // (`off` [nblk][nlist] holds the per-workgroup tile sums: [nblk][ceil(nlist / 256)] of it)
  hipLaunchKernelGGL(ivf_wgsum_kernel, dim3((nlist + 255) / 256, nblk), dim3(256), 0, st, masks, ntile, nlist, off);
  hipLaunchKernelGGL(ivf_expand_kernel, dim3((nlist + 255) / 256, nblk), dim3(256), 0, st, masks, tile0, ntile, size, nwork, work, work_stride, nlist, off);
  return hipGetLastError();
}
// tiles of the lists at least one of the nblk query blocks probes -> *out
__global__ __launch_bounds__(256) void ivf_union_tiles_kernel(const unsigned* __restrict__ masks, int nblk, int nlist,
                                                             const unsigned* __restrict__ ntile, unsigned* __restrict__ out) {
  unsigned t = 0;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < nlist; l += gridDim.x * 256) {
    unsigned m = 0;
    for (int b = 0; b < nblk; ++b) m |= masks[(size_t)b * nlist + l];
    if (m) t += ntile[l];
  }
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if ((threadIdx.x & 63) == 0 && t) atomicAdd(out, t);
}
hipError_t launch_ivf_union_tiles(const unsigned* masks, int nblk, int nlist, const unsigned* ntile, unsigned* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ivf_union_tiles_kernel, dim3(std::min((nlist + 255) / 256, 256)), dim3(256), 0, st, masks, nblk, nlist, ntile, out);
  return hipGetLastError();
}
hipError_t launch_ivf_relayout(const _Float16* src, _Float16* dst, int d, int nlist, const int64_t* src0, const unsigned* tile0,
                               const unsigned* ntile, const unsigned* size, const int64_t* ids, int64_t id_lo, int64_t n_ids,
                               int64_t* idmap, uint32_t* inv, hipStream_t st) {
  hipLaunchKernelGGL(ivf_relayout_kernel, dim3(nlist), dim3(256), 0, st, src, dst, d, src0, tile0, ntile, size, ids, id_lo,
                     n_ids, idmap, inv);
  return hipGetLastError();
}
hipError_t launch_gather_inv(const _Float16* X, int d, int64_t id_lo, int64_t n_ids, const uint32_t* inv, const int64_t* ids,
                             int64_t n, float* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(knn_gather_rows_inv_kernel, dim3((unsigned)n), dim3(256), 0, st, X, d, id_lo, n_ids, inv, ids, n, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (declared in knn_kernels.h)
// ---------------------------------------------------------------------------------------------
size_t scan_smem_bytes(int d, int cap, int nqs) { return (size_t)d * 128 + (size_t)nqs * cap * 8 + nqs * 8 + 16; }

hipError_t launch_prep(const float* q_dev, int nq, int d, _Float16* qfrag, int* thr_g, unsigned* range_cnt, int wide,
                       const unsigned* gate, hipStream_t st) {
  hipLaunchKernelGGL(knn_prep_queries_kernel, dim3(d / 16), dim3(64), 0, st, q_dev, nq, d, qfrag, thr_g, range_cnt, wide, gate,
                     (int*)nullptr);
  return hipGetLastError();
}
// ceil(nq / 32) blocks of 32 queries: fragment images [nblk][d * 64 halves], thresholds thr_a / thr_b [32 nblk] reset
hipError_t launch_prep_blocks(const float* q_dev, int nq, int d, _Float16* qfrag, int* thr_a, int* thr_b, hipStream_t st, int wide) {
  const int qpb = wide ? KNN_NQ_MAX : KNN_NQ;
  hipLaunchKernelGGL(knn_prep_queries_kernel, dim3(d / 16, (nq + qpb - 1) / qpb), dim3(64), 0, st, q_dev, nq, d, qfrag, thr_a,
                     (unsigned*)nullptr, wide ? 1 : 0, (const unsigned*)nullptr, thr_b);
  return hipGetLastError();
}
hipError_t launch_maxnorm(const _Float16* X, int64_t n, int d, int* maxnorm, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = std::min<int64_t>((n + 3) / 4, 8192);
  hipLaunchKernelGGL(knn_maxnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, n, d, maxnorm);
  return hipGetLastError();
}
hipError_t launch_rescore(const _Float16* X, int d, const float* q, const int64_t* cand, const float* approx, int nq, int kw,
                          int k, int64_t id_base, const int* maxnorm, float* D, int64_t* I, unsigned* need, unsigned* gate,
                          unsigned long long* stats, hipStream_t st) {
  hipLaunchKernelGGL(knn_rescore_kernel, dim3(nq), dim3(256), 0, st, X, d, q, cand, approx, kw, k, id_base, maxnorm, D, I, need);
  hipLaunchKernelGGL(knn_gates_kernel, dim3(1), dim3(64), 0, st, need, nq, gate, stats);
  return hipGetLastError();
}
hipError_t launch_select(const unsigned* need, int q0, int nq, int k, const float* Dfb, const int64_t* Ifb, float* D, int64_t* I,
                         hipStream_t st) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(knn_select_kernel, dim3((nq * k + 255) / 256), dim3(256), 0, st, need, q0, nq, k, Dfb, Ifb, D, I);
  return hipGetLastError();
}

template <int MODE, bool NT, bool IVF = false, int QB = 1>
static hipError_t launch_scan_mode(const ScanArgs& a, hipStream_t st) {
  const size_t smem = scan_smem_bytes(a.d, a.cap, 32 * QB);
#define KNN_LAUNCH(NCH)                                                                                         \
  {                                                                                                             \
    auto kern = knn_scan_kernel<NCH, MODE, NT, IVF, QB>;                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                  \
    if (e != hipSuccess) return e;                                                                              \
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(KNN_WG), smem, st, a.X, a.N, a.qfrag, a.nq, a.k, a.cap,         \
                       a.thr_g, a.part_s, a.part_i, a.part_n, a.range_thr, a.range_cnt, a.range_cap, a.range_s, \
                       a.range_i, a.work, a.nwork, a.gate, a.tstride > 1 ? a.tstride : 1, a.nblk > 1 ? a.nblk : 1, \
                       a.work_stride);                                                                           \
    return hipGetLastError();                                                                                   \
  }
  switch (a.d) {
    case 256: KNN_LAUNCH(2)
    case 512: KNN_LAUNCH(4)
    case 768: KNN_LAUNCH(6)
    case 1024: KNN_LAUNCH(8)
    default: return hipErrorInvalidValue;
  }
#undef KNN_LAUNCH
}

hipError_t launch_scan(const ScanArgs& a, hipStream_t st) {
  if (a.nblk > 1 && (a.mode == 1 || a.grid % a.nblk != 0)) return hipErrorInvalidValue;
  if (a.wide) return (a.mode == 0 && !a.work) ? launch_scan_mode<0, false, false, 2>(a, st) : hipErrorInvalidValue;
  if (a.work) {  // IVF work list: top-k (mode 0) or range (mode 1) over the rows of the probed lists
    if (a.mode == 0) return launch_scan_mode<0, false, true>(a, st);
    return a.mode == 1 ? launch_scan_mode<1, false, true>(a, st) : hipErrorInvalidValue;
  }
  if (a.mode == 0) return a.nt ? launch_scan_mode<0, true>(a, st) : launch_scan_mode<0, false>(a, st);
  if (a.mode == 2) return launch_scan_mode<2, false>(a, st);
  return launch_scan_mode<1, false>(a, st);
}

hipError_t launch_merge_u32(const float* ps, const uint32_t* pi, const int* pn, int P, int nq_stride, int kin,
                            int nq, int k, int64_t id_base, const int64_t* idmap, float* D, int64_t* I,
                            const unsigned* gate, hipStream_t st, int blk_q, const unsigned* blk_work, int nblk, int G) {
  if (k > 64) return hipErrorInvalidValue;
  const size_t smem = (size_t)((P * kin + 3) & ~3) * 4 + 64 * 12 + 96;
  auto kern = knn_merge_kernel<uint32_t>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nq), dim3(MERGE_NT), smem, st, ps, pi, pn, P, nq_stride, kin, k, id_base, idmap, D, I, gate, blk_q, blk_work,
                     nblk, G);
  return hipGetLastError();
}
hipError_t launch_merge_i64(const float* ps, const int64_t* pi, int P, int nq, int kin, int k, float* D,
                            int64_t* I, hipStream_t st) {
  if (k > 64) return hipErrorInvalidValue;
  const size_t smem = (size_t)((P * kin + 3) & ~3) * 4 + 64 * 12 + 96;
  if (smem + 9472 > (size_t)KNN_LDS_BYTES) return hipErrorInvalidValue;  // (+ the kernel's static LDS: radix histogram, list lengths)
  auto kern = knn_merge_kernel<int64_t>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nq), dim3(MERGE_NT), smem, st, ps, pi, (const int*)nullptr, P, nq, kin, k, (int64_t)0,
                     (const int64_t*)nullptr, D, I, (const unsigned*)nullptr, 0, (const unsigned*)nullptr, 0, 0);
  return hipGetLastError();
}
// P-way merge of P sorted lists per query (score desc, id asc; id < 0 = padding at the tail of a list) -> the sorted top-k.
// One thread per query, k * P steps: the large-k (k > 64) exchange of the in-process sharded index (rare: the front-end's
// num_result_ids = 3000 requests), where a handful of queries carry a few thousand results each.
__global__ void knn_merge_sorted_kernel(const float* __restrict__ Dp, const int64_t* __restrict__ Ip, int P, int n, int k,
                                        float* __restrict__ D, int64_t* __restrict__ I) {
  const int qq = blockIdx.x * blockDim.x + threadIdx.x;
  if (qq >= n) return;
  int head[64];
  for (int p = 0; p < P; ++p) head[p] = 0;
  for (int j = 0; j < k; ++j) {
    int best = -1;
    float bs = 0.f;
    int64_t bi = 0;
    for (int p = 0; p < P; ++p) {
      if (head[p] >= k) continue;
      const size_t o = ((size_t)p * n + qq) * k + head[p];
      const int64_t id = Ip[o];
      if (id < 0) continue;
      const float sc = Dp[o];
      if (best < 0 || sc > bs || (sc == bs && id < bi)) { best = p; bs = sc; bi = id; }
    }
    if (best < 0) {
      D[(size_t)qq * k + j] = -FLT_MAX;
      I[(size_t)qq * k + j] = -1;
    } else {
      D[(size_t)qq * k + j] = bs;
      I[(size_t)qq * k + j] = bi;
      head[best]++;
    }
  }
}
hipError_t launch_merge_sorted(const float* Dp, const int64_t* Ip, int P, int n, int k, float* D, int64_t* I, hipStream_t st) {
  if (P <= 0 || P > 64) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(knn_merge_sorted_kernel, dim3((n + 63) / 64), dim3(64), 0, st, Dp, Ip, P, n, k, D, I);
  return hipGetLastError();
}

hipError_t launch_gather(const _Float16* X, int64_t N, int d, int64_t id_base, const int64_t* ids, int64_t n,
                         float* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, X, N, d, id_base, ids, n, out);
  return hipGetLastError();
}
hipError_t launch_f32_to_f16(const float* in, _Float16* out, int64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(knn_f32_to_f16_kernel, dim3(2048), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}
hipError_t launch_range_sort(const float* rs, const uint32_t* ri, const unsigned* cnt, unsigned cap,
                             const int64_t* lims, int64_t id_base, const int64_t* idmap, int nq, float* D, int64_t* I,
                             hipStream_t st) {
  hipLaunchKernelGGL(knn_range_sort_kernel, dim3(nq), dim3(256), 0, st, rs, ri, cnt, cap, lims, id_base, idmap, D, I,
                     (unsigned)RANGE_SORT_SMALL);
  return hipGetLastError();
}
hipError_t launch_synth(_Float16* X, int64_t row_begin, int64_t n, int d, uint64_t seed, hipStream_t st, int dominant) {
  if (n == 0) return hipSuccess;
  const int64_t chunk = 1 << 22;  // rows per launch (grid.x limit)
  for (int64_t o = 0; o < n; o += chunk) {
    const int64_t m = (n - o) < chunk ? (n - o) : chunk;
    hipLaunchKernelGGL(knn_synth_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, X, row_begin + o, m, d, seed, dominant);
  }
  return hipGetLastError();
}

hipError_t launch_synth_mix(_Float16* X, int64_t row_begin, int64_t row_stride, int64_t n, int d, uint64_t seed, int64_t n_clusters,
                            short* P_table, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (d % 2 != 0 || d > 1024 || n_clusters <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(knn_mix_table_kernel, dim3((MIX_M * d + 255) / 256), dim3(256), 0, st, P_table, d, seed);
  const int64_t chunk = 1 << 22;  // rows per launch (grid.x limit)
  for (int64_t o = 0; o < n; o += chunk) {
    const int64_t m = (n - o) < chunk ? (n - o) : chunk;
    hipLaunchKernelGGL(knn_synth_mix_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, X + (size_t)o * d,
                       row_begin + o * row_stride, row_stride, m, d, seed, (uint64_t)n_clusters, P_table);
  }
  return hipGetLastError();
}
size_t synth_mix_table_bytes(int d) { return (size_t)MIX_M * d * sizeof(short); }

}  // namespace knnx
