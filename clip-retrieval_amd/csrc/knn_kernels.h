// knn_kernels.h -- launchers of the gfx950 kNN kernels (internal; the public ABI is include/knnx.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace knnx {

constexpr int KNN_NQ = 32;     // query columns of one exact scan (MFMA N dimension)
constexpr int KNN_NQ_MAX = 64; // queries of one wide scan (two 32-column blocks, fp16-hi scores + exact re-scoring)
constexpr int KNN_WIDE_KW = 64;   // candidates per query the wide scan keeps
constexpr int KNN_WIDE_MAX_K = 48;  // largest k it serves (needs a gap between the k-th exact and the 64th approx score)
constexpr int KNN_WAVES = 8;   // waves per workgroup, one 32-row tile each
constexpr int KNN_WG = KNN_WAVES * 64;
constexpr float KNN_LO_SCALE = 2048.f;
constexpr float KNN_LO_INV = 1.f / 2048.f;
constexpr int KNN_LDS_BYTES = 160 * 1024;

struct ScanArgs {
  const _Float16* X;
  int64_t N;
  int d;
  const _Float16* qfrag;
  int nq;
  int k;
  int cap;
  int grid;
  int mode;  // 0 top-k, 1 range, 2 dump every score into range_s [nq, range_cap = N]
  int nt;    // 1 = nontemporal X loads (A/B switch, KNNX_NT=1)
  int* thr_g;
  float* part_s;
  uint32_t* part_i;
  int* part_n;
  float range_thr;
  unsigned* range_cnt;
  unsigned range_cap;
  float* range_s;
  uint32_t* range_i;
  // IVF: walk this work list of {tile, query mask, valid rows, -} items instead of all tiles (null = flat scan)
  const uint4* work;
  const unsigned* nwork;
  int wide;              // 1: QB = 2 wide scan (64 queries, approximate scores; flat top-k only)
  const unsigned* gate;  // run only if *gate != 0 (null: always)
  int tstride;           // flat scans: visit every tstride-th 32-row tile only (0 / 1 = all): the sample pass of the RQ scan
  // nblk > 1: blocks of 32 (wide: 64) queries side by side in ONE launch (modes 0 and 2; grid % nblk == 0): qfrag [nblk][d * 64],
  // thr_g [32 nblk] (wide: [64 nblk]), work [nblk][work_stride], nwork [nblk], part_* block-major (knn_kernels.hip: knn_scan_kernel)
  int nblk;
  unsigned work_stride;
};

size_t scan_smem_bytes(int d, int cap, int nq_slots);
hipError_t launch_prep(const float* q_dev, int nq, int d, _Float16* qfrag, int* thr_g, unsigned* range_cnt, int wide,
                       const unsigned* gate, hipStream_t st);
// (wide: blocks of 64 queries, the fp16-hi fragment images of the QB = 2 scan)
hipError_t launch_prep_blocks(const float* q_dev, int nq, int d, _Float16* qfrag, int* thr_a, int* thr_b_or_null, hipStream_t st, int wide = 0);
hipError_t launch_maxnorm(const _Float16* X, int64_t n, int d, int* maxnorm_enc, hipStream_t st);
hipError_t launch_rescore(const _Float16* X, int d, const float* q, const int64_t* cand, const float* approx, int nq, int kw,
                          int k, int64_t id_base, const int* maxnorm_enc, float* D, int64_t* I, unsigned* need, unsigned* gate,
                          unsigned long long* stats, hipStream_t st);
hipError_t launch_select(const unsigned* need, int q0, int nq, int k, const float* Dfb, const int64_t* Ifb, float* D, int64_t* I,
                         hipStream_t st);
hipError_t launch_scan(const ScanArgs& a, hipStream_t st);
hipError_t launch_merge_u32(const float* ps, const uint32_t* pi, const int* pn, int P, int nq_stride, int kin,
                            int nq, int k, int64_t id_base, const int64_t* idmap_or_null, float* D, int64_t* I,
                            const unsigned* gate_or_null, hipStream_t st, int blk_q = 0,  // blk_q = 32: block-major partial lists, or
                            const unsigned* blk_work = nullptr, int nblk = 0, int G = 0);  // (after a multi-block list scan of G
                            // workgroups) block b's lists are the slots of the workgroups that served it: P = upper bound (LDS size)
// IVF-Flat helpers (see knn_kernels.hip)
hipError_t launch_ivf_worklist(const int64_t* Ic, int nq, int nprobe, int nlist, unsigned* masks, const unsigned* tile0,
                               const unsigned* ntile, const unsigned* size, unsigned* off, uint4* work, unsigned* nwork,
                               hipStream_t st, unsigned work_stride = 0);
hipError_t launch_ivf_worklist_from_scores(const float* scores, int nq, int nprobe, int nlist, unsigned* masks, const unsigned* tile0,
                                           const unsigned* ntile, const unsigned* size, unsigned* off, uint4* work, unsigned* nwork,
                                           hipStream_t st, unsigned work_stride = 0);
hipError_t launch_ivf_union_tiles(const unsigned* masks, int nblk, int nlist, const unsigned* ntile, unsigned* out, hipStream_t st);
hipError_t launch_ivf_relayout(const _Float16* src, _Float16* dst, int d, int nlist, const int64_t* src0, const unsigned* tile0,
                               const unsigned* ntile, const unsigned* size, const int64_t* ids, int64_t id_lo, int64_t n_ids,
                               int64_t* idmap, uint32_t* inv, hipStream_t st);
hipError_t launch_gather_inv(const _Float16* X, int d, int64_t id_lo, int64_t n_ids, const uint32_t* inv, const int64_t* ids,
                             int64_t n, float* out, hipStream_t st);
hipError_t launch_merge_i64(const float* ps, const int64_t* pi, int P, int nq, int kin, int k, float* D,
                            int64_t* I, hipStream_t st);
hipError_t launch_merge_sorted(const float* Dp, const int64_t* Ip, int P, int n, int k, float* D, int64_t* I, hipStream_t st);
hipError_t launch_gather(const _Float16* X, int64_t N, int d, int64_t id_base, const int64_t* ids, int64_t n,
                         float* out, hipStream_t st);
hipError_t launch_f32_to_f16(const float* in, _Float16* out, int64_t n, hipStream_t st);
hipError_t launch_range_sort(const float* rs, const uint32_t* ri, const unsigned* cnt, unsigned cap,
                             const int64_t* lims, int64_t id_base, const int64_t* idmap_or_null, int nq, float* D, int64_t* I,
                             hipStream_t st);
// hit lists longer than this are sorted by launch_range_sort_long (radix) instead of launch_range_sort (rank by counting)
constexpr unsigned RANGE_SORT_SMALL = 4096;
hipError_t launch_range_sort_long(const float* vs, const uint32_t* ri, unsigned n, int64_t id_base, const int64_t* idmap_or_null,
                                  int key_bits, uint32_t* k0, uint32_t* k1, float* v0, float* v1, unsigned* hist, float* D, int64_t* I,
                                  hipStream_t st);
hipError_t launch_synth(_Float16* X, int64_t row_begin, int64_t n, int d, uint64_t seed, hipStream_t st, int dominant = 0);
// the mixture corpus of BASELINE config 5 (knn_kernels.hip: knn_synth_mix_kernel): destination row i = corpus row
// row_begin + i * row_stride; P_table = synth_mix_table_bytes(d) bytes of device scratch
hipError_t launch_synth_mix(_Float16* X, int64_t row_begin, int64_t row_stride, int64_t n, int d, uint64_t seed, int64_t n_clusters,
                            short* P_table, hipStream_t st);
size_t synth_mix_table_bytes(int d);
hipError_t launch_copy_rows(const _Float16* src, int d, const int64_t* src_rows, const int32_t* dst_rows, int64_t n, _Float16* dst,
                            hipStream_t st);
hipError_t launch_ivf_hist(const int32_t* lists, int64_t n, int nlist, unsigned long long* hist, hipStream_t st);

// ---- register-stationary-queries (RQ) scan, knn_rq_kernels.hip: up to rq_queries_per_pass(d) queries per pass over HBM
constexpr int KNN_RQ_MAX = 256;        // queries of one RQ pass at d <= 768 (128 at d = 1024)
// The sample pass visits every S-th 32-row tile, S = min(1024, tiles / 4096) (>= 4096 tiles = 131 k rows sampled); the
// threshold is the J-th best sample score with J = k + 8 up to S = 128 and J = (k + 8) * 128 / S (>= 6) beyond, i.e. ~6 k
// expected hits per query whatever the index size.  The number of index rows above the J-th best of a 1/S sample is
// distribution-free (~ S * Gamma(J)): at J = 9, S = 763 (100 M rows) P(hits > 32768) < 1e-15 and P(hits < k) = 0.
// (A 1/128 sample of 100 M rows cost 0.9 ms per 64-query pass -- cold LDS queues are pruned every round -- 3.6 ms of a
// 45 ms search; the 1/763 sample 0.3 ms.)
constexpr int KNN_RQ_STRIDE = 1024;
constexpr int KNN_RQ_MARGIN = 8;
constexpr unsigned KNN_RQ_CAP = 32768; // hit list entries per query
constexpr int64_t KNN_RQ_MIN_ROWS = (int64_t)1 << 21;  // below this the 64-query scan is used
int rq_queries_per_pass(int d);  // 0: no RQ kernel for this d
hipError_t launch_rq_prep(const float* q_dev, int nq, int d, _Float16* qfrag, const float* samp, int kw, int J, float slack,
                          float* thr, unsigned* cnt, unsigned* lost, hipStream_t st);
hipError_t launch_rq_scan(const _Float16* X, int64_t N, int d, int nq, const _Float16* qfrag, const float* thr, unsigned* cnt,
                          unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, const unsigned* gate, int grid,
                          hipStream_t st, uint32_t row_off = 0);  // row_off: X is a slice starting at that row of the index (added to the hits' rows)
// IVF build (knn_rq_kernels.hip / knn_kernels.hip): out[i] = argmax_l <P[i], C[l]> (fp16 rows, exact fp32 scores, ties -> smaller l)
hipError_t launch_assign(const _Float16* C, int64_t nlist, int d, const _Float16* P, int64_t n, int32_t* out, hipStream_t st);
// one Lloyd update (spherical): cent[l] = fp16(unit-norm mean of X[order[off[l] .. off[l+1])]) (lists with no member keep their row); one workgroup per list
hipError_t launch_kmeans_update(const _Float16* X, int d, const int64_t* order, const int64_t* off, int nlist, _Float16* cent,
                                hipStream_t st);
// scatter n assigned rows into the tile-padded list-sorted arena: dst row = tile0[list] * 32 + pos; lays down idmap / inv
// (ids == null: row i carries id id0 + i)
hipError_t launch_ivf_scatter(const _Float16* src, int64_t n, int d, const int32_t* lists, const int32_t* pos, const int64_t* ids,
                              int64_t id0, const unsigned* tile0, int64_t id_lo, int64_t n_ids, _Float16* dst, int64_t* idmap,
                              uint32_t* inv, hipStream_t st);
hipError_t launch_rq_rescore(const _Float16* X, int d, const float* q, int nq, const unsigned* cnt, unsigned cap, float* hit_s,
                             const uint32_t* hit_r, int* cntc, hipStream_t st);
hipError_t launch_rq_proof(const float* q, int nq, int d, int k, const float* D, const float* thr, const unsigned* cnt,
                           unsigned cap, const unsigned* lost, const int* maxnorm, unsigned* need, unsigned* gate,
                           unsigned long long* stats, hipStream_t st);

// ---- int8 first stage of the flat scans (knn_rq_kernels.hip, "int8 first stage"): exact results from half the bytes
constexpr int KNN_I8_STRIDE = 32;          // the sample pass visits every S-th tile, S = min(this, tiles / 4096): threshold ~ rank (k + 8) S
constexpr unsigned KNN_I8_CAP = KNN_RQ_CAP;  // hit list entries per query (knn_merge_kernel holds a query's whole list in LDS: 32 768 x 4 B)
int i8_supported(int d);
// Dominant columns of an index: up to four columns whose scale is far above the others' sit in BYTES 0..3 of every row of
// the int8 image (a column permutation private to the image: position p of the image holds column p of the row, except the nfix
// positions listed here) and are left out of the int8 MFMA product; the scan adds their part with 14-bit query digits on the
// vector ALU (knn_rq8_scan_kernel, DOM).  n = 0: no dominant columns, the image is in column order.
struct I8Dom {
  int n = 0;     // dominant columns = positions 0 .. n - 1
  int nfix = 0;  // positions whose column is not the position
  int pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int src[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
hipError_t launch_i8_scales(const _Float16* X, int64_t N, int d, int* colmax_enc, float* colscale, int* ab_enc, hipStream_t st);
hipError_t launch_i8_quant(const _Float16* X, int64_t N, int64_t row_from, int64_t row_to, int d, const float* colscale, const I8Dom& dom,
                           int8_t* X8, int* ab_enc, hipStream_t st);
hipError_t launch_i8_prep(const float* q_dev, int nq, int d, const float* colscale, const I8Dom& dom, const int* ab_enc, const int* maxnorm,
                          const float* samp, int kw, int J, int planes, int refine, int8_t* qfrag8, int8_t* qdom, int* thr_i, float* thr_lb,
                          float* thr_rest, unsigned* cnt, unsigned* lost, hipStream_t st);
hipError_t launch_rq8_scan(const int8_t* X8, int64_t N, int d, int nq, int planes, const int8_t* qfrag8, const int8_t* qdom, const int* thr_i,
                           unsigned* cnt, unsigned cap, float* hit_s, uint32_t* hit_r, unsigned* lost, int grid, int tstep, hipStream_t st);
hipError_t launch_i8_proof(int nq, int k, const float* D, const float* thr_lb, const unsigned* cnt, unsigned cap, const unsigned* lost,
                           unsigned* need, unsigned* gate, unsigned long long* stats, hipStream_t st);

}  // namespace knnx
