// knnx_api.hip -- host side of the C ABI declared in include/knnx.h (search half of the hot path).
//
// Replaces the faiss Index object the reference holds in ClipResource.image_index
// (clip_retrieval/clip_back.py:781-782) and calls at clip_back.py:362 / clip_filter.py:52,55.
// Owns: the HBM arena of fp16 rows, one HIP stream, per-scan scratch.  No CPU compute path
// exists here on purpose: if HIP is unavailable every entry point fails with KNNX_E_HIP.

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/knnx.h"
#include "knn_kernels.h"

using namespace knnx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fail(_e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP,                         \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                             \
  } while (0)

struct knnx_index {
  int device = 0;
  int d = 0;
  int n_cu = 256;
  int64_t ntotal = 0;
  int64_t capacity = 0;
  int64_t id_base = 0;
  _Float16* rows = nullptr;  // [capacity, d]
  bool borrowed = false;
  hipStream_t stream = nullptr;
  std::mutex mu;

  // Request coalescer ("knnx_search is re-entrant; internally a batching queue"): concurrent single-query callers
  // (the werkzeug request threads of clip_back.py:1018) queue here; one of them -- the leader -- serves everything that queued
  // while the GPU was busy, up to one scan's worth with the same k, in ONE pass over HBM, then hands the lead on.
  bool coalesce = true;
  std::mutex co_mu;
  std::condition_variable co_cv;
  std::deque<struct CoReq*> co_q;
  bool co_leader = false;
  int64_t co_batches = 0, co_queries = 0, co_largest = 0;
  std::vector<float> co_qbuf, co_Dbuf;
  std::vector<int64_t> co_Ibuf;

  // per-scan scratch (sized for KNN_NQ_MAX queries, grid = n_cu workgroups, k <= KNNX_MAX_K_FAST)
  _Float16* qfrag = nullptr;
  float* q_dev = nullptr;  // [KNN_NQ, d]
  int* thr_g = nullptr;
  float* part_s = nullptr;
  uint32_t* part_i = nullptr;
  int* part_n = nullptr;
  float* D_dev = nullptr;    // [KNN_NQ, KNNX_MAX_K_FAST]
  int64_t* I_dev = nullptr;  // [KNN_NQ, KNNX_MAX_K_FAST]
  unsigned* range_cnt = nullptr;
  float* range_s = nullptr;
  uint32_t* range_i = nullptr;
  size_t range_pool = 0;  // entries in range_s / range_i, shared by the queries of one scan

  // pinned staging for host<->device hand-over
  void* pin = nullptr;
  size_t pin_bytes = 0;
  // device scratch of reconstruct / range_fetch, kept between calls (hipMalloc + hipFree per request cost more than a
  // small search: hipFree synchronises the device); grown on demand, released with the index.  Used under ix->mu only.
  void* scratch[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

  // IVF-Flat state (knnx_ivf_set_lists): rows live list-sorted and tile-padded in `rows`; see knn_kernels.hip
  int ivf_nlist = 0, ivf_nprobe = 1;
  knnx_index* cent = nullptr;  // coarse quantiser: a flat index over the fp16 centroids
  unsigned *ivf_tile0 = nullptr, *ivf_ntile = nullptr, *ivf_size = nullptr, *ivf_masks = nullptr, *ivf_off = nullptr,
           *ivf_nwork = nullptr;
  uint4* ivf_work = nullptr;
  int64_t* ivf_idmap = nullptr;  // padded arena row -> id (-1 on pad rows)
  uint32_t* ivf_inv = nullptr;   // id - id_base -> padded arena row
  int64_t* ivf_Ic = nullptr;     // [KNN_NQ, KNNX_MAX_K_FAST] coarse result
  float* ivf_Dc = nullptr;
  float* ivf_scores = nullptr;   // [KNN_NQ, nlist] coarse scores (nprobe > 64 only; allocated on first use)
  // Unproven queries of the proof-based scans (wide / RQ / int8) get the exact 32-query scan of their group.  On the device-buffer
  // entry point that scan is launched for every group, gated on the device (no host round trip); the host-buffer entry points
  // (search_fast_locked) read the gates back with the results instead and launch it only for a group that needs it -- 4 gated
  // no-op kernels per group, 32 per 256-query batch, ~0.16 ms of launch slots (profiles/r06r_knn_b256_timeline.log).
  int sample_one_launch = 1;  // the sample scans of a register-stationary batch as blocks of one launch (KNNX_SAMPLE_STREAMS=1: a stream each)
  _Float16* rq_m_qfrag = nullptr;
  int* rq_m_thr = nullptr;
  bool defer_fb = false;
  int defer_kind = 0, defer_nq = 0;  // set by the scan that skipped its fallback: 1 = rq_need / rq_gate, 2 = wide_need / wide_gate
  // multi-block pass (scan_topk_ivf_multi): up to IVFM_BLK blocks of 32 queries in ONE coarse scan + ONE list scan; allocated on
  // first use, ivfm_ok = 0 (KNNX_IVF_MULTI=0, or an allocation failed) keeps the 32-queries-per-pass path
  int ivfm_ok = 1;
  int ivfm_last_blk = 0;           // blocks of the most recent IVF scan (0: the single-block path) -- knnx_ivf_last_scan_tiles
  bool ivfm_union_valid = false;   // the most recent pass also counted the union of its lists (profiling was on)
  _Float16* ivfm_qfrag = nullptr;  // [IVFM_BLK][d * 64] fragment images
  int *ivfm_thr_c = nullptr, *ivfm_thr_f = nullptr;  // [32 IVFM_BLK] thresholds of the coarse / the list scan
  unsigned *ivfm_masks = nullptr, *ivfm_off = nullptr, *ivfm_nwork = nullptr;  // nwork: 16 counters (8: union tiles) + masks [IVFM_BLK][nlist] behind them (one allocation); off: unused since the atomic work list
  uint4* ivfm_work = nullptr;      // [IVFM_BLK][ivfm_stride]
  unsigned ivfm_stride = 0;
  float* ivfm_scores = nullptr;    // [32 IVFM_BLK, nlist] coarse scores of one pass
  // streaming build (knnx_ivf_begin .. knnx_ivf_end)
  int ivfb_nlist = 0;
  int64_t ivfb_total = 0, ivfb_added = 0;
  std::vector<uint16_t> ivfb_cent;
  void *ivfb_rows = nullptr, *ivfb_ids = nullptr, *ivfb_lists = nullptr, *ivfb_pos = nullptr;  // device staging of one chunk
  // host mirror of the layout, so that every (list, position) a caller hands over is checked before it is scattered
  // (a position past its list, or used twice, would silently overwrite a neighbouring list's rows)
  std::vector<uint32_t> ivfb_size, ivfb_fill, ivfb_tile0;
  std::vector<uint64_t> ivfb_taken;  // one bit per padded arena row

  // int8 first stage of the flat scans (knn_rq_kernels.hip): allocated and built on first use, rebuilt after the rows change
  int i8_ok = 1;                  // KNNX_I8=0 disables; cleared for good when its memory cannot be had
  bool i8_valid = false;          // rows8 / colscale / ab describe the current rows
  int64_t i8_cap_rows = 0;
  int64_t i8_nrows = 0;           // rows [0, i8_nrows) are quantised with the current column scales (add() appends: only the new rows are done)
  int64_t i8_scale_rows = 0;      // rows the column scales were taken over (a full rebuild once the index has doubled since)
  int i8_planes = 1;              // int8 planes of a query: 2 when the column scales differ widely (decided at every full build)
  int i8_planes_env = 0;          // KNNX_I8_PLANES=1|2 forces the choice (and switches the dominant-column form off)
  I8Dom i8_dom;                   // dominant columns of the index (knn_kernels.h), decided with the planes
  int i8_dom_off = 0;             // KNNX_I8_DOM=0: never use the dominant-column form (two planes instead)
  int8_t* i8_qdom = nullptr;      // [2][256][4] the queries' 14-bit digits at the dominant positions (knn_i8_prep_kernel)
  int8_t* i8_rows = nullptr;      // tile-ordered image of rows [0, i8_nrows) (knn_i8_quant_kernel): i8_cap_rows / 32 tiles of 32 d bytes
  int64_t i8_budget = 0;          // KNNX_I8_MAX_BYTES: cap on the image (0: none) -- a PARTIAL copy: the other rows are scanned in fp16
  unsigned long long i8_rest_served = 0;  // queries whose pass ran over an int8 part AND an fp16 rest
  float* i8_colscale = nullptr;   // [d]
  int *i8_colmax = nullptr, *i8_ab = nullptr;  // [d] (encoded), [2] (encoded A, B)
  int8_t* i8_qfrag = nullptr;     // [16 blocks][d / 64][64][16]
  int* i8_thr = nullptr;          // [256]
  float* i8_lb = nullptr;         // [256]
  float* i8_hit_s = nullptr;      // [256, KNN_I8_CAP]
  uint32_t* i8_hit_r = nullptr;
  unsigned long long i8_served = 0;  // queries answered through the int8 path (knnx_i8_active reports it)

  // wide scan (64 queries per pass): candidates, fallback results, proof flags; largest row norm (order-encoded)
  int wide_ok = 1;             // KNNX_WIDE=0 disables
  int* maxnorm = nullptr;
  int64_t* wide_cand = nullptr;  // [64, 64]
  float* wide_approx = nullptr;  // [64, 64]
  unsigned* wide_need = nullptr; // [64]
  unsigned* wide_gate = nullptr; // [2]
  float* wide_Dfb = nullptr;     // [32, 64]
  int64_t* wide_Ifb = nullptr;

  // RQ scan (register-stationary queries, up to 256 per pass; knn_rq_kernels.hip): allocated on first use
  int rq_ok = 1;               // KNNX_RQ=0 disables
  int64_t rq_min_rows = KNN_RQ_MIN_ROWS;  // KNNX_RQ_MIN_ROWS overrides (tests run the RQ path on small indexes)
  int rq_sample_grid = 0;      // workgroups of the threshold-sample scans (0: chosen from the sample size; KNNX_RQ_SAMPLE_GRID overrides)
  _Float16* rq_qfrag = nullptr;  // [8 blocks][d/16][64][8]
  float* rq_thr = nullptr;       // [256]
  unsigned *rq_cnt = nullptr, *rq_lost = nullptr, *rq_need = nullptr, *rq_gate = nullptr;  // [256] x3, [8]
  int* rq_cntc = nullptr;        // [256]
  float* rq_hit_s = nullptr;     // [256, KNN_RQ_CAP]
  uint32_t* rq_hit_r = nullptr;
  float* rq_samp = nullptr;      // [256, 64] sample scores
  int64_t* rq_samp_i = nullptr;
  // the threshold-sample scans of a batch (up to 4 groups of 64 / 32 queries) run CONCURRENTLY: groups 1 .. 3 on side streams
  // with their own copy of the scan scratch (fragments, global thresholds, per-workgroup part lists)
  static constexpr int RQ_SIDE = 3;
  hipStream_t rq_side[RQ_SIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t rq_fork = nullptr, rq_join[RQ_SIDE] = {nullptr, nullptr, nullptr};
  _Float16* rq_s_qfrag[RQ_SIDE] = {nullptr, nullptr, nullptr};
  int* rq_s_thr_g[RQ_SIDE] = {nullptr, nullptr, nullptr};
  float* rq_s_part_s[RQ_SIDE] = {nullptr, nullptr, nullptr};
  uint32_t* rq_s_part_i[RQ_SIDE] = {nullptr, nullptr, nullptr};
  int* rq_s_part_n[RQ_SIDE] = {nullptr, nullptr, nullptr};
  unsigned long long* stats = nullptr;  // device counters: [0] queries served by a proof-based path, [1] proofs that failed

  // scratch hand-over between streams: the per-handle scratch above is shared by every call, so each launch sequence
  // first waits for the event the previous sequence recorded, whichever stream that ran on
  hipEvent_t ev_scratch = nullptr;
  bool ev_valid = false;

  int nt_loads = 0;
  bool prof = false;
  int64_t prof_launches = 0;
  double prof_ms = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
};

static int set_dev(const knnx_index* ix) {
  HIPCHK(hipSetDevice(ix->device));
  return 0;
}

static int ensure_pin(knnx_index* ix, size_t bytes) {
  if (ix->pin_bytes >= bytes) return 0;
  if (ix->pin) hipHostFree(ix->pin);
  ix->pin = nullptr;
  ix->pin_bytes = 0;
  HIPCHK(hipHostMalloc(&ix->pin, bytes, hipHostMallocDefault));
  ix->pin_bytes = bytes;
  return 0;
}

// the int8 copy is an accelerator, not data: whoever needs device memory and cannot get it takes the copy's back and tries once more
// (the lazily built copy must not starve the range pools / large-k scratch that fitted before it existed)
static void i8_release(knnx_index* ix) {
  if (!ix->i8_rows) return;
  hipFree(ix->i8_rows);
  ix->i8_rows = nullptr;
  ix->i8_cap_rows = 0;
  ix->i8_nrows = 0;
  ix->i8_valid = false;
}
static hipError_t malloc_or_reclaim(knnx_index* ix, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess && ix->i8_rows) {
    (void)hipGetLastError();
    i8_release(ix);
    ix->i8_ok = 0;  // it did not fit next to what the index needs: do not rebuild it on the next search
    e = hipMalloc(p, bytes);
  }
  return e;
}

static int ensure_scratch(knnx_index* ix, int slot, size_t bytes, void** out) {
  if (ix->scratch_bytes[slot] < bytes) {
    if (ix->scratch[slot]) hipFree(ix->scratch[slot]);
    ix->scratch[slot] = nullptr;
    ix->scratch_bytes[slot] = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    HIPCHK(malloc_or_reclaim(ix, &ix->scratch[slot], want));
    ix->scratch_bytes[slot] = want;
  }
  *out = ix->scratch[slot];
  return 0;
}

static int scan_cap(int d, int k) {
  long avail = (long)KNN_LDS_BYTES - (long)d * 128 - KNN_NQ * 8 - 16;
  int cap = (int)(avail / (KNN_NQ * 8));
  cap = std::min(cap, 128);
  cap &= ~1;
  if (cap < k + 16) return -1;
  return cap;
}

extern "C" const char* knnx_last_error(void) { return g_err.c_str(); }
// used by knnx_sharded.hip (same library, other translation unit): set the thread-local message
extern "C" int knnx_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }

extern "C" int knnx_create(int device, int d, int metric, knnx_index** out) {
  if (!out) return fail(KNNX_E_ARG, "out is null");
  *out = nullptr;
  if (metric != KNNX_METRIC_INNER_PRODUCT) return fail(KNNX_E_UNSUPPORTED, "only inner product is implemented");
  if (d <= 0 || d % 256 != 0 || d > 1024)
    return fail(KNNX_E_UNSUPPORTED, "d must be a multiple of 256 and <= 1024 (pad rows with zeros otherwise)");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(KNNX_E_ARG, "no such HIP device");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(KNNX_E_UNSUPPORTED, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);

  knnx_index* ix = new knnx_index();
  ix->device = device;
  ix->d = d;
  ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const char* nt = getenv("KNNX_NT");
  ix->nt_loads = (nt && nt[0] == '1') ? 1 : 0;
  const char* wd = getenv("KNNX_WIDE");
  ix->wide_ok = (wd && wd[0] == '0') ? 0 : 1;
  const char* rq = getenv("KNNX_RQ");
  ix->rq_ok = (rq && rq[0] == '0') ? 0 : 1;
  {
    const char* i8 = getenv("KNNX_I8");
    ix->i8_ok = (i8 && i8[0] == '0') ? 0 : 1;
    const char* pl = getenv("KNNX_I8_PLANES");
    ix->i8_planes_env = (pl && (pl[0] == '1' || pl[0] == '2')) ? pl[0] - '0' : 0;
    const char* dm = getenv("KNNX_I8_DOM");
    ix->i8_dom_off = (dm && dm[0] == '0') ? 1 : 0;
    const char* i8b = getenv("KNNX_I8_MAX_BYTES");
    ix->i8_budget = i8b ? std::max<int64_t>(0, atoll(i8b)) : 0;
  }
  const char* ss = getenv("KNNX_SAMPLE_STREAMS");
  ix->sample_one_launch = (ss && ss[0] == '1') ? 0 : 1;
  const char* im = getenv("KNNX_IVF_MULTI");
  ix->ivfm_ok = (im && im[0] == '0') ? 0 : 1;
  const char* rqm = getenv("KNNX_RQ_MIN_ROWS");
  if (rqm && rqm[0]) ix->rq_min_rows = atoll(rqm);
  const char* sg = getenv("KNNX_RQ_SAMPLE_GRID");
  if (sg && atoi(sg) > 0) ix->rq_sample_grid = atoi(sg);
  const char* gr = getenv("KNNX_GRID");
  if (gr && atoi(gr) > 0) ix->n_cu = atoi(gr);
  hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
  const size_t G = (size_t)ix->n_cu;
  if (e == hipSuccess) e = hipMalloc(&ix->qfrag, (size_t)d * 128);
  if (e == hipSuccess) e = hipMalloc(&ix->q_dev, (size_t)KNN_RQ_MAX * d * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ix->thr_g, KNN_NQ_MAX * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&ix->part_s, G * KNN_NQ_MAX * KNNX_MAX_K_FAST * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ix->part_i, G * KNN_NQ_MAX * KNNX_MAX_K_FAST * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&ix->part_n, G * KNN_NQ_MAX * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&ix->D_dev, (size_t)KNN_RQ_MAX * KNNX_MAX_K_FAST * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ix->I_dev, (size_t)KNN_RQ_MAX * KNNX_MAX_K_FAST * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->range_cnt, KNN_NQ * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->maxnorm, sizeof(int));
  if (e == hipSuccess) e = hipMemset(ix->maxnorm, 0, sizeof(int));  // order-encoded +0.0f
  if (e == hipSuccess) e = hipMalloc(&ix->wide_cand, (size_t)KNN_NQ_MAX * KNN_WIDE_KW * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->wide_approx, (size_t)KNN_NQ_MAX * KNN_WIDE_KW * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ix->wide_need, KNN_NQ_MAX * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->wide_gate, 2 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->wide_Dfb, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ix->wide_Ifb, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->stats, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(ix->stats, 0, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_scratch, hipEventDisableTiming);
  if (e != hipSuccess) {
    std::string m = std::string("knnx_create: ") + hipGetErrorString(e);
    knnx_destroy(ix);
    return fail(KNNX_E_HIP, m);
  }
  *out = ix;
  return KNNX_OK;
}

extern "C" void knnx_destroy(knnx_index* ix) {
  if (!ix) return;
  hipSetDevice(ix->device);
  if (ix->stream) hipStreamSynchronize(ix->stream);
  if (ix->rows && !ix->borrowed) hipFree(ix->rows);
  hipFree(ix->qfrag);
  hipFree(ix->q_dev);
  hipFree(ix->thr_g);
  hipFree(ix->part_s);
  hipFree(ix->part_i);
  hipFree(ix->part_n);
  hipFree(ix->D_dev);
  hipFree(ix->I_dev);
  hipFree(ix->range_cnt);
  hipFree(ix->maxnorm);
  hipFree(ix->wide_cand);
  hipFree(ix->wide_approx);
  hipFree(ix->wide_need);
  hipFree(ix->wide_gate);
  hipFree(ix->wide_Dfb);
  hipFree(ix->wide_Ifb);
  hipFree(ix->stats);
  hipFree(ix->i8_rows);
  hipFree(ix->i8_colscale);
  hipFree(ix->i8_qdom);
  hipFree(ix->i8_colmax);
  hipFree(ix->i8_ab);
  hipFree(ix->i8_qfrag);
  hipFree(ix->i8_thr);
  hipFree(ix->i8_lb);
  hipFree(ix->i8_hit_s);
  hipFree(ix->i8_hit_r);
  hipFree(ix->rq_qfrag);
  hipFree(ix->rq_thr);
  hipFree(ix->rq_cnt);
  hipFree(ix->rq_lost);
  hipFree(ix->rq_need);
  hipFree(ix->rq_gate);
  hipFree(ix->rq_cntc);
  hipFree(ix->rq_hit_s);
  hipFree(ix->rq_hit_r);
  hipFree(ix->rq_samp);
  hipFree(ix->rq_samp_i);
  hipFree(ix->rq_m_qfrag);
  hipFree(ix->rq_m_thr);
  for (int i = 0; i < knnx_index::RQ_SIDE; ++i) {
    hipFree(ix->rq_s_qfrag[i]);
    hipFree(ix->rq_s_thr_g[i]);
    hipFree(ix->rq_s_part_s[i]);
    hipFree(ix->rq_s_part_i[i]);
    hipFree(ix->rq_s_part_n[i]);
    if (ix->rq_side[i]) (void)hipStreamDestroy(ix->rq_side[i]);
    if (ix->rq_join[i]) (void)hipEventDestroy(ix->rq_join[i]);
  }
  if (ix->rq_fork) (void)hipEventDestroy(ix->rq_fork);
  if (ix->ev_scratch) hipEventDestroy(ix->ev_scratch);
  if (ix->range_s) hipFree(ix->range_s);
  if (ix->range_i) hipFree(ix->range_i);
  if (ix->cent) knnx_destroy(ix->cent);
  hipSetDevice(ix->device);
  hipFree(ix->ivf_tile0);
  hipFree(ix->ivf_ntile);
  hipFree(ix->ivf_size);
  hipFree(ix->ivf_masks);
  hipFree(ix->ivf_off);
  hipFree(ix->ivf_nwork);
  hipFree(ix->ivf_work);
  hipFree(ix->ivf_idmap);
  hipFree(ix->ivf_inv);
  hipFree(ix->ivf_Ic);
  hipFree(ix->ivf_Dc);
  hipFree(ix->ivf_scores);
  hipFree(ix->ivfm_qfrag);
  hipFree(ix->ivfm_thr_c);
  hipFree(ix->ivfm_thr_f);
  hipFree(ix->ivfm_off);
  hipFree(ix->ivfm_nwork);
  hipFree(ix->ivfm_work);
  hipFree(ix->ivfm_scores);
  hipFree(ix->ivfb_rows);
  hipFree(ix->ivfb_ids);
  hipFree(ix->ivfb_lists);
  hipFree(ix->ivfb_pos);
  if (ix->pin) hipHostFree(ix->pin);
  for (int i = 0; i < 9; ++i)
    if (ix->scratch[i]) hipFree(ix->scratch[i]);
  for (auto& ev : ix->prof_events) {
    hipEventDestroy(ev.first);
    hipEventDestroy(ev.second);
  }
  if (ix->stream) hipStreamDestroy(ix->stream);
  delete ix;
}

extern "C" int64_t knnx_ntotal(const knnx_index* ix) { return ix ? ix->ntotal : 0; }
extern "C" int knnx_dim(const knnx_index* ix) { return ix ? ix->d : 0; }

// faiss Index.reset(): forget the rows, keep the arena (the per-request dedup index of clip_back.py:290-294 is rebuilt from
// the <= k result vectors of every query; allocating a fresh index per request would cost more than the search)
extern "C" int knnx_reset(knnx_index* ix) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->ivf_nlist || ix->ivfb_nlist) return fail(KNNX_E_STATE, "reset of an IVF index is not supported (destroy it instead)");
  if (set_dev(ix)) return KNNX_E_HIP;
  HIPCHK(hipStreamSynchronize(ix->stream));
  ix->ntotal = 0;
  ix->i8_valid = false;
  ix->i8_nrows = 0;
  {  // a copy that was given up for lack of memory may be tried again for the new rows; KNNX_I8=0 still rules
    const char* i8 = getenv("KNNX_I8");
    ix->i8_ok = (i8 && i8[0] == '0') ? 0 : 1;
  }
  HIPCHK(hipMemsetAsync(ix->maxnorm, 0, sizeof(int), ix->stream));
  return KNNX_OK;
}

extern "C" int knnx_set_id_base(knnx_index* ix, int64_t id_base) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  ix->id_base = id_base;
  return KNNX_OK;
}

static int grow(knnx_index* ix, int64_t need_rows) {
  if (need_rows <= ix->capacity) return 0;
  if (ix->borrowed) return fail(KNNX_E_STATE, "index borrows caller memory; cannot grow");
  if (need_rows > (int64_t)0xffffffffll) return fail(KNNX_E_UNSUPPORTED, "more than 2^32 rows per device");
  _Float16* nr = nullptr;
  if (hipMalloc(&nr, (size_t)need_rows * ix->d * sizeof(_Float16)) != hipSuccess) {
    (void)hipGetLastError();
    // the int8 copy is an accelerator, not data: give its memory back and try once more
    i8_release(ix);
    HIPCHK(hipMalloc(&nr, (size_t)need_rows * ix->d * sizeof(_Float16)));
  }
  if (ix->rows && ix->ntotal > 0) {
    hipError_t e = hipMemcpyAsync(nr, ix->rows, (size_t)ix->ntotal * ix->d * sizeof(_Float16),
                                  hipMemcpyDeviceToDevice, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
      hipFree(nr);
      return fail(KNNX_E_HIP, std::string("grow copy: ") + hipGetErrorString(e));
    }
  }
  if (ix->rows) hipFree(ix->rows);
  ix->rows = nr;
  ix->capacity = need_rows;
  return 0;
}

extern "C" int knnx_reserve(knnx_index* ix, int64_t n_rows) {
  if (!ix || n_rows < 0) return fail(KNNX_E_ARG, "bad reserve");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  return grow(ix, n_rows);
}

static int add_common(knnx_index* ix, const void* rows, int64_t n, bool is_f32) {
  if (!ix || (!rows && n > 0) || n < 0) return fail(KNNX_E_ARG, "bad add arguments");
  if (n == 0) return KNNX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (ix->borrowed) return fail(KNNX_E_STATE, "add() on an index that borrows device rows");
  if (ix->ivf_nlist) return fail(KNNX_E_STATE, "add() after knnx_ivf_set_lists (rebuild the index instead)");
  if (ix->ntotal + n > ix->capacity) {
    int64_t want = std::max<int64_t>(ix->ntotal + n, ix->capacity + ix->capacity / 2);
    int r = grow(ix, want);
    if (r) return r;
  }
  const size_t esz = is_f32 ? 4 : 2;
  const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)(ix->d * esz));
  int r = ensure_pin(ix, (size_t)chunk_rows * ix->d * esz);
  if (r) return r;
  float* tmp32 = nullptr;
  if (is_f32) HIPCHK(hipMalloc(&tmp32, (size_t)chunk_rows * ix->d * 4));
  for (int64_t o = 0; o < n; o += chunk_rows) {
    const int64_t m = std::min(chunk_rows, n - o);
    const size_t bytes = (size_t)m * ix->d * esz;
    memcpy(ix->pin, (const char*)rows + (size_t)o * ix->d * esz, bytes);
    _Float16* dst = ix->rows + (size_t)(ix->ntotal + o) * ix->d;
    hipError_t e;
    if (is_f32) {
      e = hipMemcpyAsync(tmp32, ix->pin, bytes, hipMemcpyHostToDevice, ix->stream);
      if (e == hipSuccess) e = launch_f32_to_f16(tmp32, dst, m * ix->d, ix->stream);
    } else {
      e = hipMemcpyAsync(dst, ix->pin, bytes, hipMemcpyHostToDevice, ix->stream);
    }
    if (e == hipSuccess) e = launch_maxnorm(dst, m, ix->d, ix->maxnorm, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
      if (tmp32) hipFree(tmp32);
      return fail(KNNX_E_HIP, std::string("add copy: ") + hipGetErrorString(e));
    }
  }
  if (tmp32) hipFree(tmp32);
  ix->ntotal += n;
  ix->i8_valid = false;  // (the rows quantised so far stay: i8_ensure quantises [i8_nrows, ntotal) only)
  return KNNX_OK;
}

extern "C" int knnx_add_f16(knnx_index* ix, const uint16_t* rows, int64_t n) { return add_common(ix, rows, n, false); }
extern "C" int knnx_add_f32(knnx_index* ix, const float* rows, int64_t n) { return add_common(ix, rows, n, true); }

extern "C" int knnx_attach_device_f16(knnx_index* ix, const void* dev_rows, int64_t n) {
  if (!ix || !dev_rows || n < 0) return fail(KNNX_E_ARG, "bad attach arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->rows && !ix->borrowed) return fail(KNNX_E_STATE, "index already owns rows");
  if (n > (int64_t)0xffffffffll) return fail(KNNX_E_UNSUPPORTED, "more than 2^32 rows per device");
  ix->rows = (_Float16*)dev_rows;
  ix->borrowed = true;
  ix->capacity = n;
  ix->ntotal = n;
  ix->i8_valid = false;
  ix->i8_nrows = 0;
  if (set_dev(ix)) return KNNX_E_HIP;
  HIPCHK(hipMemsetAsync(ix->maxnorm, 0, sizeof(int), ix->stream));
  HIPCHK(launch_maxnorm(ix->rows, n, ix->d, ix->maxnorm, ix->stream));
  HIPCHK(hipStreamSynchronize(ix->stream));
  return KNNX_OK;
}

extern "C" int knnx_synth_fill(knnx_index* ix, int64_t n, uint64_t seed) {
  if (!ix || n < 0) return fail(KNNX_E_ARG, "bad synth arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (!ix->borrowed) {
    int r = grow(ix, n);
    if (r) return r;
  } else if (n > ix->capacity) {
    return fail(KNNX_E_NOMEM, "attached arena is smaller than n");
  }
  HIPCHK(launch_synth(ix->rows, 0, n, ix->d, seed, ix->stream));
  HIPCHK(hipMemsetAsync(ix->maxnorm, 0, sizeof(int), ix->stream));
  HIPCHK(launch_maxnorm(ix->rows, n, ix->d, ix->maxnorm, ix->stream));
  HIPCHK(hipStreamSynchronize(ix->stream));
  ix->ntotal = n;
  ix->i8_valid = false;
  ix->i8_nrows = 0;
  return KNNX_OK;
}

static int scan_topk(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out,
                     hipStream_t st, const unsigned* gate = nullptr);

// IVF: the work list (ix->ivf_work / ivf_nwork) of <= KNN_NQ queries already in HBM
static int ivf_build_worklist(knnx_index* ix, const float* q_dev, int nq, hipStream_t st, const unsigned* gate) {
  {
    // coarse quantiser: top-nprobe centroids per query (the same flat scan over the [nlist, d] centroid rows), then
    // the work list = tiles of every list probed by at least one of the <= 32 queries, each with its query mask
    const int np = std::min(ix->ivf_nprobe, ix->ivf_nlist);
    if (np <= KNNX_MAX_K_FAST) {
      int r = scan_topk(ix->cent, q_dev, nq, np, ix->ivf_Dc, ix->ivf_Ic, st);
      if (r) return r;
      HIPCHK(launch_ivf_worklist(ix->ivf_Ic, nq, np, ix->ivf_nlist, ix->ivf_masks, ix->ivf_tile0, ix->ivf_ntile, ix->ivf_size,
                                 ix->ivf_off, ix->ivf_work, ix->ivf_nwork, st));
    } else {
      // nprobe > 64 (BASELINE config 5 asks for 256): the centroid scan dumps every score, a selection kernel marks the
      // nprobe best lists of each query
      knnx_index* c = ix->cent;
      if (!ix->ivf_scores) HIPCHK(hipMalloc(&ix->ivf_scores, (size_t)KNN_NQ * ix->ivf_nlist * sizeof(float)));
      HIPCHK(launch_prep(q_dev, nq, c->d, c->qfrag, c->thr_g, nullptr, 0, gate, st));
      ScanArgs ca{};
      ca.gate = gate;
      ca.X = c->rows;
      ca.N = c->ntotal;
      ca.d = c->d;
      ca.qfrag = c->qfrag;
      ca.nq = nq;
      ca.k = 1;
      ca.cap = 2;
      ca.grid = c->n_cu;
      ca.mode = 2;
      ca.thr_g = c->thr_g;
      ca.range_cap = (unsigned)ix->ivf_nlist;
      ca.range_s = ix->ivf_scores;
      HIPCHK(launch_scan(ca, st));
      HIPCHK(launch_ivf_worklist_from_scores(ix->ivf_scores, nq, np, ix->ivf_nlist, ix->ivf_masks, ix->ivf_tile0, ix->ivf_ntile,
                                             ix->ivf_size, ix->ivf_off, ix->ivf_work, ix->ivf_nwork, st));
    }
  }
  return 0;
}

// one scan of <= KNN_NQ queries already in HBM; results land in D_out/I_out (device, [nq, k])
static int scan_topk(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out,
                     hipStream_t st, const unsigned* gate) {
  const int cap = scan_cap(ix->d, k);
  if (cap < 0) return fail(KNNX_E_UNSUPPORTED, "k too large for the LDS queues at this d");
  if (ix->ivf_nlist) {
    int r = ivf_build_worklist(ix, q_dev, nq, st, gate);
    if (r) return r;
    ix->ivfm_last_blk = 0;
  }
  HIPCHK(launch_prep(q_dev, nq, ix->d, ix->qfrag, ix->thr_g, nullptr, 0, gate, st));
  ScanArgs a{};
  a.gate = gate;
  a.X = ix->rows;
  a.N = ix->ivf_nlist ? ix->capacity : ix->ntotal;
  a.d = ix->d;
  a.qfrag = ix->qfrag;
  a.nq = nq;
  a.k = k;
  a.cap = cap;
  a.grid = ix->n_cu;
  a.mode = 0;
  a.nt = ix->nt_loads;
  a.thr_g = ix->thr_g;
  a.part_s = ix->part_s;
  a.part_i = ix->part_i;
  a.part_n = ix->part_n;
  a.work = ix->ivf_nlist ? ix->ivf_work : nullptr;
  a.nwork = ix->ivf_nwork;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool prof = ix->prof && !gate;  // gated fallback scans normally exit at once: not a scan launch worth timing
  if (prof) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
  }
  HIPCHK(launch_scan(a, st));
  if (prof) {
    HIPCHK(hipEventRecord(e1, st));
    ix->prof_events.emplace_back(e0, e1);
  }
  HIPCHK(launch_merge_u32(ix->part_s, ix->part_i, ix->part_n, ix->n_cu, KNN_NQ, k, nq, k, ix->id_base,
                          ix->ivf_nlist ? ix->ivf_idmap : nullptr, D_out, I_out, gate, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// IVF, more than 32 queries: the multi-block pass.  The 32-query scan keeps one block's hi / lo fragments in the LDS (d * 128 B:
// 128 KiB at d = 1024), so a batch of B queries used to be ceil(B / 32) passes, each with its own coarse scan, work-list kernels,
// list scan and merge (~0.55 ms of small kernels and launch gaps per pass on config 5's shard: 4.4 of the 7.2 ms of a 256-query
// batch at nprobe 16).  Here the blocks run SIDE BY SIDE: one prep, one coarse scan, one work-list build, one list scan, one merge
// for up to IVFM_BLK x 32 queries; workgroup g of a scan serves block g % nblk (knn_kernels.hip: knn_scan_kernel, nblk).  A list
// probed by queries of one block is read once (as before); probed from two blocks it is read by both (the L2 / MALL may catch the
// second read: the blocks walk their lists at the same time).  Results are those of the 32-query path, bit for bit: every block
// runs the same exact hi / lo scan over the same candidate set (reference call: clip_back.py:357-369, nprobe on the IVF branch).
// ---------------------------------------------------------------------------------------------
constexpr int IVFM_BLK = 8;

static bool ivfm_usable(const knnx_index* ix, int nq, int k) {
  // (any batch size since the coarse quantiser of this pass is the score dump + radix select: at 32 queries and fewer it is one block)
  // (n_cu >= IVFM_BLK: every block needs a workgroup of its own, and the partial-list arrays hold n_cu slots -- KNNX_GRID can set fewer)
  return ix->ivf_nlist && ix->ivfm_ok && ix->cent && nq >= 1 && k <= KNNX_MAX_K_FAST && scan_cap(ix->d, k) > 0 && ix->n_cu >= IVFM_BLK;
}

// 0: buffers are there; 1: not available (the caller falls back to the 32-query passes; ivfm_ok is cleared)
static int ivfm_alloc(knnx_index* ix) {
  const size_t nl = (size_t)ix->ivf_nlist;
  hipError_t e = hipSuccess;
  if (!ix->ivfm_qfrag) {
    const size_t tiles = (size_t)std::max<int64_t>(ix->capacity / 32, 1);
    ix->ivfm_stride = (unsigned)tiles;
    e = hipMalloc(&ix->ivfm_qfrag, (size_t)IVFM_BLK * ix->d * 128);
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_thr_c, (size_t)IVFM_BLK * 32 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_thr_f, (size_t)IVFM_BLK * 32 * sizeof(int));
    // (the tile counters sit in the 16 words in front of the masks: one memset clears both, knn_kernels.hip launch_ivf_worklist*)
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_nwork, ((size_t)IVFM_BLK * nl + 16) * sizeof(unsigned));
    if (e == hipSuccess) ix->ivfm_masks = ix->ivfm_nwork + 16;
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_off, (size_t)IVFM_BLK * nl * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_work, (size_t)IVFM_BLK * tiles * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc(&ix->ivfm_scores, (size_t)IVFM_BLK * 32 * nl * sizeof(float));
  }
  if (e == hipSuccess) return 0;
  (void)hipGetLastError();
  hipFree(ix->ivfm_qfrag); ix->ivfm_qfrag = nullptr;
  hipFree(ix->ivfm_thr_c); ix->ivfm_thr_c = nullptr;
  hipFree(ix->ivfm_thr_f); ix->ivfm_thr_f = nullptr;
  ix->ivfm_masks = nullptr;  // (inside the ivfm_nwork allocation)
  hipFree(ix->ivfm_off); ix->ivfm_off = nullptr;
  hipFree(ix->ivfm_nwork); ix->ivfm_nwork = nullptr;
  hipFree(ix->ivfm_work); ix->ivfm_work = nullptr;
  hipFree(ix->ivfm_scores); ix->ivfm_scores = nullptr;
  ix->ivfm_ok = 0;
  return 1;
}

// one pass of 1 .. 32 IVFM_BLK queries already in HBM
static int scan_topk_ivf_multi(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out, hipStream_t st) {
  const int cap = scan_cap(ix->d, k);
  const int nblk = (nq + KNN_NQ - 1) / KNN_NQ;
  const int grid = std::max(1, ix->n_cu / nblk) * nblk;  // (part_* hold n_cu x 64 lists: grid <= n_cu whenever n_cu >= nblk)
  if (cap < 0 || nblk > IVFM_BLK || grid > std::max(ix->n_cu, nblk)) return fail(KNNX_E_STATE, "internal: multi-block IVF pass misuse");
  knnx_index* c = ix->cent;
  const int np = std::min(ix->ivf_nprobe, ix->ivf_nlist);
  HIPCHK(launch_prep_blocks(q_dev, nq, ix->d, ix->ivfm_qfrag, ix->ivfm_thr_c, ix->ivfm_thr_f, st));
  // coarse quantiser: ONE scan over the centroid rows dumps every block's scores (mode 2: no queues -- a top-nprobe queue scan of
  // 65 536 centroids spends its time pruning cold queues: 245 us for two blocks, profiles/r06k_*), a radix select marks the nprobe
  // best lists of every query (knn_kernels.hip: ivf_select_mark_kernel), whatever nprobe is
  ScanArgs ca{};
  ca.X = c->rows;
  ca.N = c->ntotal;
  ca.d = c->d;
  ca.qfrag = ix->ivfm_qfrag;
  ca.nq = nq;
  ca.grid = grid;
  ca.thr_g = ix->ivfm_thr_c;
  ca.nblk = nblk;
  ca.k = 1;
  ca.cap = 2;
  ca.mode = 2;
  ca.range_cap = (unsigned)ix->ivf_nlist;
  ca.range_s = ix->ivfm_scores;
  HIPCHK(launch_scan(ca, st));
  HIPCHK(launch_ivf_worklist_from_scores(ix->ivfm_scores, nq, np, ix->ivf_nlist, ix->ivfm_masks, ix->ivf_tile0, ix->ivf_ntile,
                                         ix->ivf_size, ix->ivfm_off, ix->ivfm_work, ix->ivfm_nwork, st, ix->ivfm_stride));
  if (ix->prof)  // tiles of the union of the blocks' lists (what one pass over shared lists would read): knnx_ivf_last_scan_tiles
    HIPCHK(launch_ivf_union_tiles(ix->ivfm_masks, nblk, ix->ivf_nlist, ix->ivf_ntile, ix->ivfm_nwork + IVFM_BLK, st));
  ScanArgs a{};
  a.X = ix->rows;
  a.N = ix->capacity;
  a.d = ix->d;
  a.qfrag = ix->ivfm_qfrag;
  a.nq = nq;
  a.k = k;
  a.cap = cap;
  a.grid = grid;
  a.mode = 0;
  a.thr_g = ix->ivfm_thr_f;
  a.part_s = ix->part_s;
  a.part_i = ix->part_i;
  a.part_n = ix->part_n;
  a.work = ix->ivfm_work;
  a.nwork = ix->ivfm_nwork;
  a.nblk = nblk;
  a.work_stride = ix->ivfm_stride;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ix->prof) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
  }
  HIPCHK(launch_scan(a, st));
  if (ix->prof) {
    HIPCHK(hipEventRecord(e1, st));
    ix->prof_events.emplace_back(e0, e1);
  }
  HIPCHK(launch_merge_u32(ix->part_s, ix->part_i, ix->part_n, grid - nblk + 1, KNN_NQ, k, nq, k, ix->id_base, ix->ivf_idmap, D_out, I_out,
                          nullptr, st, KNN_NQ, ix->ivfm_nwork, nblk, grid));
  ix->ivfm_last_blk = nblk;
  ix->ivfm_union_valid = ix->prof;
  return 0;
}

// the exact scan of every 32-query group of a proof-based scan, gated on the device by gate[g] -- or (ix->defer_fb) left to the host
// entry point, which reads the gates back: see knnx_index::defer_fb
static int fallback_groups(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out, hipStream_t st, int kind) {
  unsigned* need = kind == 1 ? ix->rq_need : ix->wide_need;
  unsigned* gate = kind == 1 ? ix->rq_gate : ix->wide_gate;
  if (ix->defer_fb) {
    ix->defer_kind = kind;
    ix->defer_nq = nq;
    return 0;
  }
  for (int g = 0; g * KNN_NQ < nq; ++g) {
    const int q0 = g * KNN_NQ, n = std::min(KNN_NQ, nq - q0);
    int r = scan_topk(ix, q_dev + (size_t)q0 * ix->d, n, k, ix->wide_Dfb, ix->wide_Ifb, st, gate + g);
    if (r) return r;
    HIPCHK(launch_select(need, q0, n, k, ix->wide_Dfb, ix->wide_Ifb, D_out, I_out, st));
  }
  return 0;
}

// LDS capacity of the 64-queue wide scan at this d
static int wide_cap(int d) {
  long avail = (long)KNN_LDS_BYTES - (long)d * 128 - KNN_NQ_MAX * 8 - 16;
  int cap = (int)(avail / (KNN_NQ_MAX * 8));
  cap = std::min(cap, 128) & ~1;
  return cap >= KNN_WIDE_KW + 16 ? cap : -1;
}
static bool wide_usable(const knnx_index* ix, int nq, int k) {
  return ix->wide_ok && !ix->ivf_nlist && nq > KNN_NQ && k <= KNN_WIDE_MAX_K && wide_cap(ix->d) > 0;
}

// one WIDE scan of 33..64 queries: approximate top-64 by fp16-hi score in ONE pass over HBM, exact re-scoring of the 64
// candidates, proof of exactness per query; a query whose proof fails gets the exact 32-query scan of its half, which
// is launched unconditionally but exits at once unless its gate was set by the proof kernel (no host round trip).
static int scan_topk_wide(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out, hipStream_t st) {
  const int cap = wide_cap(ix->d);
  HIPCHK(launch_prep(q_dev, nq, ix->d, ix->qfrag, ix->thr_g, nullptr, 1, nullptr, st));
  ScanArgs a{};
  a.X = ix->rows;
  a.N = ix->ntotal;
  a.d = ix->d;
  a.qfrag = ix->qfrag;
  a.nq = nq;
  a.k = KNN_WIDE_KW;
  a.cap = cap;
  a.grid = ix->n_cu;
  a.mode = 0;
  a.wide = 1;
  a.thr_g = ix->thr_g;
  a.part_s = ix->part_s;
  a.part_i = ix->part_i;
  a.part_n = ix->part_n;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ix->prof) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
  }
  HIPCHK(launch_scan(a, st));
  if (ix->prof) {
    HIPCHK(hipEventRecord(e1, st));
    ix->prof_events.emplace_back(e0, e1);
  }
  HIPCHK(launch_merge_u32(ix->part_s, ix->part_i, ix->part_n, ix->n_cu, KNN_NQ_MAX, KNN_WIDE_KW, nq, KNN_WIDE_KW, 0, nullptr,
                          ix->wide_approx, ix->wide_cand, nullptr, st));
  HIPCHK(launch_rescore(ix->rows, ix->d, q_dev, ix->wide_cand, ix->wide_approx, nq, KNN_WIDE_KW, k, ix->id_base, ix->maxnorm,
                        D_out, I_out, ix->wide_need, ix->wide_gate, ix->stats, st));
  return fallback_groups(ix, q_dev, nq, k, D_out, I_out, st, 2);
}


// ---- scratch guard: call at the start / end of every launch sequence that uses the handle's scratch (caller holds mu)
static int scratch_acquire(knnx_index* ix, hipStream_t st) {
  if (ix->ev_valid) HIPCHK(hipStreamWaitEvent(st, ix->ev_scratch, 0));
  return 0;
}
static int scratch_release(knnx_index* ix, hipStream_t st) {
  HIPCHK(hipEventRecord(ix->ev_scratch, st));
  ix->ev_valid = true;
  return 0;
}

// ---- RQ scan: up to rq_queries_per_pass(d) queries in ONE pass over HBM (see knn_rq_kernels.hip)
static bool rq_usable(const knnx_index* ix, int nq, int k) {
  return ix->rq_ok && ix->wide_ok && !ix->ivf_nlist && nq > KNN_NQ_MAX && k <= KNN_WIDE_MAX_K && rq_queries_per_pass(ix->d) > 0 &&
         ix->ntotal >= ix->rq_min_rows && scan_cap(ix->d, KNN_WIDE_KW) > 0;
}
static int rq_alloc(knnx_index* ix) {
  if (ix->rq_qfrag) return 0;
  const size_t Q = KNN_RQ_MAX;
  HIPCHK(hipMalloc(&ix->rq_qfrag, (size_t)(Q / 32) * (ix->d / 16) * 64 * 16));
  HIPCHK(hipMalloc(&ix->rq_thr, Q * sizeof(float)));
  HIPCHK(hipMalloc(&ix->rq_cnt, Q * sizeof(unsigned)));
  HIPCHK(hipMalloc(&ix->rq_lost, Q * sizeof(unsigned)));
  HIPCHK(hipMalloc(&ix->rq_need, Q * sizeof(unsigned)));
  HIPCHK(hipMalloc(&ix->rq_gate, 8 * sizeof(unsigned)));
  HIPCHK(hipMalloc(&ix->rq_cntc, Q * sizeof(int)));
  HIPCHK(hipMalloc(&ix->rq_hit_s, Q * KNN_RQ_CAP * sizeof(float)));
  HIPCHK(hipMalloc(&ix->rq_hit_r, Q * KNN_RQ_CAP * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&ix->rq_samp, Q * KNN_WIDE_KW * sizeof(float)));
  HIPCHK(hipMalloc(&ix->rq_samp_i, Q * KNN_WIDE_KW * sizeof(int64_t)));
  HIPCHK(hipMalloc(&ix->rq_m_qfrag, (Q / KNN_NQ) * (size_t)ix->d * 128));  // fragment images of the sample scans of one batch, side by side
  HIPCHK(hipMalloc(&ix->rq_m_thr, Q * sizeof(int)));
  const size_t G = (size_t)ix->n_cu;
  HIPCHK(hipEventCreateWithFlags(&ix->rq_fork, hipEventDisableTiming));
  for (int i = 0; i < knnx_index::RQ_SIDE; ++i) {
    HIPCHK(hipStreamCreateWithFlags(&ix->rq_side[i], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ix->rq_join[i], hipEventDisableTiming));
    HIPCHK(hipMalloc(&ix->rq_s_qfrag[i], (size_t)ix->d * 128));
    HIPCHK(hipMalloc(&ix->rq_s_thr_g[i], KNN_NQ_MAX * sizeof(int)));
    HIPCHK(hipMalloc(&ix->rq_s_part_s[i], G * KNN_NQ_MAX * KNNX_MAX_K_FAST * sizeof(float)));
    HIPCHK(hipMalloc(&ix->rq_s_part_i[i], G * KNN_NQ_MAX * KNNX_MAX_K_FAST * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&ix->rq_s_part_n[i], G * KNN_NQ_MAX * sizeof(int)));
  }
  return 0;
}
// thresholds of a register-stationary pass: the 64-query scan over every S-th tile, S = min(stride_cap, tiles / 4096); the J-th best
// sample score of query q -> ix->rq_samp[q * KNN_WIDE_KW + J - 1] (knn_kernels.h: KNN_RQ_STRIDE).  Returns S and J through the pointers.
// hits_target > 0: J = hits_target / S (>= 6) instead of the rule above -- the threshold that leaves ~hits_target rows of the index above it
static int rq_sample_pass(knnx_index* ix, const float* q_dev, int nq, int k, int stride_cap, hipStream_t st, int* tstride_out, int* J_out,
                          bool* wide_samp_out, int hits_target = 0) {
  const int d = ix->d;
  const int64_t ntiles = (ix->ntotal + 31) / 32;
  const int tstride = (int)std::max<int64_t>(1, std::min<int64_t>(stride_cap, ntiles / 4096));
  // (the 64-query wide scan where its LDS queues fit, d <= 768; at d = 1024 the exact 32-query scan, whose scores are exact
  // rather than fp16-hi: the threshold then gets a slack of a few eps so that the sample rows themselves still reach it)
  const bool wide_samp = wide_cap(d) > 0;
  const int gsz = wide_samp ? KNN_NQ_MAX : KNN_NQ;
  // The sample scans cost a fixed ~0.45 ms each whatever their grid (cold candidate queues: every workgroup floods and prunes
  // its queues a few times before its thresholds bite; 256 -> 64 workgroups: same time, tools/rq_sample_grid.py), and one of
  // them needs a whole CU's LDS per workgroup.  So the groups of a batch run side by side, each on its own quarter of the CUs
  // (own stream, own scratch): 4 x (prep + scan + merge) in the time of one.
  const int ngroups = (nq + gsz - 1) / gsz;
  // Round 6: the groups of a batch as blocks of ONE launch (knn_scan_kernel, nblk: workgroup g serves group g % ngroups) instead of
  // one stream each -- four streams only run side by side while the runtime has four hardware queues to give them: in
  // profiles/r06r_knn_b256_timeline.log the fourth scan starts when the first three have finished (0.5 ms of a 256-query batch).
  // KNNX_SAMPLE_STREAMS=1 keeps the streams.
  if (ngroups > 1 && ix->sample_one_launch && ngroups <= KNN_RQ_MAX / KNN_NQ && ix->n_cu >= ngroups) {
    const int grid = std::max(1, ix->n_cu / ngroups) * ngroups;
    // only the J-th best sample score of a query is used: the queues keep J entries (>= 8), not 64 -- their thresholds rise sooner and
    // the cold-queue pruning that makes up most of this scan's time (see above) ends sooner
    const int jf0 = std::min(KNN_WIDE_KW, k + KNN_RQ_MARGIN);
    const int Jq = hits_target > 0 ? std::max(6, std::min(jf0, (hits_target + tstride - 1) / tstride))
                                   : (tstride <= 128 ? jf0 : std::max(6, std::min(jf0, (jf0 * 128 + tstride - 1) / tstride)));
    const int ks = std::min(KNN_WIDE_KW, std::max(8, Jq));
    if (grid <= ix->n_cu) {
      HIPCHK(launch_prep_blocks(q_dev, nq, d, ix->rq_m_qfrag, ix->rq_m_thr, nullptr, st, wide_samp ? 1 : 0));
      ScanArgs a{};
      a.X = ix->rows;
      a.N = ix->ntotal;
      a.d = d;
      a.qfrag = ix->rq_m_qfrag;
      a.nq = nq;
      a.k = ks;
      a.cap = wide_samp ? wide_cap(d) : scan_cap(d, KNN_WIDE_KW);
      a.grid = grid;
      a.mode = 0;
      a.wide = wide_samp ? 1 : 0;
      a.tstride = tstride;
      a.thr_g = ix->rq_m_thr;
      a.part_s = ix->part_s;
      a.part_i = ix->part_i;
      a.part_n = ix->part_n;
      a.nblk = ngroups;
      HIPCHK(launch_scan(a, st));
      HIPCHK(launch_merge_u32(ix->part_s, ix->part_i, ix->part_n, grid / ngroups, gsz, ks, nq, KNN_WIDE_KW, 0, nullptr, ix->rq_samp,
                              ix->rq_samp_i, nullptr, st, gsz));  // (rows of 64 with the entries past ks padded: the preps read entry J - 1)
      *tstride_out = tstride;
      *J_out = Jq;
      *wide_samp_out = wide_samp;
      return 0;
    }
  }
  const int lanes = std::min(ngroups, 1 + knnx_index::RQ_SIDE);
  const int sgrid = std::max(1, std::min(ix->n_cu, ix->rq_sample_grid > 0 ? ix->rq_sample_grid : ix->n_cu / lanes));
  if (lanes > 1) HIPCHK(hipEventRecord(ix->rq_fork, st));
  for (int g = 0; g * gsz < nq; ++g) {
    const int q0 = g * gsz, n = std::min(gsz, nq - q0);
    const int side = g % lanes;  // 0: the caller's stream and the index's own scratch
    hipStream_t sg = side == 0 ? st : ix->rq_side[side - 1];
    _Float16* qf = side == 0 ? ix->qfrag : ix->rq_s_qfrag[side - 1];
    int* thr_g = side == 0 ? ix->thr_g : ix->rq_s_thr_g[side - 1];
    float* part_s = side == 0 ? ix->part_s : ix->rq_s_part_s[side - 1];
    uint32_t* part_i = side == 0 ? ix->part_i : ix->rq_s_part_i[side - 1];
    int* part_n = side == 0 ? ix->part_n : ix->rq_s_part_n[side - 1];
    if (side != 0 && g < lanes) HIPCHK(hipStreamWaitEvent(sg, ix->rq_fork, 0));
    HIPCHK(launch_prep(q_dev + (size_t)q0 * d, n, d, qf, thr_g, nullptr, wide_samp ? 1 : 0, nullptr, sg));
    ScanArgs a{};
    a.X = ix->rows;
    a.N = ix->ntotal;
    a.d = d;
    a.qfrag = qf;
    a.nq = n;
    a.k = KNN_WIDE_KW;
    a.cap = wide_samp ? wide_cap(d) : scan_cap(d, KNN_WIDE_KW);
    a.grid = sgrid;
    a.mode = 0;
    a.wide = wide_samp ? 1 : 0;
    a.tstride = tstride;
    a.thr_g = thr_g;
    a.part_s = part_s;
    a.part_i = part_i;
    a.part_n = part_n;
    HIPCHK(launch_scan(a, sg));
    HIPCHK(launch_merge_u32(part_s, part_i, part_n, sgrid, gsz, KNN_WIDE_KW, n, KNN_WIDE_KW, 0, nullptr,
                            ix->rq_samp + (size_t)q0 * KNN_WIDE_KW, ix->rq_samp_i + (size_t)q0 * KNN_WIDE_KW, nullptr, sg));
  }
  for (int side = 1; side < lanes; ++side) {
    HIPCHK(hipEventRecord(ix->rq_join[side - 1], ix->rq_side[side - 1]));
    HIPCHK(hipStreamWaitEvent(st, ix->rq_join[side - 1], 0));
  }
  const int jfull = std::min(KNN_WIDE_KW, k + KNN_RQ_MARGIN);
  *tstride_out = tstride;
  *J_out = tstride <= 128 ? jfull : std::max(6, std::min(jfull, (jfull * 128 + tstride - 1) / tstride));
  *wide_samp_out = wide_samp;
  return 0;
}

static int scan_topk_rq(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out, hipStream_t st) {
  int r = rq_alloc(ix);
  if (r) return r;
  const int d = ix->d;
  // 1. thresholds: the sample pass
  int tstride = 1, J = 1;
  bool wide_samp = false;
  r = rq_sample_pass(ix, q_dev, nq, k, KNN_RQ_STRIDE, st, &tstride, &J, &wide_samp);
  if (r) return r;
  HIPCHK(launch_rq_prep(q_dev, nq, d, ix->rq_qfrag, ix->rq_samp, KNN_WIDE_KW, J, wide_samp ? 0.f : 1e-3f, ix->rq_thr, ix->rq_cnt,
                        ix->rq_lost, st));
  // 2. the pass over the whole index
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ix->prof) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
  }
  HIPCHK(launch_rq_scan(ix->rows, ix->ntotal, d, nq, ix->rq_qfrag, ix->rq_thr, ix->rq_cnt, KNN_RQ_CAP, ix->rq_hit_s, ix->rq_hit_r,
                        ix->rq_lost, nullptr, ix->n_cu, st));
  if (ix->prof) {
    HIPCHK(hipEventRecord(e1, st));
    ix->prof_events.emplace_back(e0, e1);
  }
  // 3. exact scores of the hits, exact top-k among them, proof
  HIPCHK(launch_rq_rescore(ix->rows, d, q_dev, nq, ix->rq_cnt, KNN_RQ_CAP, ix->rq_hit_s, ix->rq_hit_r, ix->rq_cntc, st));
  HIPCHK(launch_merge_u32(ix->rq_hit_s, ix->rq_hit_r, ix->rq_cntc, 1, nq, (int)KNN_RQ_CAP, nq, k, ix->id_base, nullptr, D_out, I_out,
                          nullptr, st));
  HIPCHK(launch_rq_proof(q_dev, nq, d, k, D_out, ix->rq_thr, ix->rq_cnt, KNN_RQ_CAP, ix->rq_lost, ix->maxnorm, ix->rq_need,
                         ix->rq_gate, ix->stats, st));
  // 4. unproven queries: the exact scan of their 32-query group, gated on the device (or left to the host entry point)
  return fallback_groups(ix, q_dev, nq, k, D_out, I_out, st, 1);
}

// ---- int8 first stage (knn_rq_kernels.hip, "int8 first stage"): 1 .. 256 queries in ONE pass over the int8 copy of the rows
static bool i8_usable(const knnx_index* ix, int nq, int k) {
  return ix->i8_ok && ix->rq_ok && ix->wide_ok && !ix->ivf_nlist && nq >= 1 && k <= KNN_WIDE_MAX_K && i8_supported(ix->d) &&
         ix->ntotal >= ix->rq_min_rows && scan_cap(ix->d, KNN_WIDE_KW) > 0;
}
// allocate (first use) and build (first use, and after the rows changed) the int8 copy; 1: ready, 0: not available (the caller takes
// another path; out of memory turns the feature off for this index), < 0: error.
// The copy may be PARTIAL (round 5): when ntotal * d bytes do not fit next to the fp16 rows -- BASELINE's headline shard, 125 M x 768 =
// 192 GB of fp16 on a 288 GB part, leaves room for ~80 M rows of int8 -- or KNNX_I8_MAX_BYTES caps it, rows [0, i8_nrows) get the int8
// first stage and the rows behind them are scanned by the fp16 register-stationary pass into the same hit lists (scan_topk_i8).
static int i8_ensure(knnx_index* ix, hipStream_t st) {
  if (ix->i8_valid) return 1;
  const size_t Q = KNN_RQ_MAX;
  auto give_up = [&]() {
    (void)hipGetLastError();
    ix->i8_ok = 0;
    i8_release(ix);
    return 0;
  };
  if (!ix->i8_colscale) {
    if (hipMalloc(&ix->i8_colscale, ix->d * sizeof(float)) != hipSuccess || hipMalloc(&ix->i8_colmax, ix->d * sizeof(int)) != hipSuccess ||
        hipMalloc(&ix->i8_ab, 2 * sizeof(int)) != hipSuccess || hipMalloc(&ix->i8_qfrag, 2 * Q * ix->d) != hipSuccess ||
        hipMalloc(&ix->i8_qdom, 2 * Q * 4) != hipSuccess ||
        hipMalloc(&ix->i8_thr, Q * sizeof(int)) != hipSuccess || hipMalloc(&ix->i8_lb, Q * sizeof(float)) != hipSuccess ||
        hipMalloc(&ix->i8_hit_s, Q * KNN_I8_CAP * sizeof(float)) != hipSuccess ||
        hipMalloc(&ix->i8_hit_r, Q * KNN_I8_CAP * sizeof(uint32_t)) != hipSuccess)
      return give_up();
  }
  // everything else the flat scans allocate lazily comes FIRST, so that a partial copy is sized around it, not the other way round
  int r = rq_alloc(ix);
  if (r) return r;
  const auto tiles32 = [](int64_t rows) { return (rows + 31) & ~(int64_t)31; };
  if (ix->i8_cap_rows < ix->ntotal) {
    // (an owned index grows by halves: follow its capacity, so that a stream of add() calls does not reallocate every time)
    int64_t want = tiles32(std::max(ix->ntotal, ix->borrowed ? ix->ntotal : ix->capacity));
    if (ix->i8_budget > 0) want = std::min(want, (ix->i8_budget / ix->d) & ~(int64_t)31);
    int8_t* nr = nullptr;
    if (want > ix->i8_cap_rows && hipMalloc(&nr, (size_t)want * ix->d) != hipSuccess) {
      (void)hipGetLastError();
      nr = nullptr;
      // what is free, less a reserve for the pools that are sized per request (range scans: up to 4 GiB, large-k scratch, reconstruct)
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess) return give_up();
      const size_t have = (size_t)ix->i8_cap_rows * ix->d;  // the old image is released below: its bytes count as free
      const size_t reserve = (size_t)8 << 30;
      want = fr + have > reserve ? (int64_t)(((fr + have - reserve) / (size_t)ix->d) & ~(size_t)31) : 0;
      if (want > tiles32(ix->ntotal)) want = tiles32(ix->ntotal);
      // a copy of less than a quarter of the rows saves under an eighth of the pass: not worth the memory
      if (want < ix->ntotal / 4 || want <= ix->i8_cap_rows) {
        if (want <= ix->i8_cap_rows && ix->i8_cap_rows >= ix->ntotal / 4) want = 0;  // keep the image that exists
        else return give_up();
      }
      if (want > 0) {
        // the old image has to go first (its bytes are part of the estimate); what it held is quantised again below
        i8_release(ix);
        if (hipMalloc(&nr, (size_t)want * ix->d) != hipSuccess) return give_up();
      }
    }
    if (nr) {
      if (ix->i8_rows && ix->i8_nrows > 0) {
        // whole tiles of the old image carry over (tile-ordered: a prefix of tiles is a prefix of bytes)
        const int64_t keep = std::min(ix->i8_nrows & ~(int64_t)31, want);
        hipError_t e = keep > 0 ? hipMemcpyAsync(nr, ix->i8_rows, (size_t)keep * ix->d, hipMemcpyDeviceToDevice, st) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
          hipFree(nr);
          return fail(KNNX_E_HIP, std::string("int8 copy: ") + hipGetErrorString(e));
        }
        ix->i8_nrows = keep;
      }
      hipFree(ix->i8_rows);
      ix->i8_rows = nr;
      ix->i8_cap_rows = want;
    }
  }
  const int64_t n8 = std::min(ix->ntotal, ix->i8_cap_rows);  // rows the image will hold
  if (n8 < ix->ntotal / 4) return give_up();
  if (ix->i8_nrows > 0 && ix->i8_nrows <= n8 && ix->ntotal < 2 * ix->i8_scale_rows) {
    // rows were appended: quantise the new ones with the scales that exist (values beyond them clamp; A and B grow with what is stored)
    if (ix->i8_nrows < n8)
      HIPCHK(launch_i8_quant(ix->rows, n8, ix->i8_nrows, n8, ix->d, ix->i8_colscale, ix->i8_dom, ix->i8_rows, ix->i8_ab, st));
  } else {
    HIPCHK(launch_i8_scales(ix->rows, ix->ntotal, ix->d, ix->i8_colmax, ix->i8_colscale, ix->i8_ab, st));
    ix->i8_scale_rows = ix->ntotal;
    // The form of the first stage is part of the build (a readback of d floats: the build is about to move the whole index).  A query's
    // u = q * c has ONE scale, so a few columns much larger than the rest (CLIP embeddings have them) leave the others' components in
    // the rounding error.  Columns whose scale is more than 3 x the median one are "dominant":
    //   1 .. 4 of them: they move to bytes 0..3 of each row of the image and the scan handles them with 14-bit query digits (I8Dom) --
    //                   one plane, 256 queries per pass
    //   more:           two int8 planes per query (twice the MFMAs, 128 queries per pass)
    std::vector<float> cs((size_t)ix->d);
    HIPCHK(hipMemcpyAsync(cs.data(), ix->i8_colscale, (size_t)ix->d * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<float> sorted(cs);
    std::nth_element(sorted.begin(), sorted.begin() + ix->d / 2, sorted.end());
    const float cmed = sorted[(size_t)ix->d / 2];
    std::vector<int> big;
    for (int c = 0; c < ix->d; ++c)
      if (cs[(size_t)c] > 3.f * cmed) big.push_back(c);
    ix->i8_dom = I8Dom();
    if (ix->i8_planes_env) {
      ix->i8_planes = ix->i8_planes_env;
    } else if (!big.empty() && big.size() <= 4 && !ix->i8_dom_off) {
      ix->i8_planes = 1;
      // the permutation that brings them to positions 0 .. n - 1: a swap per dominant column
      std::vector<int> perm((size_t)ix->d);
      for (int c = 0; c < ix->d; ++c) perm[(size_t)c] = c;
      for (size_t j = 0; j < big.size(); ++j) {
        const size_t at = (size_t)(std::find(perm.begin(), perm.end(), big[j]) - perm.begin());
        std::swap(perm[j], perm[at]);
      }
      ix->i8_dom.n = (int)big.size();
      for (int c = 0; c < ix->d; ++c)
        if (perm[(size_t)c] != c) {
          ix->i8_dom.pos[ix->i8_dom.nfix] = c;
          ix->i8_dom.src[ix->i8_dom.nfix] = perm[(size_t)c];
          ++ix->i8_dom.nfix;  // (at most two positions per swap: 8)
        }
    } else {
      ix->i8_planes = big.empty() ? 1 : 2;
    }
    HIPCHK(launch_i8_quant(ix->rows, n8, 0, n8, ix->d, ix->i8_colscale, ix->i8_dom, ix->i8_rows, ix->i8_ab, st));
  }
  ix->i8_nrows = n8;
  ix->i8_valid = true;
  return 1;
}
static int scan_topk_i8(knnx_index* ix, const float* q_dev, int nq, int k, float* D_out, int64_t* I_out, hipStream_t st) {
  int r = rq_alloc(ix);
  if (r) return r;
  const int d = ix->d;
  const int64_t n8 = ix->i8_nrows;          // rows [0, n8) have an int8 image ...
  const int64_t nrest = ix->ntotal - n8;    // ... the rows behind them are scanned in fp16 (partial copy, i8_ensure)
  // 1. thresholds: the exact sample pass (denser than the fp16 path's: the int8 bound widens the admission band, a tighter
  // threshold buys the hits back)
  int tstride = 1, J = 1;
  bool wide_samp = false;
  const int64_t ntiles = (ix->ntotal + 31) / 32;
  // Up to 64 queries: ONE exact sample pass over every 32nd tile (one 64-query scan, ~0.8 ms at 100 M rows).  More: four such scans
  // side by side cost ~3 ms of a 25 ms batch, so the threshold comes in two levels -- the coarse exact sample of the fp16 path (every
  // 763rd tile), then an int8 pass over every 32nd tile with THAT threshold (one pass for all the queries: 2.4 GB), whose hits are
  // re-scored exactly; their J-th best score is the threshold of the pass over everything.
  bool two_level = nq > KNN_NQ_MAX && ntiles >= (int64_t)4096 * KNN_I8_STRIDE;
  // KNNX_I8_ONE_LEVEL=S (round 6 experiment, OFF by default): ONE exact sample over every S-th tile with J = 1536 / S instead of the two
  // levels -- the same expected number of index rows above the threshold (J S = 48 x 32) from one scan instead of scan + int8 sample
  // pass + re-score + merge + second prep.  S = 256 (J = 6): 20.2 against 21.3 ms per 256-query batch on the isotropic 100 M x 768 index,
  // same ids and scores -- and 175 ms with 39 fallbacks on the corpus with dominant columns (profiles/r06af_*): the J-th best of a sparse
  // sample is a NOISY threshold (S x Gamma(6): +-41 %), and where the score density near it is steep a threshold one sigma low admits
  // several times the rows; the second level's threshold (48th best of a 1/32 sample: +-14 %) is what keeps the hit lists bounded.
  static const int one_level = [] { const char* e = getenv("KNNX_I8_ONE_LEVEL"); return e ? atoi(e) : 0; }();
  if (two_level && one_level > 0 && ix->sample_one_launch) {
    two_level = false;
    r = rq_sample_pass(ix, q_dev, nq, k, one_level, st, &tstride, &J, &wide_samp, 1536);
  } else
  r = rq_sample_pass(ix, q_dev, nq, k, two_level ? KNN_RQ_STRIDE : KNN_I8_STRIDE, st, &tstride, &J, &wide_samp);
  if (r) return r;
  float* thr_rest = nrest > 0 ? ix->rq_thr : nullptr;
  const int8_t* qdom = ix->i8_dom.n > 0 ? ix->i8_qdom : nullptr;
  HIPCHK(launch_i8_prep(q_dev, nq, d, ix->i8_colscale, ix->i8_dom, ix->i8_ab, ix->maxnorm, ix->rq_samp, KNN_WIDE_KW, J, ix->i8_planes, 0,
                        ix->i8_qfrag, ix->i8_qdom, ix->i8_thr, ix->i8_lb, thr_rest, ix->rq_cnt, ix->rq_lost, st));
  if (two_level) {
    HIPCHK(launch_rq8_scan(ix->i8_rows, n8, d, nq, ix->i8_planes, ix->i8_qfrag, qdom, ix->i8_thr, ix->rq_cnt, KNN_I8_CAP, ix->i8_hit_s,
                           ix->i8_hit_r, ix->rq_lost, ix->n_cu, KNN_I8_STRIDE, st));
    HIPCHK(launch_rq_rescore(ix->rows, d, q_dev, nq, ix->rq_cnt, KNN_I8_CAP, ix->i8_hit_s, ix->i8_hit_r, ix->rq_cntc, st));
    // the exact top-64 of the sampled hits, where the sample passes put theirs: the refining prep reads its J-th entry
    HIPCHK(launch_merge_u32(ix->i8_hit_s, ix->i8_hit_r, ix->rq_cntc, 1, nq, (int)KNN_I8_CAP, nq, KNN_WIDE_KW, 0, nullptr, ix->rq_samp,
                            ix->rq_samp_i, nullptr, st));
    const int J2 = std::min(KNN_WIDE_KW, k + KNN_RQ_MARGIN);
    HIPCHK(launch_i8_prep(q_dev, nq, d, ix->i8_colscale, ix->i8_dom, ix->i8_ab, ix->maxnorm, ix->rq_samp, KNN_WIDE_KW, J2, ix->i8_planes, 1,
                          ix->i8_qfrag, ix->i8_qdom, ix->i8_thr, ix->i8_lb, thr_rest, ix->rq_cnt, ix->rq_lost, st));
  }
  // 2. the pass over the int8 rows
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ix->prof) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
  }
  HIPCHK(launch_rq8_scan(ix->i8_rows, n8, d, nq, ix->i8_planes, ix->i8_qfrag, qdom, ix->i8_thr, ix->rq_cnt, KNN_I8_CAP, ix->i8_hit_s, ix->i8_hit_r,
                         ix->rq_lost, ix->n_cu, 1, st));
  // 2b. the rows without an int8 image: the fp16 register-stationary pass, admission threshold T - eps_hi (knn_i8_prep_kernel), into
  // the SAME hit lists (global row = n8 + row of the slice); the lower bound T of the proof covers both parts
  if (nrest > 0) {
    const int per = rq_queries_per_pass(d);
    const int blk_halves = (d / 32) * 64 * 8;  // fp16 values of one 16-query fragment block (knn_rq_prep_kernel)
    (void)blk_halves;
    for (int q0 = 0; q0 < nq; q0 += per) {
      const int n = std::min(per, nq - q0);
      HIPCHK(launch_rq_prep(q_dev + (size_t)q0 * d, n, d, ix->rq_qfrag, nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, st));  // fragments only
      HIPCHK(launch_rq_scan(ix->rows + (size_t)n8 * d, nrest, d, n, ix->rq_qfrag, ix->rq_thr + q0, ix->rq_cnt + q0, KNN_I8_CAP,
                            ix->i8_hit_s + (size_t)q0 * KNN_I8_CAP, ix->i8_hit_r + (size_t)q0 * KNN_I8_CAP, ix->rq_lost + q0, nullptr,
                            ix->n_cu, st, (uint32_t)n8));
    }
    ix->i8_rest_served += (unsigned long long)nq;
  }
  if (ix->prof) {
    HIPCHK(hipEventRecord(e1, st));
    ix->prof_events.emplace_back(e0, e1);
  }
  // 3. exact scores of the hits (fp16 rows), exact top-k among them, proof
  HIPCHK(launch_rq_rescore(ix->rows, d, q_dev, nq, ix->rq_cnt, KNN_I8_CAP, ix->i8_hit_s, ix->i8_hit_r, ix->rq_cntc, st));
  HIPCHK(launch_merge_u32(ix->i8_hit_s, ix->i8_hit_r, ix->rq_cntc, 1, nq, (int)KNN_I8_CAP, nq, k, ix->id_base, nullptr, D_out, I_out,
                          nullptr, st));
  HIPCHK(launch_i8_proof(nq, k, D_out, ix->i8_lb, ix->rq_cnt, KNN_I8_CAP, ix->rq_lost, ix->rq_need, ix->rq_gate, ix->stats, st));
  // 4. unproven queries (a hit list overflowed): the exact scan of their 32-query group, gated on the device (or left to the host
  // entry point)
  ix->i8_served += (unsigned long long)nq;
  return fallback_groups(ix, q_dev, nq, k, D_out, I_out, st, 1);
}

// how many of `remaining` queries the next scan step takes (the same choice scan_step makes)
static int step_queries(knnx_index* ix, int remaining, int k) {
  if (ivfm_usable(ix, remaining, k)) return std::min(IVFM_BLK * KNN_NQ, remaining);
  // (two planes: 128 queries per int8 pass -- a batch of more goes to the fp16 register-stationary pass, 36 ms for 256 against 2 x 19.6)
  if (i8_usable(ix, remaining, k) && !(ix->i8_planes == 2 && remaining > 128 && rq_usable(ix, remaining, k)))
    return std::min(ix->i8_planes == 2 ? 128 : KNN_RQ_MAX, remaining);
  if (rq_usable(ix, remaining, k)) return std::min(rq_queries_per_pass(ix->d), remaining);
  if (wide_usable(ix, remaining, k)) return std::min(KNN_NQ_MAX, remaining);
  return std::min(KNN_NQ, remaining);
}

// one step of a search over device buffers: picks the widest scan that serves the remaining queries; returns the number taken
static int scan_step(knnx_index* ix, const float* q_dev, int remaining, int k, float* D_out, int64_t* I_out, hipStream_t st,
                     int* taken) {
  int nq, r;
  if (ivfm_usable(ix, remaining, k) && ivfm_alloc(ix) == 0) {
    nq = std::min(IVFM_BLK * KNN_NQ, remaining);
    r = scan_topk_ivf_multi(ix, q_dev, nq, k, D_out, I_out, st);
    *taken = nq;
    return r;
  }
  if (i8_usable(ix, remaining, k)) {
    r = i8_ensure(ix, st);
    if (r < 0) return r;
    if (r == 1 && !(ix->i8_planes == 2 && remaining > 128 && rq_usable(ix, remaining, k))) {
      nq = std::min(ix->i8_planes == 2 ? 128 : KNN_RQ_MAX, remaining);  // (two planes: four waves x 32 queries per pass)
      r = scan_topk_i8(ix, q_dev, nq, k, D_out, I_out, st);
      *taken = nq;
      return r;
    }
  }
  if (rq_usable(ix, remaining, k)) {
    nq = std::min(rq_queries_per_pass(ix->d), remaining);
    r = scan_topk_rq(ix, q_dev, nq, k, D_out, I_out, st);
  } else if (wide_usable(ix, remaining, k)) {
    nq = std::min(KNN_NQ_MAX, remaining);
    r = scan_topk_wide(ix, q_dev, nq, k, D_out, I_out, st);
  } else {
    nq = std::min(KNN_NQ, remaining);
    r = scan_topk(ix, q_dev, nq, k, D_out, I_out, st);
  }
  *taken = nq;
  return r;
}

extern "C" int64_t knnx_i8_served(knnx_index* ix) {
  if (!ix) return -1;
  std::lock_guard<std::mutex> lk(ix->mu);
  return (int64_t)ix->i8_served;
}
extern "C" int64_t knnx_i8_rows(knnx_index* ix) {
  if (!ix) return -1;
  std::lock_guard<std::mutex> lk(ix->mu);
  return ix->i8_valid ? ix->i8_nrows : 0;
}
extern "C" int knnx_i8_planes(knnx_index* ix) {
  if (!ix) return -1;
  std::lock_guard<std::mutex> lk(ix->mu);
  return ix->i8_valid ? ix->i8_planes : 0;
}
extern "C" int knnx_i8_dominant(knnx_index* ix, int* cols4) {
  if (!ix) return -1;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!ix->i8_valid) return 0;
  for (int j = 0; j < ix->i8_dom.n && cols4; ++j) {
    int c = j;  // the column at position j
    for (int f = 0; f < ix->i8_dom.nfix; ++f)
      if (ix->i8_dom.pos[f] == j) c = ix->i8_dom.src[f];
    cols4[j] = c;
  }
  return ix->i8_dom.n;
}

extern "C" int knnx_search_device(knnx_index* ix, const float* q_dev, int n, int k, float* D_dev, int64_t* I_dev,
                                  void* stream) {
  if (!ix || !q_dev || !D_dev || !I_dev || n < 0 || k <= 0) return fail(KNNX_E_ARG, "bad search arguments");
  if (k > KNNX_MAX_K_FAST) return fail(KNNX_E_UNSUPPORTED, "device-buffer search supports k <= 64");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  hipStream_t st = stream ? (hipStream_t)stream : ix->stream;
  if (scratch_acquire(ix, st)) return KNNX_E_HIP;
  for (int o = 0; o < n;) {
    int nq = 0;
    int r = scan_step(ix, q_dev + (size_t)o * ix->d, n - o, k, D_dev + (size_t)o * k, I_dev + (size_t)o * k, st, &nq);
    if (r) return r;
    o += nq;
  }
  if (scratch_release(ix, st)) return KNNX_E_HIP;
  return KNNX_OK;
}

// k <= 64, host buffers; caller holds ix->mu
static int search_fast_locked(knnx_index* ix, const float* q, int n, int k, float* D, int64_t* I) {
  hipStream_t st = ix->stream;
  const size_t QM = KNN_RQ_MAX;  // the widest scan's query count: staging is sized for it
  const size_t qb = QM * ix->d * sizeof(float);
  const size_t db = QM * k * sizeof(float), ib = QM * k * sizeof(int64_t);
  int r = ensure_pin(ix, qb + db + ib + 64);
  if (r) return r;
  char* pin = (char*)ix->pin;
  if (scratch_acquire(ix, st)) return KNNX_E_HIP;
  for (int o = 0; o < n;) {
    const int rem = n - o;
    const int nq = step_queries(ix, rem, k);
    memcpy(pin, q + (size_t)o * ix->d, (size_t)nq * ix->d * sizeof(float));
    HIPCHK(hipMemcpyAsync(ix->q_dev, pin, (size_t)nq * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
    int took = 0;
    // the proof-based scans leave their fallback to this function (knnx_index::defer_fb): the gates come back with the results
    ix->defer_fb = true;
    ix->defer_kind = 0;
    r = scan_step(ix, ix->q_dev, nq, k, ix->D_dev, ix->I_dev, st, &took);
    ix->defer_fb = false;
    if (r) return r;
    // (the step may take fewer than were staged: the int8 path can turn itself off -- out of memory -- between the two decisions)
    if (took <= 0 || took > nq) return fail(KNNX_E_STATE, "internal: scan step size mismatch");
    unsigned* gates = reinterpret_cast<unsigned*>(pin + qb + db + ib);
    const int kind = ix->defer_kind, ngroup = kind ? (ix->defer_nq + KNN_NQ - 1) / KNN_NQ : 0;
    if (kind) HIPCHK(hipMemcpyAsync(gates, kind == 1 ? ix->rq_gate : ix->wide_gate, (size_t)ngroup * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pin + qb, ix->D_dev, (size_t)took * k * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pin + qb + db, ix->I_dev, (size_t)took * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    bool again = false;
    for (int g = 0; g < ngroup; ++g) {
      if (!gates[g]) continue;
      // a query of this group is unproven (rare: the bench counts them as fallbacks): its group's exact scan, then the rows again
      const int q0 = g * KNN_NQ, m = std::min(KNN_NQ, ix->defer_nq - q0);
      r = scan_topk(ix, ix->q_dev + (size_t)q0 * ix->d, m, k, ix->wide_Dfb, ix->wide_Ifb, st);
      if (r) return r;
      HIPCHK(launch_select(kind == 1 ? ix->rq_need : ix->wide_need, q0, m, k, ix->wide_Dfb, ix->wide_Ifb, ix->D_dev, ix->I_dev, st));
      again = true;
    }
    if (again) {
      HIPCHK(hipMemcpyAsync(pin + qb, ix->D_dev, (size_t)took * k * sizeof(float), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(pin + qb + db, ix->I_dev, (size_t)took * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    memcpy(D + (size_t)o * k, pin + qb, (size_t)took * k * sizeof(float));
    memcpy(I + (size_t)o * k, pin + qb + db, (size_t)took * k * sizeof(int64_t));
    o += took;
  }
  if (scratch_release(ix, st)) return KNNX_E_HIP;
  return 0;
}

static int search_large_k_locked(knnx_index* ix, const float* q, int n, int k, float* D, int64_t* I);

// ---------------------------------------------------------------------------------------------
// Request coalescer: single-query calls from many threads -> one scan per batch
// ---------------------------------------------------------------------------------------------
hipError_t knnx_launch_dedup_pairs(const float* rows, const int64_t* ids, int m, int k, int d, float thr, const unsigned char* want_or_null,
                                   int32_t* pairs, int cap, int* npairs, hipStream_t st);  // postfilter.hip

struct CoReq {
  const float* q;     // [d]
  int k;
  float* D;           // [k]
  int64_t* I;         // [k]
  float* R;           // [k, d] or null
  bool dedup;         // also report the request's duplicate links (knnx_search_dedup)
  float thr;
  int32_t* pairs;     // [2 * pairs_cap] (i, j), i < j
  int pairs_cap;
  int* n_pairs;
  int rc = 0;
  std::string err;
  bool done = false;
};

constexpr int CO_PAIR_CAP = 512;  // links kept per request on the device (a request with more falls back to the range scan)

// One batch: `b` holds m <= KNN_RQ_MAX requests with the same k.  Search (one scan), one gather of the m k result rows on the
// device when a request wants them back or wants its dedup links, the link kernel for all of them at once, then every request's
// slice into its own buffers.  Runs under ix->mu; the requests' threads sleep on ix->co_cv meanwhile.
static int co_run_batch_locked(knnx_index* ix, std::vector<CoReq*>& b);
//  The per-index staging vectors co_qbuf / co_Dbuf / co_Ibuf are shared by every caller: with coalescing off
// knnx_search_dedup reaches this function from concurrent request threads, so the index mutex is taken BEFORE they are touched;
// and nothing may leave this function as a C++ exception (a bad_alloc in a resize would otherwise skip the leader's hand-over in
// co_submit and strand every waiter).
static int co_run_batch(knnx_index* ix, std::vector<CoReq*>& b) {
  std::lock_guard<std::mutex> lk(ix->mu);
  try {
    return co_run_batch_locked(ix, b);
  } catch (const std::exception& e) {
    return fail(KNNX_E_NOMEM, std::string("coalesced batch: ") + e.what());
  }
}
static int co_run_batch_locked(knnx_index* ix, std::vector<CoReq*>& b) {
  const int m = (int)b.size(), k = b[0]->k, d = ix->d;
  ix->co_qbuf.resize((size_t)m * d);
  ix->co_Dbuf.resize((size_t)m * k);
  ix->co_Ibuf.resize((size_t)m * k);
  for (int i = 0; i < m; ++i) memcpy(ix->co_qbuf.data() + (size_t)i * d, b[i]->q, (size_t)d * sizeof(float));
  if (set_dev(ix)) return KNNX_E_HIP;
  int r = search_fast_locked(ix, ix->co_qbuf.data(), m, k, ix->co_Dbuf.data(), ix->co_Ibuf.data());
  if (r) return r;
  bool any_r = false, any_dd = false;
  for (CoReq* c : b) { any_r |= c->R != nullptr; any_dd |= c->dedup; }
  if (any_r || any_dd) {
    hipStream_t st = ix->stream;
    const int64_t nrow = (int64_t)m * k;
    int64_t* ids_dev = nullptr;
    float* rows_dev = nullptr;
    int32_t* pairs_dev = nullptr;
    int* np_dev = nullptr;
    unsigned char* want_dev = nullptr;
    if ((r = ensure_scratch(ix, 0, (size_t)nrow * sizeof(int64_t), (void**)&ids_dev))) return r;
    if ((r = ensure_scratch(ix, 1, (size_t)nrow * d * sizeof(float), (void**)&rows_dev))) return r;
    size_t pin_need = (size_t)nrow * sizeof(int64_t);
    if (any_dd) pin_need = std::max(pin_need, (size_t)m * ((size_t)CO_PAIR_CAP * 4 + 8));
    if (any_r) pin_need = std::max(pin_need, std::min((size_t)nrow * d * sizeof(float), (size_t)32 << 20));
    if ((r = ensure_pin(ix, pin_need))) return r;
    memcpy(ix->pin, ix->co_Ibuf.data(), (size_t)nrow * sizeof(int64_t));
    HIPCHK(hipMemcpyAsync(ids_dev, ix->pin, (size_t)nrow * sizeof(int64_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // the pinned buffer is reused below
    hipError_t e = ix->ivf_nlist ? launch_gather_inv(ix->rows, d, ix->id_base, ix->ntotal, ix->ivf_inv, ids_dev, nrow, rows_dev, st)
                                 : launch_gather(ix->rows, ix->ntotal, d, ix->id_base, ids_dev, nrow, rows_dev, st);
    if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("coalesced gather: ") + hipGetErrorString(e));
    if (any_dd) {
      if ((r = ensure_scratch(ix, 6, (size_t)m * CO_PAIR_CAP * sizeof(int32_t), (void**)&pairs_dev))) return r;
      if ((r = ensure_scratch(ix, 7, (size_t)m * sizeof(int), (void**)&np_dev))) return r;
      if ((r = ensure_scratch(ix, 8, (size_t)m, (void**)&want_dev))) return r;
      unsigned char* want = (unsigned char*)ix->pin;
      for (int i = 0; i < m; ++i) want[i] = b[i]->dedup ? 1 : 0;
      HIPCHK(hipMemcpyAsync(want_dev, want, (size_t)m, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemsetAsync(np_dev, 0, (size_t)m * sizeof(int), st));
      HIPCHK(hipStreamSynchronize(st));
      // one threshold per batch: requests with another threshold run in their own launch (a service has one: 0.94)
      std::vector<float> thrs;
      for (CoReq* c : b)
        if (c->dedup && std::find(thrs.begin(), thrs.end(), c->thr) == thrs.end()) thrs.push_back(c->thr);
      for (float t : thrs) {
        for (int i = 0; i < m; ++i) want[i] = (b[i]->dedup && b[i]->thr == t) ? 1 : 0;
        if (thrs.size() > 1) {
          HIPCHK(hipMemcpyAsync(want_dev, want, (size_t)m, hipMemcpyHostToDevice, st));
          HIPCHK(hipStreamSynchronize(st));
        }
        e = knnx_launch_dedup_pairs(rows_dev, ids_dev, m, k, d, t, want_dev, pairs_dev, CO_PAIR_CAP, np_dev, st);
        if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("dedup links: ") + hipGetErrorString(e));
      }
      int* np_h = (int*)ix->pin;
      int32_t* pairs_h = (int32_t*)((char*)ix->pin + (size_t)m * 8);
      HIPCHK(hipMemcpyAsync(np_h, np_dev, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(pairs_h, pairs_dev, (size_t)m * CO_PAIR_CAP * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      for (int i = 0; i < m; ++i) {
        CoReq* c = b[i];
        if (!c->dedup) continue;
        const int n = np_h[i];
        *c->n_pairs = n;  // may exceed pairs_cap / CO_PAIR_CAP: the caller sees that and takes the general path
        const int keep = std::min(n, std::min(c->pairs_cap, CO_PAIR_CAP));
        // the device appends in completion order: sort so that the answer does not depend on scheduling
        std::vector<int32_t> ps(pairs_h + (size_t)i * CO_PAIR_CAP, pairs_h + (size_t)i * CO_PAIR_CAP + std::min(n, CO_PAIR_CAP));
        std::sort(ps.begin(), ps.end());
        for (int j = 0; j < keep; ++j) { c->pairs[2 * j] = ps[j] >> 16; c->pairs[2 * j + 1] = ps[j] & 0xffff; }
      }
    }
    if (any_r) {
      const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(ix->pin_bytes / ((size_t)d * sizeof(float))));
      for (int i = 0; i < m; ++i) {
        if (!b[i]->R) continue;
        for (int64_t o = 0; o < k; o += chunk_rows) {
          const int64_t n = std::min<int64_t>(chunk_rows, k - o);
          HIPCHK(hipMemcpyAsync(ix->pin, rows_dev + ((size_t)i * k + o) * d, (size_t)n * d * sizeof(float), hipMemcpyDeviceToHost, st));
          HIPCHK(hipStreamSynchronize(st));
          memcpy(b[i]->R + (size_t)o * d, ix->pin, (size_t)n * d * sizeof(float));
        }
      }
    }
  }
  for (int i = 0; i < m; ++i) {
    memcpy(b[i]->D, ix->co_Dbuf.data() + (size_t)i * k, (size_t)k * sizeof(float));
    memcpy(b[i]->I, ix->co_Ibuf.data() + (size_t)i * k, (size_t)k * sizeof(int64_t));
  }
  return KNNX_OK;
}

// Called by every single-query request.  The first thread to find no leader leads: it serves batches (its own request is in
// the first one) until its request is done, then gives the lead away -- a waiting thread picks it up -- so no caller works for
// the others longer than one batch.
static int co_submit(knnx_index* ix, CoReq& me) {
  std::unique_lock<std::mutex> lk(ix->co_mu);
  ix->co_q.push_back(&me);
  for (;;) {
    if (me.done) break;
    if (ix->co_leader) {
      ix->co_cv.wait(lk);
      continue;
    }
    ix->co_leader = true;
    std::vector<CoReq*> batch;
    const int k0 = ix->co_q.front()->k;
    for (auto it = ix->co_q.begin(); it != ix->co_q.end() && (int)batch.size() < KNN_RQ_MAX;) {
      if ((*it)->k == k0) { batch.push_back(*it); it = ix->co_q.erase(it); } else ++it;
    }
    lk.unlock();
    const int r = co_run_batch(ix, batch);
    const std::string msg = r ? std::string(g_err) : std::string();
    lk.lock();
    ix->co_batches++;
    ix->co_queries += (int64_t)batch.size();
    ix->co_largest = std::max<int64_t>(ix->co_largest, (int64_t)batch.size());
    for (CoReq* c : batch) { c->rc = r; c->err = msg; c->done = true; }
    ix->co_leader = false;
    ix->co_cv.notify_all();
  }
  lk.unlock();
  if (me.rc) g_err = me.err;  // the message is thread-local: hand the leader's to this thread
  return me.rc;
}

extern "C" int knnx_set_coalesce(knnx_index* ix, int on) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->co_mu);
  ix->coalesce = on != 0;
  return KNNX_OK;
}

extern "C" int knnx_coalesce_stats(knnx_index* ix, int64_t* batches, int64_t* queries, int64_t* largest_batch) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->co_mu);
  if (batches) *batches = ix->co_batches;
  if (queries) *queries = ix->co_queries;
  if (largest_batch) *largest_batch = ix->co_largest;
  return KNNX_OK;
}

extern "C" int knnx_search_dedup(knnx_index* ix, const float* q, int k, float* D, int64_t* I, float* R_or_null, float dedup_thr,
                                 int32_t* pairs, int pairs_cap, int* n_pairs) {
  if (!ix || !q || !D || !I || k <= 0 || !pairs || pairs_cap < 0 || !n_pairs) return fail(KNNX_E_ARG, "bad search_dedup arguments");
  if (k > KNNX_MAX_K_FAST) return fail(KNNX_E_UNSUPPORTED, "search_dedup serves k <= 64 (larger answers: knnx_search + knnx_range_search_once)");
  CoReq me{q, k, D, I, R_or_null, true, dedup_thr, pairs, pairs_cap, n_pairs};
  *n_pairs = 0;
  if (ix->coalesce) return co_submit(ix, me);
  std::vector<CoReq*> one{&me};
  return co_run_batch(ix, one);
}

extern "C" int knnx_search(knnx_index* ix, const float* q, int n, int k, float* D, int64_t* I, float* R) {
  if (!ix || (n > 0 && (!q || !D || !I)) || n < 0 || k <= 0) return fail(KNNX_E_ARG, "bad search arguments");
  if (k > KNNX_MAX_K) return fail(KNNX_E_UNSUPPORTED, "k > 131072 is not implemented");
  if (n == 0) return KNNX_OK;
  if (n == 1 && k <= KNNX_MAX_K_FAST && ix->coalesce) {  // concurrent single-query callers share one scan
    CoReq me{q, k, D, I, R, false, 0.f, nullptr, 0, nullptr};
    return co_submit(ix, me);
  }
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (set_dev(ix)) return KNNX_E_HIP;
    int r = (k <= KNNX_MAX_K_FAST) ? search_fast_locked(ix, q, n, k, D, I) : search_large_k_locked(ix, q, n, k, D, I);
    if (r) return r;
  }
  if (R) return knnx_reconstruct(ix, I, (int64_t)n * k, R);
  return KNNX_OK;
}

extern "C" int knnx_reconstruct(knnx_index* ix, const int64_t* ids, int64_t n, float* out) {
  if (!ix || (n > 0 && (!ids || !out)) || n < 0) return fail(KNNX_E_ARG, "bad reconstruct arguments");
  if (n == 0) return KNNX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(32u << 20) / (ix->d * 4)));
  int64_t* ids_dev = nullptr;
  float* out_dev = nullptr;
  int r = ensure_scratch(ix, 0, (size_t)chunk * sizeof(int64_t), (void**)&ids_dev);
  if (r) return r;
  if ((r = ensure_scratch(ix, 1, (size_t)chunk * ix->d * sizeof(float), (void**)&out_dev))) return r;
  // pageable host memory: stage ids and rows through the pinned buffer so that the copies are truly asynchronous DMAs
  if ((r = ensure_pin(ix, (size_t)chunk * (sizeof(int64_t) + (size_t)ix->d * sizeof(float))))) return r;
  int64_t* ids_pin = (int64_t*)ix->pin;
  float* out_pin = (float*)((char*)ix->pin + (size_t)chunk * sizeof(int64_t));
  hipError_t e = hipSuccess;
  for (int64_t o = 0; o < n && e == hipSuccess; o += chunk) {
    const int64_t m = std::min(chunk, n - o);
    memcpy(ids_pin, ids + o, (size_t)m * sizeof(int64_t));
    e = hipMemcpyAsync(ids_dev, ids_pin, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess)
      e = ix->ivf_nlist ? launch_gather_inv(ix->rows, ix->d, ix->id_base, ix->ntotal, ix->ivf_inv, ids_dev, m, out_dev, ix->stream)
                        : launch_gather(ix->rows, ix->ntotal, ix->d, ix->id_base, ids_dev, m, out_dev, ix->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(out_pin, out_dev, (size_t)m * ix->d * sizeof(float), hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e == hipSuccess) memcpy(out + (size_t)o * ix->d, out_pin, (size_t)m * ix->d * sizeof(float));
  }
  if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("reconstruct: ") + hipGetErrorString(e));
  return KNNX_OK;
}

// range scan of <= KNN_NQ queries; leaves per-query counts in `counts` and the hits on the device
// (query i's hits at range_s/range_i[i*cap .. i*cap+counts[i]), cap = range_pool / nq)
static const size_t RANGE_POOL_MAX = (size_t)1 << 29;  // 512 Mi hits = 4 GiB of scratch
constexpr int64_t RANGE_BATCH_MAX_ROWS = 1 << 16;  // range_scan_batched: indexes up to this many rows (a scan is launch-bound there)

static int range_scan(knnx_index* ix, const float* q_host, int nq, float thr, std::vector<unsigned>& counts,
                      unsigned* cap_out) {
  hipStream_t st = ix->stream;
  int r = ensure_pin(ix, (size_t)KNN_NQ * ix->d * sizeof(float));
  if (r) return r;
  if (scratch_acquire(ix, st)) return KNNX_E_HIP;
  memcpy(ix->pin, q_host, (size_t)nq * ix->d * sizeof(float));
  HIPCHK(hipMemcpyAsync(ix->q_dev, ix->pin, (size_t)nq * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
  for (;;) {
    if (ix->range_pool == 0) {
      ix->range_pool = (size_t)1 << 21;
      HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_s, ix->range_pool * sizeof(float)));
      HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_i, ix->range_pool * sizeof(uint32_t)));
    }
    const unsigned cap = (unsigned)std::min<size_t>(ix->range_pool / (size_t)nq, 0xffffffffu);
    if (ix->ivf_nlist) {  // IVF: the range is taken over the rows of the nprobe best lists of each query (faiss IndexIVF semantics)
      r = ivf_build_worklist(ix, ix->q_dev, nq, st, nullptr);
      if (r) return r;
    }
    HIPCHK(launch_prep(ix->q_dev, nq, ix->d, ix->qfrag, ix->thr_g, ix->range_cnt, 0, nullptr, st));
    ScanArgs a{};
    a.X = ix->rows;
    a.N = ix->ivf_nlist ? ix->capacity : ix->ntotal;
    a.work = ix->ivf_nlist ? ix->ivf_work : nullptr;
    a.nwork = ix->ivf_nwork;
    a.d = ix->d;
    a.qfrag = ix->qfrag;
    a.nq = nq;
    a.k = 1;
    a.cap = 2;
    a.grid = ix->n_cu;
    a.mode = 1;
    a.thr_g = ix->thr_g;
    a.range_thr = thr;
    a.range_cnt = ix->range_cnt;
    a.range_cap = cap;
    a.range_s = ix->range_s;
    a.range_i = ix->range_i;
    HIPCHK(launch_scan(a, st));
    counts.assign(KNN_NQ, 0u);
    HIPCHK(hipMemcpyAsync(counts.data(), ix->range_cnt, KNN_NQ * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unsigned mx = 0;
    for (int i = 0; i < nq; ++i) mx = std::max(mx, counts[i]);
    if (mx <= cap) {
      *cap_out = cap;
      return 0;
    }
    // overflow: regrow and rescan (the counts are exact even when the buffers overflowed)
    size_t want = ix->range_pool;
    while (want / (size_t)nq < (size_t)mx) want <<= 1;
    if (want > RANGE_POOL_MAX) return fail(KNNX_E_NOMEM, "range_search result exceeds the 512 Mi-hit scratch limit");
    hipFree(ix->range_s);
    hipFree(ix->range_i);
    ix->range_s = nullptr;
    ix->range_i = nullptr;
    ix->range_pool = 0;
    HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_s, want * sizeof(float)));
    HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_i, want * sizeof(uint32_t)));
    ix->range_pool = want;
  }
}

// Many queries against a SMALL flat index (the per-request dedup of clip_back.py:290-294: the k result vectors against themselves,
// k up to 3 000): all 32-query groups are launched back to back, every group with its own counters and its own slice of the hit
// pool, and the host synchronises ONCE for the counts -- the one-group-at-a-time loop costs two synchronisations per group
// (94 groups at k = 3 000: 10 ms of a 45 ms request, profiles/r03h_request.log).  counts [n]; *cnt_dev_out = the device counters
// (scratch slot 4) for range_fetch.
static int range_scan_batched(knnx_index* ix, const float* q_host, int n, float thr, std::vector<unsigned>& counts, unsigned* cap_out,
                              unsigned** cnt_dev_out) {
  hipStream_t st = ix->stream;
  const int ngroups = (n + KNN_NQ - 1) / KNN_NQ;
  const size_t nr = (size_t)ngroups * KNN_NQ;
  float* q_all = nullptr;
  unsigned* cnt_all = nullptr;
  int r = ensure_scratch(ix, 3, nr * ix->d * sizeof(float), (void**)&q_all);
  if (!r) r = ensure_scratch(ix, 4, nr * sizeof(unsigned), (void**)&cnt_all);
  if (r) return r;
  if (scratch_acquire(ix, st)) return KNNX_E_HIP;
  HIPCHK(hipMemcpyAsync(q_all, q_host, (size_t)n * ix->d * sizeof(float), hipMemcpyHostToDevice, st));
  for (;;) {
    if (ix->range_pool == 0) {
      ix->range_pool = (size_t)1 << 21;
      HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_s, ix->range_pool * sizeof(float)));
      HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_i, ix->range_pool * sizeof(uint32_t)));
    }
    const unsigned cap = (unsigned)std::min<size_t>(ix->range_pool / nr, 0xffffffffu);
    unsigned mx = 0;
    if (cap > 0) {
      for (int g = 0; g < ngroups; ++g) {
        const int q0 = g * KNN_NQ, nq = std::min(KNN_NQ, n - q0);
        HIPCHK(launch_prep(q_all + (size_t)q0 * ix->d, nq, ix->d, ix->qfrag, ix->thr_g, cnt_all + q0, 0, nullptr, st));
        ScanArgs a{};
        a.X = ix->rows;
        a.N = ix->ntotal;
        a.d = ix->d;
        a.qfrag = ix->qfrag;
        a.nq = nq;
        a.k = 1;
        a.cap = 2;
        a.grid = ix->n_cu;
        a.mode = 1;
        a.thr_g = ix->thr_g;
        a.range_thr = thr;
        a.range_cnt = cnt_all + q0;
        a.range_cap = cap;
        a.range_s = ix->range_s + (size_t)q0 * cap;
        a.range_i = ix->range_i + (size_t)q0 * cap;
        HIPCHK(launch_scan(a, st));
      }
      counts.assign(nr, 0u);
      HIPCHK(hipMemcpyAsync(counts.data(), cnt_all, nr * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      for (int i = 0; i < n; ++i) mx = std::max(mx, counts[i]);
      if (mx <= cap) {
        *cap_out = cap;
        *cnt_dev_out = cnt_all;
        return 0;
      }
    } else {
      mx = 1;
    }
    // overflow: regrow and rescan (the counts are exact even when the buffers overflowed)
    size_t want = ix->range_pool;
    while (want / nr < (size_t)mx) want <<= 1;
    if (want > RANGE_POOL_MAX) return fail(KNNX_E_NOMEM, "range_search result exceeds the 512 Mi-hit scratch limit");
    hipFree(ix->range_s);
    hipFree(ix->range_i);
    ix->range_s = nullptr;
    ix->range_i = nullptr;
    ix->range_pool = 0;
    HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_s, want * sizeof(float)));
    HIPCHK(malloc_or_reclaim(ix, (void**)&ix->range_i, want * sizeof(uint32_t)));
    ix->range_pool = want;
  }
}

// copy the hits of the last range_scan to the host, ids ascending inside each query
static int range_fetch(knnx_index* ix, int nq, const std::vector<unsigned>& counts, unsigned cap, float* D,
                       int64_t* I, const unsigned* cnt_dev = nullptr) {
  if (!cnt_dev) cnt_dev = ix->range_cnt;
  std::vector<int64_t> loc(nq + 1);
  loc[0] = 0;
  for (int i = 0; i < nq; ++i) loc[i + 1] = loc[i] + counts[i];
  if (loc[nq] == 0) return 0;
  int64_t* lims_dev = nullptr;
  float* D_dev = nullptr;
  int64_t* I_dev = nullptr;
  int rr = ensure_scratch(ix, 0, (nq + 1) * sizeof(int64_t), (void**)&lims_dev);
  if (!rr) rr = ensure_scratch(ix, 1, (size_t)loc[nq] * sizeof(float), (void**)&D_dev);
  if (!rr) rr = ensure_scratch(ix, 2, (size_t)loc[nq] * sizeof(int64_t), (void**)&I_dev);
  if (rr) return rr;
  hipError_t e = hipMemcpyAsync(lims_dev, loc.data(), (nq + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream);
  if (e == hipSuccess)
    e = launch_range_sort(ix->range_s, ix->range_i, cnt_dev, cap, lims_dev, ix->id_base,
                          ix->ivf_nlist ? ix->ivf_idmap : nullptr, nq, D_dev, I_dev, ix->stream);
  // hit lists too long for rank-by-counting: radix sort by id, one query at a time
  unsigned longest = 0;
  for (int i = 0; i < nq; ++i) longest = std::max(longest, counts[i] > RANGE_SORT_SMALL ? counts[i] : 0u);
  if (e == hipSuccess && longest) {
    uint32_t *k0 = nullptr, *k1 = nullptr;
    float *v0 = nullptr, *v1 = nullptr;
    unsigned* hist = nullptr;
    e = hipMalloc(&k0, (size_t)longest * 4);
    if (e == hipSuccess) e = hipMalloc(&k1, (size_t)longest * 4);
    if (e == hipSuccess) e = hipMalloc(&v0, (size_t)longest * 4);
    if (e == hipSuccess) e = hipMalloc(&v1, (size_t)longest * 4);
    if (e == hipSuccess) e = hipMalloc(&hist, (size_t)16 * ((longest + 2047) / 2048) * sizeof(unsigned));
    int key_bits = 1;
    while (key_bits < 32 && ((int64_t)1 << key_bits) < std::max<int64_t>(ix->ntotal, 2)) ++key_bits;
    for (int i = 0; i < nq && e == hipSuccess; ++i)
      if (counts[i] > RANGE_SORT_SMALL)
        e = launch_range_sort_long(ix->range_s + (size_t)i * cap, ix->range_i + (size_t)i * cap, counts[i], ix->id_base,
                                   ix->ivf_nlist ? ix->ivf_idmap : nullptr, key_bits, k0, k1, v0, v1, hist, D_dev + loc[i],
                                   I_dev + loc[i], ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(hist);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(D, D_dev, (size_t)loc[nq] * sizeof(float), hipMemcpyDeviceToHost, ix->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(I, I_dev, (size_t)loc[nq] * sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
  if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("range fetch: ") + hipGetErrorString(e));
  return 0;
}

extern "C" int knnx_range_search(knnx_index* ix, const float* q, int n, float thresh, int64_t* lims, float* D,
                                 int64_t* I) {
  if (!ix || !lims || (n > 0 && !q) || n < 0) return fail(KNNX_E_ARG, "bad range_search arguments");
  if ((D == nullptr) != (I == nullptr)) return fail(KNNX_E_ARG, "D and I must both be null or both be set");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  const bool fill = D != nullptr;
  int64_t run = 0;
  std::vector<unsigned> counts;
  if (!fill) lims[0] = 0;
  for (int o = 0; o < n; o += KNN_NQ) {
    const int nq = std::min(KNN_NQ, n - o);
    unsigned cap = 0;
    int r = range_scan(ix, q + (size_t)o * ix->d, nq, thresh, counts, &cap);
    if (r) return r;
    if (!fill) {
      for (int i = 0; i < nq; ++i) {
        run += counts[i];
        lims[o + i + 1] = run;
      }
      continue;
    }
    // filling: the caller passes the lims of the counting call; verify them as we go
    for (int i = 0; i < nq; ++i)
      if (lims[o + i + 1] - lims[o + i] != (int64_t)counts[i])
        return fail(KNNX_E_STATE, "lims do not match this query/threshold (index changed between the two calls?)");
    r = range_fetch(ix, nq, counts, cap, D + lims[o], I + lims[o]);
    if (r) return r;
  }
  return KNNX_OK;
}

// range_search in ONE pass when the caller's buffers are large enough: lims [n + 1] is always filled; D / I (capacity entries)
// receive the hits if lims[n] <= capacity (return KNNX_OK), otherwise nothing is written to them and the return value is 1 --
// the caller then allocates lims[n] entries and uses knnx_range_search's filling call.  Saves the counting scan of the two-call
// protocol for callers with a good guess (the per-request dedup of clip_back.py:290-294: ~k hits for k result vectors).
extern "C" int knnx_range_search_once(knnx_index* ix, const float* q, int n, float thresh, int64_t* lims, float* D, int64_t* I,
                                      int64_t capacity) {
  if (!ix || !lims || (n > 0 && !q) || n < 0 || capacity < 0 || (capacity > 0 && (!D || !I))) return fail(KNNX_E_ARG, "bad range_search_once arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  int64_t run = 0;
  bool fits = true;
  std::vector<unsigned> counts;
  lims[0] = 0;
  if (!ix->ivf_nlist && n > KNN_NQ && ix->ntotal <= RANGE_BATCH_MAX_ROWS) {  // small flat index, many queries: one synchronisation
    unsigned cap = 0;
    unsigned* cnt_dev = nullptr;
    int r = range_scan_batched(ix, q, n, thresh, counts, &cap, &cnt_dev);
    if (r) return r;
    for (int i = 0; i < n; ++i) {
      run += counts[i];
      lims[i + 1] = run;
    }
    if (run > capacity) return 1;
    counts.resize(n);
    return range_fetch(ix, n, counts, cap, D, I, cnt_dev);
  }
  for (int o = 0; o < n; o += KNN_NQ) {
    const int nq = std::min(KNN_NQ, n - o);
    unsigned cap = 0;
    int r = range_scan(ix, q + (size_t)o * ix->d, nq, thresh, counts, &cap);
    if (r) return r;
    const int64_t first = run;
    for (int i = 0; i < nq; ++i) {
      run += counts[i];
      lims[o + i + 1] = run;
    }
    if (run > capacity) fits = false;
    if (fits) {
      r = range_fetch(ix, nq, counts, cap, D + first, I + first);
      if (r) return r;
    }
  }
  return fits ? KNNX_OK : 1;
}

// k > 64 (the front-end's num_result_ids = 3000, clip_back.py:358 asks for up to 1e5): one range scan with a threshold that
// lets >= k rows through, the hits ranked on the host (k log k, tiny next to a scan).  Exact: every row scoring above the
// threshold is a hit, so cnt >= k hits contain the top k.  The threshold:
//   * flat index with N >= 128 k rows: the j-th best EXACT score of a strided sample (every S-th 32-row tile, the 32-query scan
//     with tstride = S: 1 / S of the bytes), S and j <= 48 chosen so that j S ~ 1.5 k rows are expected above it.  The count
//     above the j-th best of a 1 / S sample is ~ S Gamma(j) whatever the score distribution (the argument of the RQ scan's
//     thresholds, knn_kernels.h), so cnt < k has a probability of a few per cent; then the 64-th sample score is tried
//     (~ S * 64 hits) before the descent below takes over.  Cost: ~1.01 scans instead of the 2 - 6 of round 2 -- and instead of
//     the 10.9 s measured at 100 M rows in round 3 (profiles/r03b_request.log), where a query with one very close neighbour
//     (top - s64 = 0.8) made the first descent step jump below zero and 50 M rows had to be fetched and sorted;
//   * otherwise (small or IVF index): descent from the top-64 scores.  The first step extrapolates the LOCAL score density
//     (32 rows lie between the 32nd and the 64th score), doubling while cnt < k; when a step overshoots by more than 16 x the
//     threshold is bisected back (at most 4 times) before the hits are fetched.
// Caller holds ix->mu.
static int sample_scores_locked(knnx_index* ix, const float* qq, int tstride, float* d64) {
  hipStream_t st = ix->stream;
  int r = ensure_pin(ix, (size_t)ix->d * sizeof(float) + KNNX_MAX_K_FAST * sizeof(float));
  if (r) return r;
  if (scratch_acquire(ix, st)) return KNNX_E_HIP;
  memcpy(ix->pin, qq, (size_t)ix->d * sizeof(float));
  HIPCHK(hipMemcpyAsync(ix->q_dev, ix->pin, (size_t)ix->d * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(launch_prep(ix->q_dev, 1, ix->d, ix->qfrag, ix->thr_g, nullptr, 0, nullptr, st));
  ScanArgs a{};
  a.X = ix->rows;
  a.N = ix->ntotal;
  a.d = ix->d;
  a.qfrag = ix->qfrag;
  a.nq = 1;
  a.k = KNNX_MAX_K_FAST;
  a.cap = scan_cap(ix->d, KNNX_MAX_K_FAST);
  a.grid = ix->n_cu;
  a.mode = 0;
  a.tstride = tstride;
  a.thr_g = ix->thr_g;
  a.part_s = ix->part_s;
  a.part_i = ix->part_i;
  a.part_n = ix->part_n;
  if (a.cap < 0) return fail(KNNX_E_UNSUPPORTED, "k too large for the LDS queues at this d");
  HIPCHK(launch_scan(a, st));
  HIPCHK(launch_merge_u32(ix->part_s, ix->part_i, ix->part_n, ix->n_cu, KNN_NQ, KNNX_MAX_K_FAST, 1, KNNX_MAX_K_FAST, 0, nullptr, ix->D_dev,
                          ix->I_dev, nullptr, st));
  float* out = (float*)((char*)ix->pin + (size_t)ix->d * sizeof(float));
  HIPCHK(hipMemcpyAsync(out, ix->D_dev, KNNX_MAX_K_FAST * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  memcpy(d64, out, KNNX_MAX_K_FAST * sizeof(float));
  if (scratch_release(ix, st)) return KNNX_E_HIP;
  return 0;
}

static int search_large_k_locked(knnx_index* ix, const float* q, int n, int k, float* D, int64_t* I) {
  std::vector<float> d64((size_t)KNNX_MAX_K_FAST);
  std::vector<int64_t> i64((size_t)KNNX_MAX_K_FAST);
  const int64_t N = ix->ntotal;
  const int64_t want = std::min<int64_t>(k, N);
  for (int qi = 0; qi < n; ++qi) {
    const float* qq = q + (size_t)qi * ix->d;
    float* Dq = D + (size_t)qi * k;
    int64_t* Iq = I + (size_t)qi * k;
    std::vector<std::pair<float, int64_t>> hits;
    std::vector<unsigned> counts;
    unsigned cap = 0;
    int r;
    bool done = false;
    auto fetch = [&](int64_t cnt) -> int {
      std::vector<float> hd((size_t)cnt);
      std::vector<int64_t> hi((size_t)cnt);
      int rr = range_fetch(ix, 1, counts, cap, hd.data(), hi.data());
      if (rr) return rr;
      hits.reserve((size_t)cnt);
      for (int64_t j = 0; j < cnt; ++j) hits.emplace_back(hd[j], hi[j]);
      return 0;
    };
    float thr_hi = FLT_MAX;  // a threshold known to let fewer than k rows through (descent / bisection bracket)
    if (!ix->ivf_nlist && N >= (int64_t)128 * k) {
      // ---- sampled threshold: S tiles apart, j-th best sample score, j S ~ 1.5 k
      const int S = (int)std::max<int64_t>(2, (3 * (int64_t)k / 2 + 47) / 48);
      const int j = (int)std::min<int64_t>(48, std::max<int64_t>(8, (3 * (int64_t)k / 2 + S - 1) / S));
      r = sample_scores_locked(ix, qq, S, d64.data());
      if (r) return r;
      for (int attempt = 0; attempt < 2 && !done; ++attempt) {
        const float sj = d64[attempt == 0 ? j - 1 : KNNX_MAX_K_FAST - 1];
        if (!(sj > -FLT_MAX)) break;  // fewer sample rows than that
        const float thr = nextafterf(sj, -FLT_MAX);  // the range is strict (>): keep the sample row itself
        r = range_scan(ix, qq, 1, thr, counts, &cap);
        if (r) return r;
        if ((int64_t)counts[0] >= want) {
          if ((r = fetch(counts[0]))) return r;
          done = true;
        } else {
          thr_hi = thr;
        }
      }
    }
    if (!done) {
      r = search_fast_locked(ix, qq, 1, KNNX_MAX_K_FAST, d64.data(), i64.data());
      if (r) return r;
      int have = 0;
      while (have < KNNX_MAX_K_FAST && i64[have] >= 0) ++have;
      if (have < KNNX_MAX_K_FAST) {  // the whole index (or all probed lists) is smaller than 64 rows
        for (int j = 0; j < have; ++j) hits.emplace_back(d64[j], i64[j]);
      } else {
        const float s32 = d64[KNNX_MAX_K_FAST / 2 - 1], s64 = d64[KNNX_MAX_K_FAST - 1];
        // local density: 32 rows within s32 - s64; k - 64 more rows are ~ (k - 64) / 32 such gaps away (fewer: the density grows)
        float step = std::max((s32 - s64) * std::min(64.f, (float)(k - KNNX_MAX_K_FAST) / 32.f) * 0.5f, 1e-4f * std::max(1.f, fabsf(s64)));
        float thr = std::min(s64 - step, thr_hi < FLT_MAX ? nextafterf(thr_hi, -FLT_MAX) : FLT_MAX);
        float lo_ok = -FLT_MAX;  // lowest threshold tried so far (lets >= k rows through once found)
        int64_t prev_cnt = -1;
        int stalled = 0, bisections = 0;
        thr_hi = std::min(thr_hi, s64);
        for (int it = 0;; ++it) {
          r = range_scan(ix, qq, 1, thr, counts, &cap);
          if (r) return r;
          const int64_t cnt = counts[0];
          // an IVF index only reaches the rows of the probed lists: when three ever larger steps add nothing, take what there
          // is.  Flat indexes never take this shortcut: cnt >= min(k, N) is reached after O(log) doublings.
          stalled = (ix->ivf_nlist && cnt == prev_cnt) ? stalled + 1 : 0;
          prev_cnt = cnt;
          if (cnt < want && thr > -FLT_MAX && stalled < 3 && it < 60) {
            thr_hi = thr;
            step *= 2.f;
            thr = (thr - step > -FLT_MAX) ? thr - step : -FLT_MAX;
            continue;
          }
          if (cnt < want && thr > -FLT_MAX) {  // stalled (IVF) or out of patience: everything the scan can reach
            thr = -FLT_MAX;
            stalled = 0;
            continue;
          }
          // enough rows.  Far too many (an overshooting step)?  Bisect back towards the last threshold that had too few.
          if (cnt > 16 * want && cnt > 65536 && bisections < 4 && thr_hi < FLT_MAX && thr > -FLT_MAX) {
            lo_ok = thr;
            thr = 0.5f * (thr + thr_hi);
            ++bisections;
            // (a middle that has too few rows raises thr_hi above and steps down again, towards lo_ok)
            step = 0.5f * (thr - lo_ok);
            continue;
          }
          if ((r = fetch(cnt))) return r;
          break;
        }
      }
    }
    // rank the hits: score descending, ties by ascending id.  Only the first k are reported, so only they are sorted (a range
    // scan hands back 1.5 .. 16 x k hits: O(hits) selection + k log k, 10 ms at k = 100 000 instead of a full sort)
    auto better = [](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
      return a.first > b.first || (a.first == b.first && a.second < b.second);
    };
    if ((int64_t)hits.size() > (int64_t)k) {
      std::nth_element(hits.begin(), hits.begin() + k, hits.end(), better);
      hits.resize((size_t)k);
    }
    std::sort(hits.begin(), hits.end(), better);
    for (int j = 0; j < k; ++j) {
      if (j < (int)hits.size()) {
        Dq[j] = hits[j].first;
        Iq[j] = hits[j].second;
      } else {
        Dq[j] = -FLT_MAX;
        Iq[j] = -1;
      }
    }
  }
  return KNNX_OK;
}

// ---------------------------------------------------------------------------------------------
// IVF-Flat
// ---------------------------------------------------------------------------------------------
extern "C" int knnx_ivf_set_lists(knnx_index* ix, int nlist, const uint16_t* centroids_f16, const int64_t* list_sizes,
                                  const int64_t* ids) {
  if (!ix || nlist <= 0 || !centroids_f16 || !list_sizes || !ids) return fail(KNNX_E_ARG, "bad ivf_set_lists arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (ix->borrowed) return fail(KNNX_E_STATE, "IVF needs an index that owns its rows");
  if (ix->ivf_nlist) return fail(KNNX_E_STATE, "lists are already set");
  std::vector<int64_t> src0(nlist);
  std::vector<unsigned> tile0(nlist), ntile(nlist), size(nlist);
  int64_t run = 0, tiles = 0;
  for (int l = 0; l < nlist; ++l) {
    if (list_sizes[l] < 0) return fail(KNNX_E_ARG, "negative list size");
    src0[l] = run;
    size[l] = (unsigned)list_sizes[l];
    ntile[l] = (unsigned)((list_sizes[l] + 31) / 32);
    tile0[l] = (unsigned)tiles;
    run += list_sizes[l];
    tiles += ntile[l];
  }
  if (run != ix->ntotal) return fail(KNNX_E_ARG, "list sizes do not add up to ntotal (add the rows list by list first)");
  if (tiles * 32 > (int64_t)0xffffffffll) return fail(KNNX_E_UNSUPPORTED, "more than 2^32 padded rows per device");
  for (int64_t i = 0; i < run; ++i)
    if (ids[i] < ix->id_base || ids[i] - ix->id_base >= run) return fail(KNNX_E_ARG, "ids must be a permutation of [id_base, id_base + ntotal)");
  const int64_t prow = std::max<int64_t>(tiles * 32, 32);
  _Float16* dst = nullptr;
  int64_t* ids_dev = nullptr;
  int64_t* src0_dev = nullptr;
  HIPCHK(hipMalloc(&dst, (size_t)prow * ix->d * sizeof(_Float16)));
  hipError_t e = hipMalloc(&ids_dev, (size_t)std::max<int64_t>(run, 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&src0_dev, nlist * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_tile0, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_ntile, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_size, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_masks, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_off, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_nwork, sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_work, (size_t)std::max<int64_t>(tiles, 1) * sizeof(uint4));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_idmap, (size_t)prow * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_inv, (size_t)std::max<int64_t>(run, 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_Ic, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_Dc, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(ids_dev, ids, (size_t)run * sizeof(int64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(src0_dev, src0.data(), nlist * sizeof(int64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->ivf_tile0, tile0.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->ivf_ntile, ntile.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->ivf_size, size.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = launch_ivf_relayout(ix->rows, dst, ix->d, nlist, src0_dev, ix->ivf_tile0, ix->ivf_ntile, ix->ivf_size, ids_dev,
                            ix->id_base, run, ix->ivf_idmap, ix->ivf_inv, ix->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
  hipFree(ids_dev);
  hipFree(src0_dev);
  if (e != hipSuccess) {
    hipFree(dst);
    return fail(e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP, std::string("ivf_set_lists: ") + hipGetErrorString(e));
  }
  if (ix->rows) hipFree(ix->rows);
  ix->rows = dst;
  ix->capacity = prow;
  // (an IVF index does not use the int8 first stage: give a copy built while the index was flat back)
  hipFree(ix->i8_rows);
  ix->i8_rows = nullptr;
  ix->i8_cap_rows = 0;
  ix->i8_nrows = 0;
  ix->i8_valid = false;
  int r = knnx_create(ix->device, ix->d, KNNX_METRIC_INNER_PRODUCT, &ix->cent);
  if (r) return r;
  r = knnx_add_f16(ix->cent, centroids_f16, nlist);
  if (r) return r;
  ix->ivf_nlist = nlist;
  ix->ivf_nprobe = std::min(ix->ivf_nprobe, nlist);
  return KNNX_OK;
}


// ---------------------------------------------------------------------------------------------
// IVF-Flat build on the device (stands in for the autofaiss call of clip_index.py:12-66 for this index
// type).  Training: Lloyd iterations on a sample that stays resident in HBM -- assignment by knn_assign_kernel (MFMA-bound,
// every workgroup streams the whole centroid matrix), update by one workgroup per list over the host-sorted member order.
// Adding: two streaming passes over the embedding files, so that no second copy of a 256 GB shard is ever needed --
// pass 1 knnx_ivfb_assign (rows -> list ids, 4 bytes per row kept by the caller), pass 2 knnx_ivf_begin /
// knnx_ivf_add_assigned / knnx_ivf_end (rows scattered straight into the tile-padded list-sorted arena).  The host only
// does integer bookkeeping (bincount, prefix sums, stable ranks inside a chunk).
// ---------------------------------------------------------------------------------------------
struct knnx_ivf_builder {
  int device = 0, d = 0, nlist = 0;
  hipStream_t stream = nullptr;
  _Float16* cent = nullptr;    // [nlist, d]
  _Float16* sample = nullptr;  // resident training sample
  int64_t n_sample = 0;
  _Float16* stage = nullptr;   // one chunk of streamed rows
  int32_t* lists = nullptr;    // assignment of a chunk / of the sample
  int64_t cap_rows = 0;
  int64_t *order = nullptr, *off = nullptr;
  void* pin = nullptr;
  size_t pin_bytes = 0;
  unsigned long long* hist = nullptr;  // [nlist] list sizes accumulated by knnx_ivfb_assign_device
  bool sample_borrowed = false;        // sample points into caller memory (knnx_ivfb_set_sample_device)
  std::vector<int32_t> h_lists;        // host scratch of knnx_ivfb_lloyd
  std::vector<int64_t> h_order, h_off;
};
static const int64_t IVFB_CHUNK = (int64_t)1 << 20;  // rows per streamed chunk

extern "C" void knnx_ivfb_destroy(knnx_ivf_builder* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  (void)hipFree(b->cent);
  if (!b->sample_borrowed) (void)hipFree(b->sample);
  (void)hipFree(b->hist);
  (void)hipFree(b->stage);
  (void)hipFree(b->lists);
  (void)hipFree(b->order);
  (void)hipFree(b->off);
  if (b->pin) (void)hipHostFree(b->pin);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

extern "C" int knnx_ivfb_create(int device, int d, int nlist, knnx_ivf_builder** out) {
  if (!out || nlist <= 0) return fail(KNNX_E_ARG, "bad ivfb_create arguments");
  *out = nullptr;
  if (d <= 0 || d % 256 != 0 || d > 1024) return fail(KNNX_E_UNSUPPORTED, "d must be a multiple of 256 and <= 1024");
  HIPCHK(hipSetDevice(device));
  knnx_ivf_builder* b = new knnx_ivf_builder();
  b->device = device;
  b->d = d;
  b->nlist = nlist;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&b->cent, (size_t)nlist * d * sizeof(_Float16));
  if (e == hipSuccess) e = hipMalloc(&b->off, (size_t)(nlist + 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&b->hist, (size_t)nlist * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(b->hist, 0, (size_t)nlist * sizeof(unsigned long long));
  if (e == hipSuccess) {
    b->pin_bytes = (size_t)IVFB_CHUNK * d * sizeof(_Float16);
    e = hipHostMalloc(&b->pin, b->pin_bytes, hipHostMallocDefault);
  }
  if (e != hipSuccess) {
    knnx_ivfb_destroy(b);
    return fail(e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP, std::string("ivfb_create: ") + hipGetErrorString(e));
  }
  *out = b;
  return KNNX_OK;
}

extern "C" int knnx_ivfb_set_centroids(knnx_ivf_builder* b, const uint16_t* centroids_f16) {
  if (!b || !centroids_f16) return fail(KNNX_E_ARG, "bad ivfb_set_centroids arguments");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemcpy(b->cent, centroids_f16, (size_t)b->nlist * b->d * sizeof(_Float16), hipMemcpyHostToDevice));
  return KNNX_OK;
}
extern "C" int knnx_ivfb_get_centroids(knnx_ivf_builder* b, uint16_t* centroids_f16) {
  if (!b || !centroids_f16) return fail(KNNX_E_ARG, "bad ivfb_get_centroids arguments");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(centroids_f16, b->cent, (size_t)b->nlist * b->d * sizeof(_Float16), hipMemcpyDeviceToHost));
  return KNNX_OK;
}

static int ivfb_rows_cap(knnx_ivf_builder* b, int64_t rows) {
  if (rows <= b->cap_rows) return 0;
  (void)hipFree(b->stage);
  (void)hipFree(b->lists);
  b->stage = nullptr;
  b->lists = nullptr;
  b->cap_rows = 0;
  HIPCHK(hipMalloc(&b->stage, (size_t)std::min<int64_t>(rows, IVFB_CHUNK) * b->d * sizeof(_Float16)));
  HIPCHK(hipMalloc(&b->lists, (size_t)rows * sizeof(int32_t)));
  b->cap_rows = rows;
  return 0;
}

// upload host rows in chunks through the pinned buffer into dst (device)
static int ivfb_upload(knnx_ivf_builder* b, const uint16_t* rows, int64_t n, _Float16* dst) {
  const size_t rb = (size_t)b->d * sizeof(_Float16);
  for (int64_t o = 0; o < n; o += IVFB_CHUNK) {
    const int64_t m = std::min(IVFB_CHUNK, n - o);
    memcpy(b->pin, rows + (size_t)o * b->d, (size_t)m * rb);
    HIPCHK(hipMemcpyAsync(dst + (size_t)o * b->d, b->pin, (size_t)m * rb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  return 0;
}

extern "C" int knnx_ivfb_set_sample(knnx_ivf_builder* b, const uint16_t* rows_f16, int64_t n) {
  if (!b || !rows_f16 || n <= 0) return fail(KNNX_E_ARG, "bad ivfb_set_sample arguments");
  HIPCHK(hipSetDevice(b->device));
  if (!b->sample_borrowed) (void)hipFree(b->sample);
  (void)hipFree(b->order);
  b->sample = nullptr;
  b->sample_borrowed = false;
  b->order = nullptr;
  HIPCHK(hipMalloc(&b->sample, (size_t)n * b->d * sizeof(_Float16)));
  HIPCHK(hipMalloc(&b->order, (size_t)n * sizeof(int64_t)));
  b->n_sample = n;
  int r = ivfb_rows_cap(b, n);
  if (r) return r;
  return ivfb_upload(b, rows_f16, n, b->sample);
}

extern "C" int knnx_ivfb_assign_sample(knnx_ivf_builder* b, int32_t* lists_out) {
  if (!b || !lists_out || !b->sample) return fail(KNNX_E_ARG, "bad ivfb_assign_sample arguments (set a sample first)");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(launch_assign(b->cent, b->nlist, b->d, b->sample, b->n_sample, b->lists, b->stream));
  HIPCHK(hipMemcpyAsync(lists_out, b->lists, (size_t)b->n_sample * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return KNNX_OK;
}

extern "C" int knnx_ivfb_update(knnx_ivf_builder* b, const int64_t* order, const int64_t* off) {
  if (!b || !order || !off || !b->sample) return fail(KNNX_E_ARG, "bad ivfb_update arguments (set a sample first)");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemcpyAsync(b->order, order, (size_t)b->n_sample * sizeof(int64_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->off, off, (size_t)(b->nlist + 1) * sizeof(int64_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(launch_kmeans_update(b->sample, b->d, b->order, b->off, b->nlist, b->cent, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return KNNX_OK;
}

extern "C" int knnx_ivfb_assign(knnx_ivf_builder* b, const uint16_t* rows_f16, int64_t n, int32_t* lists_out) {
  if (!b || (n > 0 && (!rows_f16 || !lists_out)) || n < 0) return fail(KNNX_E_ARG, "bad ivfb_assign arguments");
  if (n == 0) return KNNX_OK;
  HIPCHK(hipSetDevice(b->device));
  int r = ivfb_rows_cap(b, std::min<int64_t>(n, IVFB_CHUNK));
  if (r) return r;
  const size_t rb = (size_t)b->d * sizeof(_Float16);
  for (int64_t o = 0; o < n; o += IVFB_CHUNK) {
    const int64_t m = std::min(IVFB_CHUNK, n - o);
    memcpy(b->pin, rows_f16 + (size_t)o * b->d, (size_t)m * rb);
    HIPCHK(hipMemcpyAsync(b->stage, b->pin, (size_t)m * rb, hipMemcpyHostToDevice, b->stream));
    HIPCHK(launch_assign(b->cent, b->nlist, b->d, b->stage, m, b->lists, b->stream));
    HIPCHK(hipMemcpyAsync(lists_out + o, b->lists, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  return KNNX_OK;
}


// ---- the same builder for rows that are already in HBM (SURVEY 8 row f1 at BASELINE config 5's size: a 256 GB shard cannot
// take a detour through host memory, and nothing of it needs to: the embeddings are produced on this GPU) ----------------
extern "C" int knnx_ivfb_set_sample_device(knnx_ivf_builder* b, const void* rows_dev, int64_t n) {
  if (!b || !rows_dev || n <= 0) return fail(KNNX_E_ARG, "bad ivfb_set_sample_device arguments");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (!b->sample_borrowed) (void)hipFree(b->sample);
  (void)hipFree(b->order);
  b->sample = (_Float16*)rows_dev;  // borrowed: the caller keeps it alive until the training is over
  b->sample_borrowed = true;
  b->order = nullptr;
  HIPCHK(hipMalloc(&b->order, (size_t)n * sizeof(int64_t)));
  b->n_sample = n;
  return ivfb_rows_cap(b, n);
}

// centroid list_ids[i] := sample row sample_rows[i] (initial seeding, re-seeding of empty lists); host index arrays
extern "C" int knnx_ivfb_seed_from_sample(knnx_ivf_builder* b, const int32_t* list_ids, const int64_t* sample_rows, int64_t n) {
  if (!b || (n > 0 && (!list_ids || !sample_rows)) || n < 0 || !b->sample) return fail(KNNX_E_ARG, "bad ivfb_seed_from_sample arguments (set a sample first)");
  if (n == 0) return KNNX_OK;
  for (int64_t i = 0; i < n; ++i)
    if (list_ids[i] < 0 || list_ids[i] >= b->nlist || sample_rows[i] < 0 || sample_rows[i] >= b->n_sample)
      return fail(KNNX_E_ARG, "list id / sample row out of range");
  HIPCHK(hipSetDevice(b->device));
  int32_t* l_dev = nullptr;
  int64_t* r_dev = nullptr;
  HIPCHK(hipMalloc(&l_dev, (size_t)n * sizeof(int32_t)));
  hipError_t e = hipMalloc(&r_dev, (size_t)n * sizeof(int64_t));
  if (e == hipSuccess) e = hipMemcpyAsync(l_dev, list_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(r_dev, sample_rows, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = launch_copy_rows(b->sample, b->d, r_dev, l_dev, n, b->cent, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  (void)hipFree(l_dev);
  (void)hipFree(r_dev);
  if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("ivfb_seed_from_sample: ") + hipGetErrorString(e));
  return KNNX_OK;
}

// One Lloyd iteration over the resident sample, driven from here: assignment kernel -> list ids to the host -> counting sort
// (member order = ascending sample row inside a list: the fixed summation order of knnx_ivfb_update) -> update kernel.
// sizes_out [nlist] (host, may be null): members per list, so that the caller can re-seed the empty ones.
extern "C" int knnx_ivfb_lloyd(knnx_ivf_builder* b, int64_t* sizes_out) {
  if (!b || !b->sample) return fail(KNNX_E_ARG, "bad ivfb_lloyd arguments (set a sample first)");
  HIPCHK(hipSetDevice(b->device));
  const int64_t n = b->n_sample;
  b->h_lists.resize((size_t)n);
  b->h_order.resize((size_t)n);
  b->h_off.assign((size_t)b->nlist + 1, 0);
  HIPCHK(launch_assign(b->cent, b->nlist, b->d, b->sample, n, b->lists, b->stream));
  HIPCHK(hipMemcpyAsync(b->h_lists.data(), b->lists, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int64_t i = 0; i < n; ++i) b->h_off[(size_t)b->h_lists[i] + 1]++;
  if (sizes_out) for (int l = 0; l < b->nlist; ++l) sizes_out[l] = b->h_off[(size_t)l + 1];
  for (int l = 0; l < b->nlist; ++l) b->h_off[(size_t)l + 1] += b->h_off[l];
  {
    std::vector<int64_t> cur(b->h_off.begin(), b->h_off.end() - 1);
    for (int64_t i = 0; i < n; ++i) b->h_order[(size_t)cur[b->h_lists[i]]++] = i;
  }
  HIPCHK(hipMemcpyAsync(b->order, b->h_order.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->off, b->h_off.data(), (size_t)(b->nlist + 1) * sizeof(int64_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(launch_kmeans_update(b->sample, b->d, b->order, b->off, b->nlist, b->cent, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return KNNX_OK;
}

// pass 1 over device-resident rows: lists_dev[i] = list of row i (device int32 [n]); the list sizes accumulate in the builder
extern "C" int knnx_ivfb_assign_device(knnx_ivf_builder* b, const void* rows_dev, int64_t n, int32_t* lists_dev) {
  if (!b || (n > 0 && (!rows_dev || !lists_dev)) || n < 0) return fail(KNNX_E_ARG, "bad ivfb_assign_device arguments");
  if (n == 0) return KNNX_OK;
  HIPCHK(hipSetDevice(b->device));
  const int64_t step = (int64_t)1 << 22;  // rows per launch (grid size)
  for (int64_t o = 0; o < n; o += step) {
    const int64_t m = std::min(step, n - o);
    HIPCHK(launch_assign(b->cent, b->nlist, b->d, (const _Float16*)rows_dev + (size_t)o * b->d, m, lists_dev + o, b->stream));
  }
  HIPCHK(launch_ivf_hist(lists_dev, n, b->nlist, b->hist, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return KNNX_OK;
}

// list sizes seen by knnx_ivfb_assign_device so far -> sizes_out [nlist] (host); reset != 0 clears the counters afterwards
extern "C" int knnx_ivfb_list_sizes(knnx_ivf_builder* b, int64_t* sizes_out, int reset) {
  if (!b || !sizes_out) return fail(KNNX_E_ARG, "bad ivfb_list_sizes arguments");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "");
  HIPCHK(hipMemcpy(sizes_out, b->hist, (size_t)b->nlist * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(b->hist, 0, (size_t)b->nlist * sizeof(unsigned long long)));
  return KNNX_OK;
}

// Benchmark corpora generated straight into caller HBM: dst row i = corpus row row_begin + i * row_stride, fp16 [n, d].
// kind 0: the isotropic corpus of knnx_synth_fill (row_stride must be 1); kind 2: the same with three dominant columns (the
// anisotropy of CLIP embeddings: the int8 first stage takes its 1-plane dominant-column form on it, include/knnx.h); kind 1: the overlapping mixture of n_clusters
// Gaussians of BASELINE config 5 (knn_kernels.hip: knn_synth_mix_kernel; oracle/knn_oracle.py: synth_mixture_rows).
// Synchronous.
extern "C" int knnx_synth_rows_device(int device, void* dst_f16, int64_t row_begin, int64_t row_stride, int64_t n, int d, uint64_t seed,
                                      int kind, int64_t n_clusters, void* stream) {
  if (!dst_f16 || n < 0 || row_begin < 0 || row_stride < 1 || d <= 0 || d % 2 || d > 1024) return fail(KNNX_E_ARG, "bad synth_rows arguments");
  if (n == 0) return KNNX_OK;
  HIPCHK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0 || kind == 2) {  // 2: the isotropic corpus with three dominant columns (knn_synth_kernel)
    if (row_stride != 1) return fail(KNNX_E_UNSUPPORTED, "the isotropic corpora are generated with row_stride 1 only");
    // the kernel addresses X[row * d]: shift the base so that corpus row row_begin lands on dst row 0
    HIPCHK(launch_synth((_Float16*)dst_f16 - (size_t)row_begin * d, row_begin, n, d, seed, st, kind == 2 ? 1 : 0));
    HIPCHK(hipStreamSynchronize(st));
    return KNNX_OK;
  }
  if (kind != 1 || n_clusters <= 0) return fail(KNNX_E_ARG, "kind must be 0, 2, or 1 (with n_clusters > 0)");
  short* table = nullptr;
  HIPCHK(hipMalloc(&table, synth_mix_table_bytes(d)));
  hipError_t e = launch_synth_mix((_Float16*)dst_f16, row_begin, row_stride, n, d, seed, n_clusters, table, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(table);
  if (e != hipSuccess) return fail(KNNX_E_HIP, std::string("synth_rows: ") + hipGetErrorString(e));
  return KNNX_OK;
}

extern "C" int knnx_ivf_begin(knnx_index* ix, int nlist, const uint16_t* centroids_f16, const int64_t* list_sizes) {
  if (!ix || nlist <= 0 || !centroids_f16 || !list_sizes) return fail(KNNX_E_ARG, "bad ivf_begin arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (ix->borrowed || ix->ntotal != 0 || ix->ivf_nlist || ix->ivfb_nlist) return fail(KNNX_E_STATE, "ivf_begin needs an empty index that owns its rows");
  std::vector<unsigned> tile0(nlist), ntile(nlist), size(nlist);
  int64_t run = 0, tiles = 0;
  for (int l = 0; l < nlist; ++l) {
    if (list_sizes[l] < 0) return fail(KNNX_E_ARG, "negative list size");
    size[l] = (unsigned)list_sizes[l];
    ntile[l] = (unsigned)((list_sizes[l] + 31) / 32);
    tile0[l] = (unsigned)tiles;
    run += list_sizes[l];
    tiles += ntile[l];
  }
  if (tiles * 32 > (int64_t)0xffffffffll) return fail(KNNX_E_UNSUPPORTED, "more than 2^32 padded rows per device");
  const int64_t prow = std::max<int64_t>(tiles * 32, 32);
  if (ix->rows) hipFree(ix->rows);
  ix->rows = nullptr;
  ix->capacity = 0;
  HIPCHK(hipMalloc(&ix->rows, (size_t)prow * ix->d * sizeof(_Float16)));
  ix->capacity = prow;
  hipError_t e = hipMemsetAsync(ix->rows, 0, (size_t)prow * ix->d * sizeof(_Float16), ix->stream);  // pad rows are zero
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_tile0, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_ntile, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_size, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_masks, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_off, nlist * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_nwork, sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_work, (size_t)std::max<int64_t>(tiles, 1) * sizeof(uint4));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_idmap, (size_t)prow * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_inv, (size_t)std::max<int64_t>(run, 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_Ic, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivf_Dc, (size_t)KNN_NQ * KNNX_MAX_K_FAST * sizeof(float));
  if (e == hipSuccess) e = hipMemsetAsync(ix->ivf_idmap, 0xFF, (size_t)prow * sizeof(int64_t), ix->stream);  // -1 on pad rows
  if (e == hipSuccess) e = hipMemcpyAsync(ix->ivf_tile0, tile0.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice, ix->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ix->ivf_ntile, ntile.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice, ix->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ix->ivf_size, size.data(), nlist * sizeof(unsigned), hipMemcpyHostToDevice, ix->stream);
  if (e == hipSuccess) e = hipMalloc(&ix->ivfb_rows, (size_t)IVFB_CHUNK * ix->d * sizeof(_Float16));
  if (e == hipSuccess) e = hipMalloc(&ix->ivfb_ids, (size_t)IVFB_CHUNK * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivfb_lists, (size_t)IVFB_CHUNK * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&ix->ivfb_pos, (size_t)IVFB_CHUNK * sizeof(int32_t));
  if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP, std::string("ivf_begin: ") + hipGetErrorString(e));
  ix->ivfb_cent.assign(centroids_f16, centroids_f16 + (size_t)nlist * ix->d);
  ix->ivfb_size.assign(size.begin(), size.end());
  ix->ivfb_tile0.assign(tile0.begin(), tile0.end());
  ix->ivfb_fill.assign(nlist, 0u);
  ix->ivfb_taken.assign((size_t)(prow + 63) / 64, 0ull);
  ix->ivfb_nlist = nlist;
  ix->ivfb_total = run;
  ix->ivfb_added = 0;
  return KNNX_OK;
}

// claim arena slot (list, pos) for one row; caller holds ix->mu.  Returns false (message set) on a bad or reused slot.
static bool ivfb_claim(knnx_index* ix, int32_t list, int64_t pos) {
  if (list < 0 || list >= ix->ivfb_nlist) { fail(KNNX_E_ARG, "list id out of range"); return false; }
  if (pos < 0 || pos >= (int64_t)ix->ivfb_size[list]) { fail(KNNX_E_ARG, "position is not inside its list (0 <= pos < list size)"); return false; }
  const size_t slot = (size_t)ix->ivfb_tile0[list] * 32 + (size_t)pos;
  uint64_t& word = ix->ivfb_taken[slot >> 6];
  const uint64_t bit = 1ull << (slot & 63);
  if (word & bit) { fail(KNNX_E_ARG, "two rows were given the same (list, position)"); return false; }
  word |= bit;
  ix->ivfb_fill[list]++;
  return true;
}

extern "C" int knnx_ivf_add_assigned(knnx_index* ix, const uint16_t* rows_f16, int64_t n, const int64_t* ids, const int32_t* lists,
                                     const int32_t* pos) {
  if (!ix || (n > 0 && (!rows_f16 || !ids || !lists || !pos)) || n < 0) return fail(KNNX_E_ARG, "bad ivf_add_assigned arguments");
  if (n == 0) return KNNX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (!ix->ivfb_nlist) return fail(KNNX_E_STATE, "call knnx_ivf_begin first");
  if (ix->ivfb_added + n > ix->ivfb_total) return fail(KNNX_E_ARG, "more rows than the list sizes announced");
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] < ix->id_base || ids[i] - ix->id_base >= ix->ivfb_total) return fail(KNNX_E_ARG, "ids must lie in [id_base, id_base + total rows)");
  for (int64_t i = 0; i < n; ++i)
    if (!ivfb_claim(ix, lists[i], pos[i])) {
      // roll the claims of this call back: a refused call leaves the build as it was
      for (int64_t j = 0; j < i; ++j) {
        const size_t slot = (size_t)ix->ivfb_tile0[lists[j]] * 32 + (size_t)pos[j];
        ix->ivfb_taken[slot >> 6] &= ~(1ull << (slot & 63));
        ix->ivfb_fill[lists[j]]--;
      }
      return KNNX_E_ARG;
    }
  const size_t rb = (size_t)ix->d * sizeof(_Float16);
  int r = ensure_pin(ix, (size_t)IVFB_CHUNK * (rb + 16));
  if (r) return r;
  char* pin = (char*)ix->pin;
  for (int64_t o = 0; o < n; o += IVFB_CHUNK) {
    const int64_t m = std::min(IVFB_CHUNK, n - o);
    char* p_rows = pin;
    char* p_ids = p_rows + (size_t)IVFB_CHUNK * rb;
    char* p_lists = p_ids + (size_t)IVFB_CHUNK * 8;
    char* p_pos = p_lists + (size_t)IVFB_CHUNK * 4;
    memcpy(p_rows, rows_f16 + (size_t)o * ix->d, (size_t)m * rb);
    memcpy(p_ids, ids + o, (size_t)m * 8);
    memcpy(p_lists, lists + o, (size_t)m * 4);
    memcpy(p_pos, pos + o, (size_t)m * 4);
    HIPCHK(hipMemcpyAsync(ix->ivfb_rows, p_rows, (size_t)m * rb, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(ix->ivfb_ids, p_ids, (size_t)m * 8, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(ix->ivfb_lists, p_lists, (size_t)m * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(ix->ivfb_pos, p_pos, (size_t)m * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(launch_ivf_scatter((const _Float16*)ix->ivfb_rows, m, ix->d, (const int32_t*)ix->ivfb_lists, (const int32_t*)ix->ivfb_pos,
                              (const int64_t*)ix->ivfb_ids, 0, ix->ivf_tile0, ix->id_base, ix->ivfb_total, ix->rows, ix->ivf_idmap,
                              ix->ivf_inv, ix->stream));
    HIPCHK(launch_maxnorm((const _Float16*)ix->ivfb_rows, m, ix->d, ix->maxnorm, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
  }
  ix->ivfb_added += n;
  return KNNX_OK;
}

// The same pass for rows that are already in HBM (a shard generated or encoded on this GPU): row i carries id id0 + i and
// goes to the next free position of lists_dev[i] -- positions inside a list follow the order of the calls and of the rows
// inside a call, exactly what knn.build_ivf_index hands to knnx_ivf_add_assigned.  The list ids travel to the host once
// (4 bytes per row) for the position bookkeeping; the rows never leave the device.
extern "C" int knnx_ivf_add_assigned_device(knnx_index* ix, const void* rows_dev, int64_t n, int64_t id0, const int32_t* lists_dev) {
  if (!ix || (n > 0 && (!rows_dev || !lists_dev)) || n < 0) return fail(KNNX_E_ARG, "bad ivf_add_assigned_device arguments");
  if (n == 0) return KNNX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  if (!ix->ivfb_nlist) return fail(KNNX_E_STATE, "call knnx_ivf_begin first");
  if (ix->ivfb_added + n > ix->ivfb_total) return fail(KNNX_E_ARG, "more rows than the list sizes announced");
  if (id0 < ix->id_base || id0 - ix->id_base + n > ix->ivfb_total) return fail(KNNX_E_ARG, "ids must lie in [id_base, id_base + total rows)");
  int r = ensure_pin(ix, (size_t)IVFB_CHUNK * 8);
  if (r) return r;
  int32_t* h_lists = (int32_t*)ix->pin;
  int32_t* h_pos = h_lists + IVFB_CHUNK;
  // Validate the WHOLE call before a single slot is claimed (a refusal half way through a call left claims behind
  // and the build could neither be retried nor finished): every list id in range, no list filled beyond its announced size.
  // Costs a second read-back of the 4-byte list ids (20 ms per 125 M rows).
  {
    std::vector<uint32_t> add((size_t)ix->ivfb_nlist, 0u);
    for (int64_t o = 0; o < n; o += IVFB_CHUNK) {
      const int64_t m = std::min(IVFB_CHUNK, n - o);
      HIPCHK(hipMemcpyAsync(h_lists, lists_dev + o, (size_t)m * 4, hipMemcpyDeviceToHost, ix->stream));
      HIPCHK(hipStreamSynchronize(ix->stream));
      for (int64_t i = 0; i < m; ++i) {
        const int32_t l = h_lists[i];
        if (l < 0 || l >= ix->ivfb_nlist) return fail(KNNX_E_ARG, "list id out of range (nothing was added)");
        if ((uint64_t)ix->ivfb_fill[l] + ++add[l] > (uint64_t)ix->ivfb_size[l])
          return fail(KNNX_E_ARG, "a list would receive more rows than its announced size (nothing was added)");
      }
    }
  }
  for (int64_t o = 0; o < n; o += IVFB_CHUNK) {
    const int64_t m = std::min(IVFB_CHUNK, n - o);
    HIPCHK(hipMemcpyAsync(h_lists, lists_dev + o, (size_t)m * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    for (int64_t i = 0; i < m; ++i) {
      const int32_t l = h_lists[i];
      const int64_t p = ix->ivfb_fill[l];
      // cannot fail after the validation above unless host-variant calls claimed positions out of order in this list
      if (!ivfb_claim(ix, l, p)) return KNNX_E_ARG;
      h_pos[i] = (int32_t)p;
    }
    HIPCHK(hipMemcpyAsync(ix->ivfb_pos, h_pos, (size_t)m * 4, hipMemcpyHostToDevice, ix->stream));
    const _Float16* src = (const _Float16*)rows_dev + (size_t)o * ix->d;
    HIPCHK(launch_ivf_scatter(src, m, ix->d, lists_dev + o, (const int32_t*)ix->ivfb_pos, nullptr, id0 + o, ix->ivf_tile0, ix->id_base,
                              ix->ivfb_total, ix->rows, ix->ivf_idmap, ix->ivf_inv, ix->stream));
    HIPCHK(launch_maxnorm(src, m, ix->d, ix->maxnorm, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
  }
  ix->ivfb_added += n;
  return KNNX_OK;
}

extern "C" int knnx_ivf_end(knnx_index* ix) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  int nlist;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (set_dev(ix)) return KNNX_E_HIP;
    if (!ix->ivfb_nlist) return fail(KNNX_E_STATE, "call knnx_ivf_begin first");
    if (ix->ivfb_added != ix->ivfb_total) return fail(KNNX_E_STATE, "fewer rows were added than the list sizes announced");
    nlist = ix->ivfb_nlist;
    for (int l = 0; l < nlist; ++l)
      if (ix->ivfb_fill[l] != ix->ivfb_size[l]) return fail(KNNX_E_STATE, "a list received fewer rows than its announced size");
    std::vector<uint32_t>().swap(ix->ivfb_size);
    std::vector<uint32_t>().swap(ix->ivfb_fill);
    std::vector<uint32_t>().swap(ix->ivfb_tile0);
    std::vector<uint64_t>().swap(ix->ivfb_taken);
    hipFree(ix->ivfb_rows); ix->ivfb_rows = nullptr;
    hipFree(ix->ivfb_ids); ix->ivfb_ids = nullptr;
    hipFree(ix->ivfb_lists); ix->ivfb_lists = nullptr;
    hipFree(ix->ivfb_pos); ix->ivfb_pos = nullptr;
  }
  int r = knnx_create(ix->device, ix->d, KNNX_METRIC_INNER_PRODUCT, &ix->cent);
  if (r) return r;
  r = knnx_add_f16(ix->cent, ix->ivfb_cent.data(), nlist);
  if (r) return r;
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->ntotal = ix->ivfb_total;
  ix->ivf_nlist = nlist;
  ix->ivf_nprobe = std::min(ix->ivf_nprobe, nlist);
  ix->ivfb_nlist = 0;
  ix->ivfb_cent.clear();
  ix->ivfb_cent.shrink_to_fit();
  return KNNX_OK;
}

extern "C" int knnx_ivf_set_nprobe(knnx_index* ix, int nprobe) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  if (nprobe < 1) return fail(KNNX_E_ARG, "nprobe must be >= 1");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->ivf_nprobe = nprobe;
  return KNNX_OK;
}

extern "C" int knnx_ivf_nlist(const knnx_index* ix) { return ix ? ix->ivf_nlist : 0; }
extern "C" int knnx_ivf_nprobe(const knnx_index* ix) { return ix ? ix->ivf_nprobe : 0; }

extern "C" int knnx_merge_topk_device(int device, const float* D_parts, const int64_t* I_parts, int P, int n, int k,
                                      float* D_out, int64_t* I_out, void* stream) {
  if (!D_parts || !I_parts || !D_out || !I_out || P <= 0 || n < 0 || k <= 0) return fail(KNNX_E_ARG, "bad merge arguments");
  if (n == 0) return KNNX_OK;
  HIPCHK(hipSetDevice(device));
  if (k > KNNX_MAX_K_FAST) {
    // large k (the front end's num_result_ids = 3000): every shard's list is sorted (score desc, id asc, -1 padding at its tail), so the
    // P-way merge of sorted lists applies -- the kernel the in-process sharded index uses for the same case (knnx_sharded.hip)
    if (P > 64) return fail(KNNX_E_ARG, "merge of k > 64 results takes at most 64 shards");
    HIPCHK(launch_merge_sorted(D_parts, I_parts, P, n, k, D_out, I_out, (hipStream_t)stream));
    return KNNX_OK;
  }
  HIPCHK(launch_merge_i64(D_parts, I_parts, P, n, k, k, D_out, I_out, (hipStream_t)stream));
  return KNNX_OK;
}

extern "C" int knnx_get_stats(knnx_index* ix, int64_t* proof_queries, int64_t* proof_failures) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipStreamSynchronize(ix->stream));
  HIPCHK(hipMemcpy(h, ix->stats, sizeof(h), hipMemcpyDeviceToHost));
  if (proof_queries) *proof_queries = (int64_t)h[0];
  if (proof_failures) *proof_failures = (int64_t)h[1];
  return KNNX_OK;
}

// IVF: 32-row tiles in the work list of the most recent scan (rows of the probed lists, padded to tiles): what that scan
// read from HBM is tiles * 32 * d * 2 bytes -- the measured side of "(nprobe / nlist) * N * d * 2" 
extern "C" int knnx_ivf_last_scan_tiles(knnx_index* ix, int64_t* tiles) {
  if (!ix || !tiles) return fail(KNNX_E_ARG, "bad ivf_last_scan_tiles arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!ix->ivf_nlist) return fail(KNNX_E_STATE, "not an IVF index");
  if (set_dev(ix)) return KNNX_E_HIP;
  HIPCHK(hipStreamSynchronize(ix->stream));
  if (ix->ivfm_last_blk > 0) {  // the multi-block pass: every block walks its own list of tiles
    unsigned n[IVFM_BLK] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(n, ix->ivfm_nwork, (size_t)ix->ivfm_last_blk * sizeof(unsigned), hipMemcpyDeviceToHost));
    *tiles = 0;
    for (int b = 0; b < ix->ivfm_last_blk; ++b) *tiles += (int64_t)n[b];
    return KNNX_OK;
  }
  unsigned n = 0;
  HIPCHK(hipMemcpy(&n, ix->ivf_nwork, sizeof(unsigned), hipMemcpyDeviceToHost));
  *tiles = (int64_t)n;
  return KNNX_OK;
}

// the same for the UNION of the lists the blocks of the most recent multi-block pass probed (a list counted once however many
// blocks read it): the bytes a single pass over shared lists would have read.  Needs profiling on (knnx_profile_enable) during the
// search; a single-block scan reports its own tiles.
extern "C" int knnx_ivf_last_scan_union_tiles(knnx_index* ix, int64_t* tiles) {
  if (!ix || !tiles) return fail(KNNX_E_ARG, "bad ivf_last_scan_union_tiles arguments");
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->ivf_nlist) return fail(KNNX_E_STATE, "not an IVF index");
    if (ix->ivfm_last_blk > 0) {
      if (!ix->ivfm_union_valid) return fail(KNNX_E_STATE, "the union of the lists is counted only while profiling is enabled (knnx_profile_enable)");
      if (set_dev(ix)) return KNNX_E_HIP;
      HIPCHK(hipStreamSynchronize(ix->stream));
      unsigned n = 0;
      HIPCHK(hipMemcpy(&n, ix->ivfm_nwork + IVFM_BLK, sizeof(unsigned), hipMemcpyDeviceToHost));
      *tiles = (int64_t)n;
      return KNNX_OK;
    }
  }
  return knnx_ivf_last_scan_tiles(ix, tiles);
}

extern "C" int knnx_profile_enable(knnx_index* ix, int on) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->prof = on != 0;
  return KNNX_OK;
}

extern "C" int knnx_profile_get(knnx_index* ix, int64_t* scan_launches, double* scan_ms) {
  if (!ix) return fail(KNNX_E_ARG, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (set_dev(ix)) return KNNX_E_HIP;
  int64_t n = 0;
  double ms = 0.0;
  for (auto& ev : ix->prof_events) {
    HIPCHK(hipEventSynchronize(ev.second));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, ev.first, ev.second));
    ms += t;
    ++n;
    hipEventDestroy(ev.first);
    hipEventDestroy(ev.second);
  }
  ix->prof_events.clear();
  if (scan_launches) *scan_launches = n;
  if (scan_ms) *scan_ms = ms;
  return KNNX_OK;
}
