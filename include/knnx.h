/*
 * knnx.h -- C ABI of the MI355X-native inner-product kNN (search half of the hot path).
 *
 * This is the boundary a clip-retrieval maintainer binds (ctypes stub in INTEGRATION.md)
 * to replace the faiss `Index` object held in `ClipResource.image_index/.text_index`
 * (reference clip_retrieval/clip_back.py:781-782, built by `load_index` :589-596).
 * Each entry point names the reference call it stands in for.  Plain pointers and
 * sizes only; no torch / faiss types.  All functions return 0 on success or a negative
 * KNNX_E_* code; knnx_last_error() gives a thread-local message.  Nothing throws.
 *
 * Index rows are stored as fp16 [ntotal, d] row-major, resident in HBM.  Scores are
 * the fp32 inner product of the fp16 row with the fp32 query (query split hi/lo into
 * two fp16 MFMA operands, fp32 accumulate).  Result order: score descending, ties by
 * ascending id; missing results are id -1 / score -FLT_MAX (faiss IP padding).
 */
#ifndef KNNX_H
#define KNNX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct knnx_index knnx_index;

enum {
  KNNX_OK = 0,
  KNNX_E_ARG = -1,        /* bad argument (null, d unsupported, k out of range ...) */
  KNNX_E_HIP = -2,        /* a HIP runtime call failed (message has the hipError) */
  KNNX_E_NOMEM = -3,      /* device allocation failed / capacity exceeded */
  KNNX_E_STATE = -4,      /* call not valid in this state (e.g. add on an attached index) */
  KNNX_E_UNSUPPORTED = -5 /* valid request this build has no kernel for */
};

#define KNNX_METRIC_INNER_PRODUCT 0 /* faiss.METRIC_INNER_PRODUCT; the only metric clip_back uses */
#define KNNX_MAX_K_FAST 64          /* k <= 64: single-scan LDS candidate queues               */
#define KNNX_MAX_K 131072           /* 64 < k <= 131072: one range scan above a sampled / extrapolated threshold, hits ranked on the
                                     * host (the reference advertises K = 100 000: README.md:301, clip_back.py:358)        */

/* faiss.IndexFlatIP(d) / faiss.read_index(...) (clip_back.py:589-596).  `device` is the HIP
 * ordinal.  d must be a multiple of 256 and <= 1024 (CLIP embedding widths 512/768/1024). */
int knnx_create(int device, int d, int metric, knnx_index** out);
void knnx_destroy(knnx_index* ix);

/* Pre-size the HBM arena for n_rows rows (one hipMalloc; 288 GB parts leave no room to
 * grow by doubling).  Optional: add() grows geometrically if it was not called. */
int knnx_reserve(knnx_index* ix, int64_t n_rows);

/* faiss Index.add(x): append n rows.  Host pointers (pageable ok).  _f16 takes the exact
 * bytes of an `img_emb_*.npy` payload (clip_inference/writer.py:67-75); _f32 rounds to fp16
 * on the device. ids are implicit: id_base + row ordinal. */
int knnx_add_f16(knnx_index* ix, const uint16_t* rows, int64_t n);
int knnx_add_f32(knnx_index* ix, const float* rows, int64_t n);

/* Borrow n rows of fp16 already resident in HBM (caller keeps ownership; used by the
 * benchmark's on-device generator and by shard loaders that hipMemcpy themselves). */
int knnx_attach_device_f16(knnx_index* ix, const void* dev_rows, int64_t n);

/* faiss Index.reset(): drop all rows, keep the arena (flat indexes; the per-request dedup index of clip_back.py:290-294). */
int knnx_reset(knnx_index* ix);

/* Global id of local row 0 (row-sharded multi-GPU index: shard g has id_base = g*N/G). */
int knnx_set_id_base(knnx_index* ix, int64_t id_base);

/* faiss Index.ntotal / Index.d (ivf_metadata_ordering.py:51). */
int64_t knnx_ntotal(const knnx_index* ix);
int knnx_dim(const knnx_index* ix);

/* faiss Index.search(x, k) (clip_filter.py:55) and Index.search_and_reconstruct(x, k)
 * (clip_back.py:362).  q: host f32 [n, d] C-contiguous.  D: f32 [n, k], I: int64 [n, k],
 * R (may be NULL): f32 [n, k, d] (rows of id -1 are filled with 0xFF bytes like faiss).
 * Re-entrant; concurrent callers are serialised on the index's stream.  One pass over HBM serves up to 32 queries
 * (exact hi/lo scores), 64 (wide scan + proof) or, on flat indexes of >= 2 Mi rows with k <= 48, 256 queries (128 at
 * d = 1024): the register-stationary scan of csrc/knn_rq_kernels.hip; every path returns the exact top-k. */
int knnx_search(knnx_index* ix, const float* q, int n, int k, float* D, int64_t* I, float* R);

/* Same with every buffer already in HBM (benchmark / all-gather path).  `stream` is a
 * hipStream_t (NULL = the index's own stream).  Asynchronous on that stream; k <= 64.  The handle's scratch is shared
 * by all calls: consecutive calls are ordered by an event even when they use different streams. */
int knnx_search_device(knnx_index* ix, const float* q_dev, int n, int k, float* D_dev,
                       int64_t* I_dev, void* stream);

/* faiss Index.reconstruct_batch: ids are global; out f32 [n, d]; id -1 -> 0xFF fill. */
int knnx_reconstruct(knnx_index* ix, const int64_t* ids, int64_t n, float* out);

/* faiss Index.range_search(x, thresh) (clip_filter.py:52; clip_back.py:294 on k<=3000 rows):
 * all rows with <q,x> > thresh.  Two calls: pass I=D=NULL to get lims[n+1] (prefix counts),
 * then call again with buffers of lims[n] entries.  Ids ascending inside each query. */
int knnx_range_search(knnx_index* ix, const float* q, int n, float thresh, int64_t* lims,
                      float* D, int64_t* I);
/* The same in ONE pass for a caller with a good guess of the result size (the per-request dedup, clip_back.py:290-294):
 * lims [n + 1] is always filled; if lims[n] <= capacity the hits are written to D / I and the call returns 0, otherwise
 * D / I are left alone and it returns 1 (retry with knnx_range_search and lims[n] entries). */
int knnx_range_search_once(knnx_index* ix, const float* q, int n, float thresh, int64_t* lims, float* D, int64_t* I,
                           int64_t capacity);

/* faiss IndexIVFFlat(quantizer=IndexFlatIP, d, nlist, METRIC_INNER_PRODUCT) (the index family of BASELINE config 5;
 * the reference gets its indices from autofaiss, clip_index.py:12-66).  Protocol: knnx_add_* the rows GROUPED BY LIST
 * (list 0 first), then call knnx_ivf_set_lists once: centroids fp16 [nlist, d] (the coarse quantiser is a flat scan
 * over them), list_sizes [nlist], ids [ntotal] = the id each added row carries (a permutation of
 * [id_base, id_base + ntotal)).  Afterwards knnx_search* probe the nprobe lists whose centroids score highest for the
 * query (faiss `nprobe`, clip_back.py:357-369) and return exactly the top-k of those lists' rows (k up to KNNX_MAX_K;
 * fewer rows than k in the probed lists: -1 / -FLT_MAX padding like faiss); knnx_range_search returns the rows of the probed
 * lists above the threshold (faiss IndexIVF.range_search; clip_filter.py:52).  add is refused on an IVF index. */
int knnx_ivf_set_lists(knnx_index* ix, int nlist, const uint16_t* centroids_f16, const int64_t* list_sizes,
                       const int64_t* ids);
int knnx_ivf_set_nprobe(knnx_index* ix, int nprobe); /* 1 .. nlist (BASELINE config 5: 16 / 64 / 256) */
int knnx_ivf_nlist(const knnx_index* ix);
int knnx_ivf_nprobe(const knnx_index* ix);  /* the value knnx_ivf_set_nprobe left (1 after knnx_ivf_set_lists / knnx_ivf_end) */

/* ---- IVF-Flat build on the device (stands in for the autofaiss call of clip_index.py:12-66) ----------
 * Training: a builder keeps the centroids and a training sample resident in HBM.  One Lloyd iteration =
 * knnx_ivfb_assign_sample (argmax over the centroids by the MFMA assignment kernel; ties -> smaller list id) + host
 * bookkeeping (stable sort of the list ids -> order, prefix sums -> off) + knnx_ivfb_update (unit-norm mean of each list's members -- spherical k-means, what inner-product assignment needs --,
 * fixed summation order; empty lists keep their centroid for the caller to re-seed through knnx_ivfb_set_centroids).
 * Adding without a second copy of the shard: pass 1 knnx_ivfb_assign streams the rows and returns their list ids;
 * pass 2 knnx_ivf_begin(list sizes) / knnx_ivf_add_assigned(rows, ids, list, position inside the list) / knnx_ivf_end
 * scatters the rows straight into the list-sorted, tile-padded arena of an empty index.  Host pointers throughout. */
typedef struct knnx_ivf_builder knnx_ivf_builder;
int knnx_ivfb_create(int device, int d, int nlist, knnx_ivf_builder** out);
void knnx_ivfb_destroy(knnx_ivf_builder* b);
int knnx_ivfb_set_centroids(knnx_ivf_builder* b, const uint16_t* centroids_f16);
int knnx_ivfb_get_centroids(knnx_ivf_builder* b, uint16_t* centroids_f16);
int knnx_ivfb_set_sample(knnx_ivf_builder* b, const uint16_t* rows_f16, int64_t n);
int knnx_ivfb_assign_sample(knnx_ivf_builder* b, int32_t* lists_out);
int knnx_ivfb_update(knnx_ivf_builder* b, const int64_t* order, const int64_t* off);
int knnx_ivfb_assign(knnx_ivf_builder* b, const uint16_t* rows_f16, int64_t n, int32_t* lists_out);
int knnx_ivf_begin(knnx_index* ix, int nlist, const uint16_t* centroids_f16, const int64_t* list_sizes);
int knnx_ivf_add_assigned(knnx_index* ix, const uint16_t* rows_f16, int64_t n, const int64_t* ids, const int32_t* lists,
                          const int32_t* pos);
int knnx_ivf_end(knnx_index* ix);
/* The same build for rows that are ALREADY IN HBM (BASELINE config 5: a 125 M x 1024 shard is 256 GB of the 288 GB -- it is
 * produced on the GPU and can neither visit host memory nor exist twice).  Device pointers; every call is synchronous.
 *   training   knnx_ivfb_set_sample_device (borrows the caller's sample rows), knnx_ivfb_seed_from_sample (initial centroids and
 *              re-seeding of empty lists: centroid list_ids[i] := sample row sample_rows[i]; host index arrays),
 *              knnx_ivfb_lloyd = one whole iteration (assign, counting sort on the host, update); sizes_out [nlist] or NULL
 *   pass 1     knnx_ivfb_assign_device: lists_dev[i] = list of row i, list sizes accumulate in the builder ->
 *              knnx_ivfb_list_sizes (host int64 [nlist]; reset != 0 clears the counters)
 *   pass 2     knnx_ivf_begin(sizes) ... knnx_ivf_add_assigned_device (row i carries id id0 + i and takes the next free
 *              position of its list) ... knnx_ivf_end
 * knnx_ivf_add_assigned (host) and _device both refuse a list id out of range, a position outside its list and a (list,
 * position) used twice; knnx_ivf_end refuses a list that did not receive its announced number of rows. */
int knnx_ivfb_set_sample_device(knnx_ivf_builder* b, const void* rows_dev_f16, int64_t n);
int knnx_ivfb_seed_from_sample(knnx_ivf_builder* b, const int32_t* list_ids, const int64_t* sample_rows, int64_t n);
int knnx_ivfb_lloyd(knnx_ivf_builder* b, int64_t* sizes_out);
int knnx_ivfb_assign_device(knnx_ivf_builder* b, const void* rows_dev_f16, int64_t n, int32_t* lists_dev);
int knnx_ivfb_list_sizes(knnx_ivf_builder* b, int64_t* sizes_out, int reset);
int knnx_ivf_add_assigned_device(knnx_index* ix, const void* rows_dev_f16, int64_t n, int64_t id0, const int32_t* lists_dev);

/* Merge P per-shard results ([P, n, k] each, already global ids) into the top-k [n, k];
 * the step after the RCCL all-gather of a row-sharded index.  Device buffers.  k <= 64: any order within a list;
 * k > 64: every list sorted as knnx_search returns it (score descending, -1 padding at the tail), P <= 64. */
int knnx_merge_topk_device(int device, const float* D_parts, const int64_t* I_parts, int P, int n,
                           int k, float* D_out, int64_t* I_out, void* stream);

/* ---- one process, several GPUs: a row-sharded index behind ONE handle ----------------------------------------
 * `KnnService` keeps one index object per modality in one process and calls it from its request threads
 * (clip_back.py:343-362, 781-782, 1018); SURVEY 8(b) `knnx_create(n_devices, devices, ...)`, 8(e).  Shard g lives on
 * devices[g] (a device may be listed more than once) and holds the contiguous global row range [lo_g, hi_g), ids =
 * global row numbers.  A search sends the queries to every device, scans all shards concurrently (one stream per device),
 * copies the per-shard top-k (n*k*12 bytes) peer-to-peer over xGMI to devices[0] and merges there with the kernel of
 * knnx_merge_topk_device.  Same result contract as knnx_search.  Thread-safe (calls are serialised). */
typedef struct knnx_shards knnx_shards;
int knnx_shards_create(int n_shards, const int* devices, int d, int metric, knnx_shards** out);
/* Take ownership of per-device indexes built elsewhere (flat or IVF-Flat with replicated centroids; shard g must have
 * been given id_base = row_lo[g]).  On success the shards are destroyed with the handle. */
int knnx_shards_adopt(int n_shards, knnx_index* const* shards, const int* devices, const int64_t* row_lo, knnx_shards** out);
void knnx_shards_destroy(knnx_shards* s);
/* Fix the row range of every shard (shard g = rows [g*T/G, (g+1)*T/G)) and size its arena; required before add. */
int knnx_shards_reserve(knnx_shards* s, int64_t total_rows);
/* faiss Index.add in global row order: fills shard 0, then shard 1, ... (streams to the owning shard). */
int knnx_shards_add_f16(knnx_shards* s, const uint16_t* rows, int64_t n);
int knnx_shards_add_f32(knnx_shards* s, const float* rows, int64_t n);
/* Benchmark corpus: shard g = knnx_synth_fill(rows_per_shard, seed + g), id_base = g * rows_per_shard. */
int knnx_shards_synth_fill(knnx_shards* s, int64_t rows_per_shard, uint64_t seed);
int64_t knnx_shards_ntotal(const knnx_shards* s);
int knnx_shards_count(const knnx_shards* s);
/* How the per-shard top-k lists reach devices[0]: 0 = hipMemcpyPeerAsync per shard (the default), 1 = one grouped ncclAllGather over
 * RCCL (opt-in: KNNX_SHARDS_RCCL=1 in the environment when the handle is created, every shard on its own device and librccl.so
 * loadable -- loaded with dlopen on first use; an exchange that fails switches the handle back to 0 and the batch is answered through
 * the peer copies), -1 = null handle.  Either way the same merge kernel produces the result. */
int knnx_shards_exchange(const knnx_shards* s);
knnx_index* knnx_shards_get(knnx_shards* s, int g); /* borrowed: profiling, nprobe */
/* faiss Index.search / search_and_reconstruct / reconstruct_batch / range_search over all shards. */
int knnx_shards_search(knnx_shards* s, const float* q, int n, int k, float* D, int64_t* I, float* R);
int knnx_shards_reconstruct(knnx_shards* s, const int64_t* ids, int64_t n, float* out);
int knnx_shards_range_search(knnx_shards* s, const float* q, int n, float thresh, int64_t* lims, float* D, int64_t* I);

/* Counters of the proof-based scans (64-query wide scan, 256-query RQ scan): queries they served and queries whose
 * exactness proof failed and were re-run by the exact 32-query scan (each failure costs one more pass over HBM). */
int knnx_get_stats(knnx_index* ix, int64_t* proof_queries, int64_t* proof_failures);

/* IVF: number of 32-row tiles the most recent scan walked (the probed lists of its queries, padded to tiles; a call of more than
 * 32 queries is ONE multi-block pass of up to 256 -- every block of 32 walks the union of its own queries' lists, and the sum over
 * the blocks is reported); bytes read from HBM = tiles * 32 * d * 2, to be compared with (nprobe / nlist) * N * d * 2. */
int knnx_ivf_last_scan_tiles(knnx_index* ix, int64_t* tiles);
/* The same for the union over ALL queries of that pass (a list counted once however many blocks read it): the bytes one pass over
 * shared lists would read.  Collected only while profiling is enabled (knnx_profile_enable) during the search. */
int knnx_ivf_last_scan_union_tiles(knnx_index* ix, int64_t* tiles);

/* Live kernel timing for bench.py: when enabled, every scan launch is bracketed with
 * hipEvents on its own stream; get returns launches and summed milliseconds, then resets. */
int knnx_profile_enable(knnx_index* ix, int on);
int knnx_profile_get(knnx_index* ix, int64_t* scan_launches, double* scan_ms);

/* Fill n rows of an attached/reserved arena with the benchmark's synthetic corpus:
 * row r = L2-normalised N(0,1)^d from a counter-based hash of (seed, r, col), rounded to
 * fp16.  Re-derivable on the CPU (oracle/knn_oracle.py:synth_rows). */
int knnx_synth_fill(knnx_index* ix, int64_t n, uint64_t seed);

/* Benchmark corpora generated straight into caller HBM (fp16 [n, d]; dst row i = corpus row row_begin + i * row_stride; any row
 * is re-derivable on the CPU: oracle/knn_oracle.py).  kind 0: the isotropic corpus of knnx_synth_fill (row_stride 1 only);
 * kind 1: BASELINE config 5's overlapping mixture of n_clusters Gaussians in a 32-dimensional latent space, where IVF recall
 * is < 1 at small nprobe and rises with it; kind 2: the isotropic corpus with three dominant columns (6 x the spread plus a common
 * offset, as a few dimensions of real CLIP embeddings have: the int8 first stage treats them as dominant columns, knnx_i8_dominant;
 * row_stride 1 only).
 * `stream`: hipStream_t or NULL; synchronous. */
int knnx_synth_rows_device(int device, void* dst_f16, int64_t row_begin, int64_t row_stride, int64_t n, int d, uint64_t seed,
                           int kind, int64_t n_clusters, void* stream);

/* ---- request coalescing ("knnx_search is re-entrant; internally a batching queue") ---------------------------
 * The service calls the index from concurrent request threads with ONE query each (clip_back.py:1018 -> :362).  With coalescing
 * on (the default), knnx_search calls with n == 1 and k <= 64 that arrive while the GPU is busy wait in a queue inside the
 * library; the first caller to find no leader serves everything queued with the same k -- up to one scan's worth, 256 queries --
 * in ONE pass over HBM (one batched gather for the callers that want R), hands each caller its slice and passes the lead on.
 * Results are those of the uncoalesced call: the same ids, scores equal to f32 summation order (the scan kernel that serves a
 * query depends on how many queries share its pass).  Calls with n > 1 or
 * k > 64 are served directly.  knnx_set_coalesce(ix, 0) turns the queue off; knnx_coalesce_stats: batches served, queries in
 * them, the largest batch so far (any pointer may be NULL). */
int knnx_set_coalesce(knnx_index* ix, int on);
int knnx_coalesce_stats(knnx_index* ix, int64_t* batches, int64_t* queries, int64_t* largest_batch);

/* int8 first stage of the flat scans (round 4; no counterpart in the reference: faiss IndexFlatIP scans its one copy of the rows,
 * clip_back.py:362).  A flat (non-IVF) index of d = 512 / 768 / 1024 with at least 2^21 rows keeps, when the memory can be had,
 * an int8 copy of its fp16 rows (one scale per column) and scans THAT with 1 .. 256 queries per pass -- half the bytes, int8 MFMA --
 * to decide which rows are re-scored exactly from the fp16 rows; the admission threshold carries a proven bound of the quantisation
 * error (every fp32 norm in it widened by 1 + 1e-3, which covers its own rounding with an order of magnitude to spare), so D and I
 * are the exact top-k as without it (a query whose hit list overflows is re-run by the exact scan).  The copy is
 * built on the first search after the rows changed (one pass over the rows) and costs ntotal * d bytes; KNNX_I8=0 in the environment
 * turns it off.  When ntotal * d bytes cannot be had next to the rows (the headline shard: 125 M x 768 fp16 = 192 GB of a 288 GB part)
 * or KNNX_I8_MAX_BYTES caps it, the copy is PARTIAL: it holds as many leading rows as fit (at least a quarter of the index, else
 * none), those get the int8 first stage and the rows behind them are scanned in fp16 into the same hit lists -- one proof, one
 * result.  Any later allocation of the index that fails takes the copy's memory back and turns the feature off.
 * knnx_i8_served: queries answered through this path so far (-1: null index); knnx_i8_rows: rows the copy holds now (0: none). */
int64_t knnx_i8_served(knnx_index* ix);
int64_t knnx_i8_rows(knnx_index* ix);
/* 0: no int8 copy at the moment; 1 / 2: int8 planes per query.  A query is quantised as u = q * (column scales) with ONE scale, so an
 * index with a few columns much larger than the rest ("dominant": column scale > 3 x the median one; CLIP embeddings have them)
 * would leave the other components in the rounding error and admit (and re-score) two orders of magnitude more rows.  Decided at
 * every full build of the copy:
 *   no dominant column:  one plane
 *   1 .. 4 of them:      one plane; the copy keeps those columns in bytes 0..3 of each row (its layout is private), the matrix product
 *                        leaves them out and the scan adds their part with 14-bit query digits on the vector ALU (round 5) -- the
 *                        speed of one plane, 256 queries per pass.  knnx_i8_dominant: how many, and which columns (cols4: room for
 *                        4 ints, may be NULL); 0 when the index has none or the form is off (KNNX_I8_DOM=0 in the environment)
 *   more:                a second plane for what the first left -- twice the matrix work, four waves x 32 queries per pass
 * KNNX_I8_PLANES=1|2 forces the plane count (and switches the dominant-column form off). */
int knnx_i8_planes(knnx_index* ix);
int knnx_i8_dominant(knnx_index* ix, int* cols4);

/* One request of KnnService.knn_search with its dedup fused (clip_back.py:362 + :290-309): the top-k (k <= 64) of ONE query, and
 * the links of the reference's `get_non_uniques` -- every pair of result ranks (i < j) whose stored vectors, L2-normalised in
 * f32 as `normalized()` does (clip_back.py:194-197, :378), have inner product > dedup_thr (the reference: 0.94, strict).  The
 * k result rows are gathered ONCE on the device for the whole coalesced batch and the links of all its requests come from one
 * launch; they need not travel to the host (R_or_null = NULL).  pairs: int32 [2 * pairs_cap] = (i0, j0, i1, j1, ...), sorted;
 * *n_pairs = number of links found -- when it exceeds pairs_cap (or 512) only the first are stored and the caller should take
 * the general path (knnx_reconstruct + knnx_range_search_once).  Connected components over the links are the caller's. */
int knnx_search_dedup(knnx_index* ix, const float* q, int k, float* D, int64_t* I, float* R_or_null, float dedup_thr,
                      int32_t* pairs, int pairs_cap, int* n_pairs);

/* ---- post filter: the safety head on the GPU --------------------------------------------------------
 * Replaces `safety_model.predict(embeddings, batch_size)` of clip_retrieval/clip_back.py:315-325 for a model that is a stack
 * of fp32 Linear layers with ReLU between them -- the H14 detector of clip_retrieval/h14_nsfw_model.py:16-34
 * (1024 -> 1024 -> 2048 -> 1024 -> 256 -> 128 -> 16 -> 1, Dropout = identity in eval mode).  dims: n_layers + 1 widths;
 * weights[l]: f32 [dims[l+1], dims[l]] row-major (torch.nn.Linear.weight), biases[l]: f32 [dims[l+1]] or NULL;
 * relu[l] != 0: ReLU after layer l (NULL: after every layer but the last).  fp32 FMA, f32 accumulate in k order. */
typedef struct knnx_mlp knnx_mlp;
int knnx_mlp_create(int device, int n_layers, const int32_t* dims, const float* const* weights, const float* const* biases,
                    const uint8_t* relu, knnx_mlp** out);
/* x: host f32 [n, dims[0]] -> y: host f32 [n, dims[n_layers]]; synchronous; calls on one handle are serialised. */
int knnx_mlp_forward(knnx_mlp* m, const float* x_host, int n, float* y_host);
int knnx_mlp_destroy(knnx_mlp* m);

const char* knnx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* KNNX_H */
