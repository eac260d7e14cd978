/*
 * clipx.h -- C ABI of the MI355X-native CLIP encoder (encode half of the hot path).
 *
 * This is the boundary a clip-retrieval maintainer binds (ctypes stub in INTEGRATION.md) to
 * replace the two model calls inside `ClipMapper.__call__`
 * (reference clip_retrieval/clip_inference/mapper.py:57 `self.model_img(...)`, :65
 * `self.model_txt(...)`) together with the normalise + fp16 cast that follows them
 * (mapper.py:58-59, 66-67), and the B=1 query encodes of `KnnService.compute_query`
 * (clip_back.py:230, 244).  Plain pointers and sizes only; no torch types.  Every function
 * returns 0 or a negative CLIPX_E_* code and never throws; clipx_last_error() is thread-local.
 *
 * Arithmetic: 16-bit MFMA operands (bf16; IEEE fp16 where the operand is the residual stream) with fp32 accumulation; the
 * residual stream is stored in fp16 and added to in fp32, LayerNorm statistics and softmax are fp32.  Outputs are unit-L2-norm
 * rows rounded to IEEE fp16, exactly the
 * `image_embs` / `text_embs` arrays the reference writer stores (writer.py:67-75).
 */
#ifndef CLIPX_H
#define CLIPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clipx_handle clipx_handle;

enum {
  CLIPX_OK = 0,
  CLIPX_E_ARG = -1,
  CLIPX_E_HIP = -2,
  CLIPX_E_NOMEM = -3,
  CLIPX_E_STATE = -4,
  CLIPX_E_UNSUPPORTED = -5,
  CLIPX_E_RANGE = -6 /* the fp16 residual stream overflowed for these weights / inputs: the call's embeddings are invalid */
};

#define CLIPX_ACT_QUICK_GELU 0 /* OpenAI CLIP checkpoints (x * sigmoid(1.702 x)) */
#define CLIPX_ACT_GELU 1       /* open_clip LAION checkpoints (erf GELU)         */

#define CLIPX_PIX_F32_NCHW 0 /* f32 [B,3,S,S], already mean/std normalised: the reference's item["image_tensor"] */
#define CLIPX_PIX_U8_NHWC 1  /* u8  [B,S,S,3] raw RGB; /255, mean/std normalised on the device               */

/* Architecture of one CLIP model = what `all_clip.load_clip(clip_model)` resolves a name to
 * (mapper.py:36-41).  Head dimension (width / heads) must be 64 or 80 (ViT-H/14) in this build. */
typedef struct clipx_model_desc {
  int image_size;   /* 224 */
  int patch_size;   /* 14 (L/14), 32 (B/32), 16 (B/16) */
  int v_width;      /* 1024 */
  int v_layers;     /* 24 */
  int v_heads;      /* 16 */
  int v_mlp;        /* 4096 */
  int ctx_len;      /* 77 */
  int vocab;        /* 49408 */
  int t_width;      /* 768 */
  int t_layers;     /* 12 */
  int t_heads;      /* 12 */
  int t_mlp;        /* 3072 */
  int embed_dim;    /* 768 */
  int act;          /* CLIPX_ACT_* */
  float ln_eps;     /* 1e-5 */
  float pix_mean[3];/* CLIP mean (0.48145466, 0.4578275, 0.40821073) -- used by CLIPX_PIX_U8_NHWC */
  float pix_std[3]; /* CLIP std  (0.26862954, 0.26130258, 0.27577711) */
} clipx_model_desc;

/* Number of floats in the weight blob for `desc` (see the order below). */
size_t clipx_blob_floats(const clipx_model_desc* desc);

/* Create an encoder on HIP device `device` from a host f32 weight blob.  Blob order, all
 * row-major, torch `nn.Linear` [out, in] weights:
 *   vision:  conv1.weight [v_width, 3*P*P]; class_embedding [v_width]; positional_embedding [T_v, v_width];
 *            ln_pre.{w,b}; per layer { ln_1.{w,b}; in_proj_weight [3w, w] (q|k|v); in_proj_bias [3w];
 *            out_proj.{weight [w,w], bias}; ln_2.{w,b}; c_fc.{weight [mlp,w], bias}; c_proj.{weight [w,mlp], bias} };
 *            ln_post.{w,b}; visual projection [embed_dim, v_width]
 *   text:    token_embedding [vocab, t_width]; positional_embedding [ctx_len, t_width]; per layer (same as above);
 *            ln_final.{w,b}; text projection [embed_dim, t_width]
 * (T_v = (image_size/patch_size)^2 + 1).  The blob is consumed during the call. */
int clipx_create(const clipx_model_desc* desc, const float* blob, size_t blob_floats, int device,
                 clipx_handle** out);
void clipx_destroy(clipx_handle* h);

/* model.encode_image(x) + `/= norm` + `.to(float16)` (mapper.py:57-59).  pixels: host pointer,
 * layout per pix_fmt; out_f16: host [B, embed_dim] IEEE fp16 bits.  Any B >= 1 (chunked internally). */
int clipx_encode_image(clipx_handle* h, const void* pixels, int B, int pix_fmt, uint16_t* out_f16);

/* model.encode_text(tokens) + normalise + fp16 (mapper.py:65-67).  ids: host int32 [B, ctx_len]
 * (SOT ... EOT 0 0 ..; the pooled position is argmax(ids) like the reference model). */
int clipx_encode_text(clipx_handle* h, const int32_t* ids, int B, uint16_t* out_f16);

/* Same encoders with an f32 [B, embed_dim] result: the unit-norm embedding BEFORE the fp16 rounding.  The query side
 * (`KnnService.compute_query`, clip_back.py:230-232, 244-246) hands fp32 features to the index. */
int clipx_encode_image_f32(clipx_handle* h, const void* pixels, int B, int pix_fmt, float* out_f32);
int clipx_encode_text_f32(clipx_handle* h, const int32_t* ids, int B, float* out_f32);

/* Asynchronous tickets: the call stages the batch (B <= clipx_max_batch()), enqueues upload, kernels and
 * download, and returns; clipx_wait() blocks until out_f16 is filled and releases the ticket.  The upload of ticket n+1
 * overlaps the kernels of ticket n, so a caller that submits batch n+1 before it waits for batch n (the Runner in
 * clip-retrieval_amd/runner.py) keeps the GPU busy across batches.  `pixels` / `ids` must stay valid until clipx_wait()
 * when they are page-locked (they are uploaded in place); pageable memory is copied during the call.  At most 4 tickets
 * can be outstanding per handle (CLIPX_E_STATE otherwise).  Every ticket must be waited for exactly once. */
typedef struct clipx_ticket clipx_ticket;
int clipx_encode_image_async(clipx_handle* h, const void* pixels, int B, int pix_fmt, uint16_t* out_f16, clipx_ticket** ticket);
int clipx_encode_text_async(clipx_handle* h, const int32_t* ids, int B, uint16_t* out_f16, clipx_ticket** ticket);
int clipx_wait(clipx_ticket* ticket);

/* Same with every buffer already resident in HBM (benchmark path, and callers that do their own
 * hipMemcpyAsync double-buffering).  `stream` is a hipStream_t (NULL = the handle's stream);
 * asynchronous on that stream.  out_f32_or_null: optional f32 [B, embed_dim] copy of the
 * normalised embedding before the fp16 rounding (parity tests measure cosine on it).  The handle's activation workspace
 * is shared by all calls: consecutive calls are ordered by an event even when they use different streams.
 * clipx_encode_text_device with B > 8 synchronises `stream` ONCE before its kernels are queued: it reads the token ids back
 * to find every caption's EOT position, and then runs the (causal) text tower on the rows up to the EOT only -- the same
 * embeddings, bit for bit (CLIPX_OPT_RAGGED_TEXT = 0 / environment CLIPX_RAGGED_TEXT=0: every row, and a fully asynchronous
 * call; a call on a `stream` that is being captured into a hipGraph takes that path by itself, with or without host ids).  Capturing
 * these calls into a caller's own hipGraph is allowed for B > 8 (smaller batches replay the library's own graphs); the replays
 * share the handle's activation workspace, so they must not run concurrently with other calls on the handle. */
int clipx_encode_image_device(clipx_handle* h, const void* pixels_dev, int B, int pix_fmt, uint16_t* out_f16_dev,
                              float* out_f32_or_null, void* stream);
int clipx_encode_text_device(clipx_handle* h, const int32_t* ids_dev, int B, uint16_t* out_f16_dev,
                             float* out_f32_or_null, void* stream);
/* The same for a caller that ALSO holds the ids on the host -- the reference's batch does: item["text_tokens"] is a CPU tensor the
 * reader tokenised (reader.py:163-170, mapper.py:63-65), uploaded by the caller for its own double-buffering.  ids_host [B, ctx_len]
 * must equal ids_dev and stay valid for the duration of the call; the EOT positions are taken from it, so nothing is read back and
 * the call never synchronises `stream`. */
int clipx_encode_text_device_ids(clipx_handle* h, const int32_t* ids_dev, const int32_t* ids_host, int B, uint16_t* out_f16_dev,
                                 float* out_f32_or_null, void* stream);

/* The geometric half of the reference's image transform on the GPU (reader.py:83,87 `self.image_transform(image)` = CLIP's
 * Resize(S, BICUBIC) + CenterCrop(S) on a PIL image), bit-identical to Pillow's 8-bit bicubic resample: B decoded RGB images of
 * any sizes, packed in one device buffer (image i = uint8 [hw[2i], hw[2i+1], 3] at byte offset offsets[i]), become the uint8
 * [B, S, S, 3] batch that clipx_encode_image_device takes as CLIPX_PIX_U8_NHWC.  offsets / hw are HOST arrays; the filter weights
 * are computed on the host in Pillow's arithmetic (a few KB per image), the pixel work runs on `stream` (asynchronous).
 * CLIPX_E_UNSUPPORTED for down-scales beyond ~30 x (a band of output rows no longer fits the LDS). */
int clipx_resize_crop_u8_device(int device, const void* src_dev, const int64_t* offsets, const int32_t* hw, int B, int S, void* out_dev,
                                void* stream);

/* Range guard.  The encoder keeps the residual stream of both towers in IEEE fp16: exact for every checkpoint that is
 * trained or served in fp16 (the reference's CUDA path runs the whole model in fp16, mapper.py:35-41), but a model whose
 * activations exceed 65 504 (possible for bf16-trained checkpoints) would overflow to inf, and LayerNorm turns a row holding inf
 * into a finite-looking WRONG embedding.  Every state of the stream is therefore checked on the device (one compare inside the
 * kernels that read it anyway) and an overflow is reported, never hidden: the host entry points (clipx_encode_image / _text /
 * _f32) and clipx_wait() return CLIPX_E_RANGE for a call in which any row overflowed (the output buffer of that call must be
 * discarded); after *_device calls, clipx_range_check(h, stream) synchronises `stream` and returns CLIPX_OK or CLIPX_E_RANGE for
 * everything queued through the *_device entry points since the previous check (it also clears the flag). */
int clipx_range_check(clipx_handle* h, void* stream);

/* Per-handle switches (defaults in brackets; the environment variables CLIPX_RAGGED_TEXT / CLIPX_FULL_LAST_BLOCK set the initial
 * values).  Both change which rows are computed, never the bytes of the embeddings (tests/test_clip_gpu.py).
 *   CLIPX_OPT_RAGGED_TEXT [1]      text batches > 8 run every layer on the rows up to each caption's EOT only
 *   CLIPX_OPT_POOL_LAST_BLOCK [1]  the last block runs past its attention on the pooled rows only
 * clipx_get_option returns the value or -1. */
#define CLIPX_OPT_RAGGED_TEXT 1
#define CLIPX_OPT_POOL_LAST_BLOCK 2
int clipx_set_option(clipx_handle* h, int option, int value);
int clipx_get_option(const clipx_handle* h, int option);

/* Largest batch one launch sequence handles (workspace is sized for it at create time;
 * CLIPX_MAX_BATCH env var, default 256).  Bigger B is processed in chunks of this size. */
int clipx_max_batch(const clipx_handle* h);
/* Number of small-batch launch sequences this handle has captured as hipGraphs so far (batches of <= 8 samples -- the B = 1
 * query encode of KnnService.compute_query, clip_back.py:207-255 -- are replayed from a graph per (tower, B, buffers, stream));
 * introspection for tests and service dashboards. */
int clipx_graphs_cached(const clipx_handle* h);
int clipx_embed_dim(const clipx_handle* h);

/* Raw bf16 GEMM of this library (out[m,n] = sum_k A[m,k] W[n,k] + bias[n]), device pointers;
 * exposed so tests and bench.py can check / time the dominant kernel in isolation.
 * epi: 0 bias->bf16, 1 bias+quick_gelu->bf16, 2 bias+gelu->bf16, 3 f32 out += acc+bias. */
int clipx_gemm_bf16_device(int device, const void* A_bf16, const void* W_bf16, const float* bias, void* out, int M,
                           int N, int K, int epi, void* stream);

/* The same with the two extras the encoder's LayerNorm-folded layers use: rowscale_or_null f32 [M] (epi 0..2:
 * out = act(acc * rowscale[m] + bias[n]); null = ones) and out16_or_null bf16 [M, N] (epi 3: also the bf16 rounding of the
 * new f32 rows -- the shadow of the residual stream the next folded GEMM reads). */
int clipx_gemm_bf16_ex_device(int device, const void* A_bf16, const void* W_bf16, const float* bias, void* out, int M, int N,
                              int K, int epi, const float* rowscale_or_null, void* out16_or_null, void* stream);
/* (the _ex entry point also takes epi 6: out is IEEE fp16 [M, N], updated in place, out = fp16(f32(out) + acc + bias) -- the
 * residual epilogue of out_proj / fc2 in the encoder, whose residual stream is stored in fp16.) */

/* The LayerNorm-folded GEMMs of the encoder (QKV, fc1) read that fp16 residual stream as their A operand: A and W are IEEE
 * fp16 (v_mfma_f32_32x32x16_f16, the same rate as bf16), out bf16 [M, N]; epi 0..2 and rowscale as above, or epi 7: epi 0 with
 * an IEEE fp16 output (the QKV projection: q and k carry three more mantissa bits into the softmax logits). */
int clipx_gemm_f16_device(int device, const void* A_f16, const void* W_f16, const float* bias, void* out_16bit, int M, int N, int K,
                          int epi, const float* rowscale_or_null, void* stream);

/* The same GEMM as the encoder's folded layers run it since round 6 (mapper.py:57,65 -> the LayerNorm in front of QKV / fc1): the row
 * scales are NOT an input but 1 / sqrt(var(A[m, :]) + eps) of the fp16 rows of A itself (K = the row length), computed inside the
 * 4-wave 256x256 kernel for the rows it multiplies and by the statistics pass into rstd_buf [M] (scratch; only those rows are
 * written) for the others -- the same bits either way (csrc/gemm_common.h: ln_rstd_onepass). */
int clipx_gemm_f16_ln_device(int device, const void* A_f16, const void* W_f16, const float* bias, void* out_16bit, int M, int N, int K,
                             int epi, float* rstd_buf, float eps, void* stream);

/* The attention and LayerNorm kernels in isolation (device pointers), for per-kernel parity tests:
 * qkv IEEE fp16 [B*T, 3*H*64] (q | k | v as the QKV projection's epi 7 writes them; the products run on fp16 MFMA, P is
 * rounded to fp16) -> out bf16 [B*T, H*64];  x f32 [M, d] -> y (bf16 if out_bf16 else f32). */
int clipx_attention_device(int device, const void* qkv_f16, void* out_bf16, int B, int T, int H, int causal,
                           void* stream);
/* Same for head dimension dh = 64 or 80 (ViT-H/14 image tower: 1280 / 16): qkv [B*T, 3*H*dh] -> out [B*T, H*dh]. */
int clipx_attention_dh_device(int device, const void* qkv_f16, void* out_bf16, int B, int T, int H, int dh,
                              int causal, void* stream);
int clipx_layernorm_device(int device, const float* x, const float* gamma, const float* beta, void* y, int out_bf16,
                           int M, int d, float eps, void* stream);
/* LayerNorm statistics of the residual stream as the LayerNorm-folded GEMMs consume them (per-kernel parity test):
 * rstd[m] = 1 / sqrt(var(x16[m, :]) + eps) by a pass over the 16-bit rows (is_f16: IEEE fp16, else bf16; is_f16 = 2: fp16 rows by the
 * one-pass canonical form that clipx_gemm_f16_ln_device uses for the rows its 4-wave kernel does not take). */
int clipx_rowstats_device(int device, const void* x16, int is_f16, float* rstd, int M, int d, float eps, void* stream);

/* Live per-kernel timing for bench.py: launches of the enabled kinds are bracketed by hipEvents on
 * their stream.  kind: 0 gemm, 1 attention, 2 layernorm, 3 other.  on: 0 off, 1 all kinds, else a bit
 * mask with bit (kind + 1): 2 = GEMMs only, 4 | 8 | 16 = everything but the GEMMs.  get() sums and resets. */
int clipx_profile_enable(clipx_handle* h, int on);
int clipx_profile_get(clipx_handle* h, int kind, int64_t* launches, double* ms, double* flops);

const char* clipx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CLIPX_H */
