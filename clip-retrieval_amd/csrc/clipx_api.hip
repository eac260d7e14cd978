// clipx_api.hip -- host side of the C ABI declared in include/clipx.h (encode half of the hot path).
//
// Stands in for `model.encode_image` / `model.encode_text` + normalise + fp16 as called by
// ClipMapper.__call__ (reference clip_retrieval/clip_inference/mapper.py:49-78).  Owns the bf16/f32
// weights in HBM, one activation workspace sized for `max_batch`, a compute stream, a copy stream
// and two pinned staging slots (H2D of chunk n+1 overlaps the kernels of chunk n).
// There is deliberately no CPU compute path: without a gfx950 device every entry point fails.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/clipx.h"
#include "clip_kernels.h"

using namespace clipx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return fail(_e == hipErrorOutOfMemory ? CLIPX_E_NOMEM : CLIPX_E_HIP,                \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                     \
  } while (0)

extern "C" const char* clipx_last_error(void) { return g_err.c_str(); }
// used by preprocess.hip (same library, other translation unit): set the thread-local message
extern "C" int clipx_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }

namespace {

struct LayerW {
  const float *ln1_w, *ln1_b, *qkv_b, *out_b, *ln2_w, *ln2_b, *fc1_b, *fc2_b;  // into the f32 device blob
  bf16 *qkv_w, *out_w, *fc1_w, *fc2_w;                                          // bf16 copies
  // ln_1 is folded into the QKV projection and ln_2 into fc1 (fold_layernorm): qkv_w / fc1_w hold
  // fp16(W gamma - rowmean(W gamma)) (IEEE fp16 bits behind the bf16-typed pointer) and these are bias + W beta; the GEMM
  // reads the fp16 residual stream itself and its epilogue multiplies by the row's 1/std (launch_rowstats)
  float *qkv_c, *fc1_c;
};

struct Tower {
  int width = 0, layers = 0, heads = 0, mlp = 0, T = 0;
  std::vector<LayerW> L;
  const float *lnf_w = nullptr, *lnf_b = nullptr;  // ln_post / ln_final
  bf16* proj = nullptr;                            // [embed, width]
};

struct ProfEvent {
  hipEvent_t a, b;
  int kind;
  double flops;
};

}  // namespace

struct clipx_handle {
  clipx_model_desc desc{};
  int device = 0;
  int max_batch = 256;
  int host_chunk = 256;  // chunk of the host-buffer pipeline (H2D of chunk i+1 overlaps the kernels of chunk i); measured
                         // at B=256: chunks of 32/64/128/256 -> 92/65/58/56.5 ms, small chunks lose more GEMM efficiency than
                         // the overlap wins
  int gemm_variant = 6;  // 6: the 4-wave 256x256 kernel (gemm256w4.hip) where it has the form, the 8-wave one (variant 3) elsewhere
  int n_cu = 256;
  std::mutex mu;
  hipStream_t stream = nullptr, copy_stream = nullptr;

  float* blob_dev = nullptr;  // the whole f32 blob (LayerNorm params, biases, embeddings live here)
  std::vector<void*> owned;   // every other device allocation
  Tower vis, txt;
  int Kp = 0, PP3 = 0;
  bf16* conv_w = nullptr;       // [v_width, Kp]
  float* clspos = nullptr;      // [T_v, v_width]: row 0 = class + pos[0], row t = pos[t]
  const float *ln_pre_w = nullptr, *ln_pre_b = nullptr;
  const float *tok_emb = nullptr, *txt_pos = nullptr;
  float *mean_dev = nullptr;

  // activation workspace (shared by both towers)
  float* x = nullptr;      // f32 [rows, width]: the patch-embedding output in front of ln_pre (vision tower only)
  float* rstd = nullptr;   // [rows] LayerNorm 1/std of the current residual rows
  // xn = THE residual stream, IEEE fp16 [rows, width] (fp16 bits behind the bf16-typed pointer): read as the A operand of the
  // LayerNorm-folded GEMMs (QKV, fc1) and updated in place by the residual epilogues of out_proj / fc2.  Round 3: it replaced
  // an f32 stream + bf16 shadow (673 MB -> 269 MB moved per residual GEMM at ViT-L/14 bs 256; same accuracy: the stream has
  // 3 mantissa bits more than the bf16 operands it used to be rounded to).
  bf16 *xn = nullptr, *qkv = nullptr, *att = nullptr, *hbuf = nullptr, *patches = nullptr;
  bool single_query = false;   // the API call being served is ONE sample (set by the entry points, not per chunk: the last
                               // chunk of a 7-sample call is one sample too, and must equal its row of an unchunked call)
  float* splitk_ws = nullptr;  // partial products of the small-M split-K GEMMs (clip_kernels.hip)
  size_t splitk_ws_bytes = 0;

  // host hand-over: CLIPX_SLOTS staging slots (pinned in/out + device in/out); a slot carries one chunk from its upload to
  // the moment its result has been copied to the caller (synchronous calls pipeline their chunks through them; every
  // asynchronous ticket owns one)
  static constexpr int NSLOT = 4;
  void* pin_in[NSLOT] = {};
  void* pin_out[NSLOT] = {};   // f16 [max_batch, E] then f32 [max_batch, E]
  void* dev_in[NSLOT] = {};
  uint16_t* dev_out[NSLOT] = {};
  float* dev_out32[NSLOT] = {};
  bool slot_busy[NSLOT] = {};
  int slot_next = 0;
  size_t in_slot_bytes = 0, out_slot_bytes = 0;
  hipEvent_t ev_copied[NSLOT] = {}, ev_done[NSLOT] = {};
  // Range guard of the fp16 residual stream (clipx.h, CLIPX_E_RANGE): one device flag per staging slot + one for the *_device
  // entry points (index NSLOT), raised by rowstats_kernel / tail_proj_kernel when a row of the stream holds inf / NaN; read back
  // with the slot's results (page-locked copies below) or by clipx_range_check().
  int* range_flags = nullptr;       // device [NSLOT + 1]
  int* range_host = nullptr;        // page-locked [NSLOT + 1]
  int* cur_flag = nullptr;          // the flag the launch sequence being enqueued reports to
  // the activation workspace is shared by every call: a launch sequence first waits for the event the previous one
  // recorded, whichever stream that ran on (callers of the *_device entry points may bring their own streams)
  hipEvent_t ev_ws = nullptr;
  bool ev_ws_valid = false;

  // Small batches (the B = 1 query encode of KnnService.compute_query, clip_back.py:207-255) are ~170 dependent launches of
  // 5-15 us kernels: their launch sequence is captured once per (tower, B, buffers, stream) into a hipGraph and replayed.
  typedef std::tuple<int, int, int, const void*, const void*, const void*, hipStream_t> GraphKey;
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::map<GraphKey, int> graph_seen;  // a launch sequence is captured the SECOND time its key shows up: a caller that brings
                                       // fresh buffers on every call (new addresses) never pays for captures it cannot reuse
  bool graphs_on = true;
  bool pool_last_block = true;  // last block past the attention on the pooled rows only (CLIPX_FULL_LAST_BLOCK=1: all rows)
  // Ragged text tower (CLIPX_RAGGED_TEXT=0: off).  The text transformer is causal and the embedding is read at the EOT token:
  // rows after a caption's EOT influence nothing that is read, so batches above the hipGraph sizes run every layer on
  // sum(eot_i + 1) rows instead of B x ctx_len.  Per call: lens / offsets / row map / pooled rows, built on the host from the
  // token ids, in a ring of RG_SLOTS page-locked + device buffers (a slot is reused when the event of its upload has passed).
  static constexpr int RG_SLOTS = 4;
  // LayerNorm statistics of the folded GEMMs inside the 4-wave GEMM kernel instead of a pass of their own: CLIPX_LN_FUSED=1, OFF by
  // default -- measured 50.5 ms per ViT-L/14 step against 42.6 (profiles/r06o_ab_ln_fused.log): the 128 v_dot2_f32_f16 per K-tile do
  // not hide behind the MFMAs of a one-wave-per-SIMD kernel (~8 cycles of issue each), they cost 8 ms to save 1.2
  bool ln_fused = false;
  bool ragged_text = true;
  int* rg_host[RG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  int* rg_dev[RG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t rg_ev[RG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  bool rg_used[RG_SLOTS] = {false, false, false, false};
  int rg_next = 0;
  int32_t* rg_ids_host = nullptr;          // device-pointer calls: the ids come back through this page-locked buffer
  const int32_t* text_ids_host = nullptr;  // host-pointer calls: the caller's ids of the chunk in flight (set by slot_submit)
  int prof = 0;  // bit k set: launches of kind k (0 gemm, 1 attention, 2 layernorm, 3 other) are bracketed by hipEvents
  std::vector<ProfEvent> prof_events;
};

static int dev_alloc(clipx_handle* h, void** p, size_t bytes) {
  HIPCHK(hipMalloc(p, bytes));
  h->owned.push_back(*p);
  return 0;
}

extern "C" size_t clipx_blob_floats(const clipx_model_desc* d) {
  if (!d) return 0;
  auto layer = [](size_t w, size_t mlp) {
    return 2 * w + 3 * w * w + 3 * w + w * w + w + 2 * w + mlp * w + mlp + w * mlp + w;
  };
  const size_t g = d->image_size / d->patch_size, Tv = g * g + 1;
  const size_t w = d->v_width, tw = d->t_width, E = d->embed_dim;
  size_t n = w * 3 * d->patch_size * d->patch_size + w + Tv * w + 2 * w;
  n += (size_t)d->v_layers * layer(w, d->v_mlp) + 2 * w + E * w;
  n += (size_t)d->vocab * tw + (size_t)d->ctx_len * tw;
  n += (size_t)d->t_layers * layer(tw, d->t_mlp) + 2 * tw + E * tw;
  return n;
}

static int check_desc(const clipx_model_desc* d) {
  if (d->image_size <= 0 || d->patch_size <= 0 || d->image_size % d->patch_size) return fail(CLIPX_E_ARG, "image_size must be a multiple of patch_size");
  if (d->v_width % 256 || d->t_width % 256) return fail(CLIPX_E_UNSUPPORTED, "tower widths must be multiples of 256");
  if (d->v_width % d->v_heads || d->t_width % d->t_heads) return fail(CLIPX_E_ARG, "width must be a multiple of heads");
  const int vdh = d->v_width / d->v_heads, tdh = d->t_width / d->t_heads;
  if ((vdh != 64 && vdh != 80) || (tdh != 64 && tdh != 80))
    return fail(CLIPX_E_UNSUPPORTED, "this build has attention kernels for head dimensions 64 and 80 only");
  if (d->v_mlp % 128 || d->t_mlp % 128) return fail(CLIPX_E_UNSUPPORTED, "mlp widths must be multiples of 128");
  const int g = d->image_size / d->patch_size;
  if (g * g + 1 > 288 || d->ctx_len > 288) return fail(CLIPX_E_UNSUPPORTED, "sequence longer than 288 tokens");
  if (d->embed_dim <= 0 || d->embed_dim % 8) return fail(CLIPX_E_ARG, "embed_dim must be a multiple of 8");
  if (d->act != CLIPX_ACT_QUICK_GELU && d->act != CLIPX_ACT_GELU) return fail(CLIPX_E_ARG, "unknown activation");
  if (d->v_layers <= 0 || d->t_layers <= 0 || d->vocab <= 0 || d->ctx_len <= 0) return fail(CLIPX_E_ARG, "bad layer/vocab/context size");
  return 0;
}

// carve one transformer tower's per-layer parameters out of the device blob, converting matrices to bf16
static int carve_layers(clipx_handle* h, Tower& t, float*& p) {
  const size_t w = t.width, mlp = t.mlp;
  t.L.resize(t.layers);
  for (int l = 0; l < t.layers; ++l) {
    LayerW& L = t.L[l];
    L.ln1_w = p; p += w;
    L.ln1_b = p; p += w;
    const float* qkv_w32 = p; p += 3 * w * w;
    L.qkv_b = p; p += 3 * w;
    const float* out_w32 = p; p += w * w;
    L.out_b = p; p += w;
    L.ln2_w = p; p += w;
    L.ln2_b = p; p += w;
    const float* fc1_w32 = p; p += mlp * w;
    L.fc1_b = p; p += mlp;
    const float* fc2_w32 = p; p += w * mlp;
    L.fc2_b = p; p += w;
    int r;
    if ((r = dev_alloc(h, (void**)&L.qkv_w, 3 * w * w * sizeof(bf16)))) return r;
    if ((r = dev_alloc(h, (void**)&L.out_w, w * w * sizeof(bf16)))) return r;
    if ((r = dev_alloc(h, (void**)&L.fc1_w, mlp * w * sizeof(bf16)))) return r;
    if ((r = dev_alloc(h, (void**)&L.fc2_w, w * mlp * sizeof(bf16)))) return r;
    if ((r = dev_alloc(h, (void**)&L.qkv_c, 3 * w * sizeof(float)))) return r;
    if ((r = dev_alloc(h, (void**)&L.fc1_c, mlp * sizeof(float)))) return r;
    HIPCHK(launch_fold_layernorm(qkv_w32, L.ln1_w, L.ln1_b, L.qkv_b, L.qkv_w, L.qkv_c, (int)(3 * w), (int)w, h->stream, 1));
    HIPCHK(launch_f32_to_bf16(out_w32, L.out_w, (int64_t)(w * w), h->stream));
    HIPCHK(launch_fold_layernorm(fc1_w32, L.ln2_w, L.ln2_b, L.fc1_b, L.fc1_w, L.fc1_c, (int)mlp, (int)w, h->stream, 1));
    HIPCHK(launch_f32_to_bf16(fc2_w32, L.fc2_w, (int64_t)(w * mlp), h->stream));
  }
  return 0;
}

static int create_impl(clipx_handle* h, const float* blob, size_t blob_floats) {
  const clipx_model_desc& d = h->desc;
  const int g = d.image_size / d.patch_size;
  Tower& V = h->vis;
  Tower& X = h->txt;
  V.width = d.v_width; V.layers = d.v_layers; V.heads = d.v_heads; V.mlp = d.v_mlp; V.T = g * g + 1;
  X.width = d.t_width; X.layers = d.t_layers; X.heads = d.t_heads; X.mlp = d.t_mlp; X.T = d.ctx_len;
  h->PP3 = 3 * d.patch_size * d.patch_size;
  h->Kp = (h->PP3 + 63) / 64 * 64;
  const size_t E = d.embed_dim;

  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc((void**)&h->blob_dev, blob_floats * sizeof(float)));
  HIPCHK(hipMemcpy(h->blob_dev, blob, blob_floats * sizeof(float), hipMemcpyHostToDevice));

  int r;
  float* p = h->blob_dev;
  // ---- vision
  const float* conv32 = p; p += (size_t)V.width * h->PP3;
  const float* cls = p; p += V.width;
  const float* vpos = p; p += (size_t)V.T * V.width;
  h->ln_pre_w = p; p += V.width;
  h->ln_pre_b = p; p += V.width;
  if ((r = dev_alloc(h, (void**)&h->conv_w, (size_t)V.width * h->Kp * sizeof(bf16)))) return r;
  HIPCHK(launch_pad_rows_bf16(conv32, h->conv_w, V.width, h->PP3, h->Kp, h->stream));
  if ((r = dev_alloc(h, (void**)&h->clspos, (size_t)V.T * V.width * sizeof(float)))) return r;
  HIPCHK(hipMemcpyAsync(h->clspos, vpos, (size_t)V.T * V.width * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  {  // row 0 += class embedding (host arithmetic on a tiny row, then upload)
    const size_t off_cls = (size_t)(cls - h->blob_dev), off_pos = (size_t)(vpos - h->blob_dev);
    std::vector<float> row0(V.width);
    for (int i = 0; i < V.width; ++i) row0[i] = blob[off_cls + i] + blob[off_pos + i];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->clspos, row0.data(), V.width * sizeof(float), hipMemcpyHostToDevice));
  }
  if ((r = carve_layers(h, V, p))) return r;
  V.lnf_w = p; p += V.width;
  V.lnf_b = p; p += V.width;
  const float* vproj32 = p; p += E * V.width;
  if ((r = dev_alloc(h, (void**)&V.proj, E * V.width * sizeof(bf16)))) return r;
  HIPCHK(launch_f32_to_bf16(vproj32, V.proj, (int64_t)(E * V.width), h->stream));
  // ---- text
  h->tok_emb = p; p += (size_t)d.vocab * X.width;
  h->txt_pos = p; p += (size_t)X.T * X.width;
  if ((r = carve_layers(h, X, p))) return r;
  X.lnf_w = p; p += X.width;
  X.lnf_b = p; p += X.width;
  const float* tproj32 = p; p += E * X.width;
  if ((r = dev_alloc(h, (void**)&X.proj, E * X.width * sizeof(bf16)))) return r;
  HIPCHK(launch_f32_to_bf16(tproj32, X.proj, (int64_t)(E * X.width), h->stream));
  if ((size_t)(p - h->blob_dev) != blob_floats) return fail(CLIPX_E_ARG, "internal: blob carve does not match clipx_blob_floats");

  // ---- workspace
  const size_t Bm = h->max_batch;
  const size_t rowsV = Bm * V.T, rowsX = Bm * X.T;
  const size_t nx = std::max(rowsV * V.width, rowsX * X.width);
  const size_t nqkv = std::max(rowsV * 3 * V.width, rowsX * 3 * X.width);
  const size_t nh = std::max(rowsV * V.mlp, rowsX * X.mlp);
  if ((r = dev_alloc(h, (void**)&h->x, rowsV * V.width * sizeof(float)))) return r;
  if ((r = dev_alloc(h, (void**)&h->xn, nx * sizeof(bf16)))) return r;
  if ((r = dev_alloc(h, (void**)&h->rstd, std::max(rowsV, rowsX) * sizeof(float)))) return r;
  if ((r = dev_alloc(h, (void**)&h->qkv, nqkv * sizeof(bf16)))) return r;
  if ((r = dev_alloc(h, (void**)&h->att, nx * sizeof(bf16)))) return r;
  if ((r = dev_alloc(h, (void**)&h->hbuf, nh * sizeof(bf16)))) return r;
  if ((r = dev_alloc(h, (void**)&h->patches, rowsV * h->Kp * sizeof(bf16)))) return r;
  h->splitk_ws_bytes = (size_t)32 << 20;
  if ((r = dev_alloc(h, (void**)&h->splitk_ws, h->splitk_ws_bytes))) return r;

  // ---- host hand-over slots
  h->in_slot_bytes = std::max((size_t)Bm * 3 * d.image_size * d.image_size * sizeof(float), (size_t)Bm * d.ctx_len * sizeof(int32_t));
  h->out_slot_bytes = Bm * E * sizeof(uint16_t);
  for (int s = 0; s < clipx_handle::NSLOT; ++s) {
    HIPCHK(hipHostMalloc(&h->pin_in[s], h->in_slot_bytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&h->pin_out[s], h->out_slot_bytes * 3, hipHostMallocDefault));
    if ((r = dev_alloc(h, &h->dev_in[s], h->in_slot_bytes))) return r;
    if ((r = dev_alloc(h, (void**)&h->dev_out[s], h->out_slot_bytes))) return r;
    if ((r = dev_alloc(h, (void**)&h->dev_out32[s], h->out_slot_bytes * 2))) return r;
    HIPCHK(hipEventCreateWithFlags(&h->ev_copied[s], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_done[s], hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&h->ev_ws, hipEventDisableTiming));
  if ((r = dev_alloc(h, (void**)&h->range_flags, (clipx_handle::NSLOT + 1) * sizeof(int)))) return r;
  HIPCHK(hipMemsetAsync(h->range_flags, 0, (clipx_handle::NSLOT + 1) * sizeof(int), h->stream));
  HIPCHK(hipHostMalloc((void**)&h->range_host, (clipx_handle::NSLOT + 1) * sizeof(int), hipHostMallocDefault));
  memset(h->range_host, 0, (clipx_handle::NSLOT + 1) * sizeof(int));
  h->cur_flag = h->range_flags + clipx_handle::NSLOT;
  {
    const size_t rg_ints = (size_t)Bm * (X.T + 3);
    for (int i = 0; i < clipx_handle::RG_SLOTS; ++i) {
      HIPCHK(hipHostMalloc((void**)&h->rg_host[i], rg_ints * sizeof(int), hipHostMallocDefault));
      if ((r = dev_alloc(h, (void**)&h->rg_dev[i], rg_ints * sizeof(int)))) return r;
      HIPCHK(hipEventCreateWithFlags(&h->rg_ev[i], hipEventDisableTiming));
    }
    HIPCHK(hipHostMalloc((void**)&h->rg_ids_host, (size_t)Bm * X.T * sizeof(int32_t), hipHostMallocDefault));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int clipx_create(const clipx_model_desc* desc, const float* blob, size_t blob_floats, int device,
                            clipx_handle** out) {
  if (!desc || !blob || !out) return fail(CLIPX_E_ARG, "null argument");
  *out = nullptr;
  int r = check_desc(desc);
  if (r) return r;
  if (blob_floats != clipx_blob_floats(desc)) return fail(CLIPX_E_ARG, "weight blob has the wrong number of floats for this model description");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(CLIPX_E_ARG, "no such HIP device");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(CLIPX_E_UNSUPPORTED, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  clipx_handle* h = new clipx_handle();
  h->desc = *desc;
  h->device = device;
  const char* mb = getenv("CLIPX_MAX_BATCH");
  if (mb && atoi(mb) > 0) h->max_batch = atoi(mb);
  const char* hc = getenv("CLIPX_HOST_CHUNK");
  if (hc && atoi(hc) > 0) h->host_chunk = atoi(hc);
  h->host_chunk = std::min(h->host_chunk, h->max_batch);
  const char* gv = getenv("CLIPX_GEMM_VARIANT");
  if (gv) h->gemm_variant = std::min(6, std::max(0, atoi(gv)));  // 5: tools build only (falls back to 3 in the product)
  const char* lf = getenv("CLIPX_LN_FUSED");
  if (lf) h->ln_fused = lf[0] == '1';
  const char* rgt = getenv("CLIPX_RAGGED_TEXT");
  if (rgt && rgt[0] == '0') h->ragged_text = false;
  const char* fl = getenv("CLIPX_FULL_LAST_BLOCK");
  if (fl && atoi(fl) > 0) h->pool_last_block = false;
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  r = create_impl(h, blob, blob_floats);
  if (r) {
    std::string keep = g_err;
    clipx_destroy(h);
    g_err = keep;
    return r;
  }
  *out = h;
  return CLIPX_OK;
}

extern "C" void clipx_destroy(clipx_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  for (auto& e : h->prof_events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.second);
  for (void* p : h->owned) (void)hipFree(p);
  if (h->blob_dev) (void)hipFree(h->blob_dev);
  for (int s = 0; s < clipx_handle::NSLOT; ++s) {
    if (h->pin_in[s]) (void)hipHostFree(h->pin_in[s]);
    if (h->pin_out[s]) (void)hipHostFree(h->pin_out[s]);
    if (h->ev_copied[s]) (void)hipEventDestroy(h->ev_copied[s]);
    if (h->ev_done[s]) (void)hipEventDestroy(h->ev_done[s]);
  }
  if (h->ev_ws) (void)hipEventDestroy(h->ev_ws);
  for (int i = 0; i < clipx_handle::RG_SLOTS; ++i) {
    if (h->rg_host[i]) (void)hipHostFree(h->rg_host[i]);
    if (h->rg_ev[i]) (void)hipEventDestroy(h->rg_ev[i]);
  }
  if (h->rg_ids_host) (void)hipHostFree(h->rg_ids_host);
  if (h->range_host) (void)hipHostFree(h->range_host);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  delete h;
}

extern "C" int clipx_set_option(clipx_handle* h, int option, int value) {
  if (!h) return fail(CLIPX_E_ARG, "handle is null");
  std::lock_guard<std::mutex> lk(h->mu);
  switch (option) {
    case CLIPX_OPT_RAGGED_TEXT: h->ragged_text = value != 0; return CLIPX_OK;
    case CLIPX_OPT_POOL_LAST_BLOCK: h->pool_last_block = value != 0; return CLIPX_OK;
    default: return fail(CLIPX_E_ARG, "unknown option");
  }
}
extern "C" int clipx_get_option(const clipx_handle* h, int option) {
  if (!h) return -1;
  switch (option) {
    case CLIPX_OPT_RAGGED_TEXT: return h->ragged_text ? 1 : 0;
    case CLIPX_OPT_POOL_LAST_BLOCK: return h->pool_last_block ? 1 : 0;
    default: return -1;
  }
}
extern "C" int clipx_max_batch(const clipx_handle* h) { return h ? h->max_batch : 0; }
extern "C" int clipx_graphs_cached(const clipx_handle* h) { return h ? (int)h->graphs.size() : 0; }
extern "C" int clipx_embed_dim(const clipx_handle* h) { return h ? h->desc.embed_dim : 0; }

// ---------------------------------------------------------------------------------------------
// launch helpers with optional event bracketing
// ---------------------------------------------------------------------------------------------
struct ProfScope {
  clipx_handle* h;
  hipStream_t st;
  hipEvent_t a = nullptr, b = nullptr;
  int kind;
  double flops;
  ProfScope(clipx_handle* h_, hipStream_t st_, int kind_, double flops_) : h(h_), st(st_), kind(kind_), flops(flops_) {
    if (h->prof & (1 << kind)) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, st);
    }
  }
  ~ProfScope() {
    if (a && b) {
      (void)hipEventRecord(b, st);
      h->prof_events.push_back({a, b, kind, flops});
    }
  }
};

static int run_gemm(clipx_handle* h, hipStream_t st, const bf16* A, const bf16* W, const float* bias, void* out,
                    const float* table, int T, int M, int N, int K, int epi, const float* rowscale = nullptr, bool f16 = false,
                    float stats_eps = 0.f) {
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.table = table; g.T = T;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.variant = h->gemm_variant; g.n_cu = h->n_cu; g.row0 = 0;
  g.rowscale = rowscale; g.out16 = nullptr; g.f16 = f16 ? 1 : 0;
  // stats_eps > 0: a LayerNorm-folded GEMM that computes the row scales itself -- inside the 4-wave kernel for the rows it takes, by
  // launch_rowstats into `rowscale` for the others (launch_gemm decides)
  g.stats_eps = stats_eps; g.range_flag = stats_eps > 0.f ? h->cur_flag : nullptr;
  // split-K only on the single-query path (B == 1, KnnService.compute_query): every batch of two or more samples is computed
  // by the unsplit kernels, whose rows do not depend on the batch they travel in (bitwise); a B = 1 row differs from the same
  // sample inside a batch by f32 summation order only
  if (h->single_query) { g.splitk_ws = h->splitk_ws; g.splitk_ws_bytes = h->splitk_ws_bytes; }
  ProfScope ps(h, st, 0, 2.0 * M * (double)N * K);
  HIPCHK(launch_gemm(g, st));
  return 0;
}

// where run_layers leaves the pooled rows (fp16 [B, width]) when it pools the last block: behind the gathered attention rows
static const void* pooled_rows(const clipx_handle* h, int B, int width) {
  return reinterpret_cast<const char*>(h->x) + (((size_t)B * width * sizeof(bf16) + 255) & ~(size_t)255);
}

struct Ragged {  // device arrays of one ragged text batch (see clipx_handle::ragged_text)
  int M;                // rows in all: sum of the lengths
  const int* offs;      // [B] first row of sample b
  const int* lens;      // [B] rows of sample b (EOT position + 1)
  const int* rowmap;    // [M] compact row -> b * ctx_len + t
  const int* poolrows;  // [B] compact row of the EOT token
  double att_flops;     // 4 heads dh sum(len^2)
};

// Returns in *pooled whether the residual stream that leaves the last block is the compact [B, width] buffer of pooled rows
// (h->x reused) instead of h->xn [B * T, width].
static int run_layers(clipx_handle* h, hipStream_t st, const Tower& t, int B, int causal, const int32_t* ids, bool* pooled,
                      const Ragged* rg = nullptr) {
  const int M = rg ? rg->M : B * t.T, w = t.width;
  const float eps = h->desc.ln_eps;
  const int act = h->desc.act == CLIPX_ACT_QUICK_GELU ? EPI_BIAS_QGELU_BF16 : EPI_BIAS_GELU_BF16;
  // On entry h->xn holds the residual stream x in fp16 (written by ln_pre / the text embedding).  Per block:
  //   rstd = rowstats(x);    qkv = fp16((x @ Wqkv'^T) * rstd + c_qkv)      [= LN1(x) @ Wqkv^T + b: LayerNorm folded; fp16 MFMA, fp16 out]
  //   att = attention(qkv);  x = fp16(x + att @ Wout^T + b_out)            [in place, bf16 MFMA, f32 add]
  //   rstd = rowstats(x);    h = act((x @ Wfc1'^T) * rstd + c_fc1);  x = fp16(x + h @ Wfc2^T + b_fc2)
  // (LayerNorm partial statistics computed by the residual epilogues instead of the rowstats pass were built twice in round 3
  // and measured slower on the same box -- the extra epilogue work costs more GEMM time than the 134 MB re-read it saves:
  // the patch is in the repository's history, round 3)
  for (int l = 0; l < t.layers; ++l) {
    const LayerW& L = t.L[l];
    int r;
    // (LayerNorm statistics: the separate pass of rounds 2 - 5, or with CLIPX_LN_FUSED=1 inside the GEMM -- run_gemm's stats_eps)
    if (!h->ln_fused) { ProfScope ps(h, st, 2, 0); HIPCHK(launch_rowstats(h->xn, h->rstd, M, w, eps, st, 1, h->cur_flag)); }
    if ((r = run_gemm(h, st, h->xn, L.qkv_w, L.qkv_c, h->qkv, nullptr, 1, M, 3 * w, w, EPI_BIAS_F16, h->rstd, true, h->ln_fused ? eps : 0.f))) return r;
    // last block of the image tower: only token 0's attention row is read afterwards -> query block 0 only (same arithmetic)
    const bool pool_here = l == t.layers - 1 && h->pool_last_block && t.T > 1;
    const int q_blocks = pool_here && !ids ? 1 : 0;
    const double att_rows = q_blocks ? std::min(32, t.T) : t.T;
    { ProfScope ps(h, st, 1, rg ? rg->att_flops * t.heads * (w / t.heads) : 4.0 * B * t.heads * att_rows * t.T * (w / t.heads));
      HIPCHK(launch_attention(h->qkv, h->att, B, t.T, t.heads, w / t.heads, causal, st, q_blocks, rg ? rg->offs : nullptr, rg ? rg->lens : nullptr)); }
    if (pool_here) {
      // The embedding reads ONE row of this block's output per sample (token 0 / the EOT token: launch_tail), and past the
      // attention every operation of a block is row-wise: out-proj, both residual adds, LayerNorm 2 and the MLP run on those B
      // rows only.  Same kernels, and a GEMM row does not depend on the rows it travels with (bitwise): the embeddings are the
      // bytes the full block gives (tests/test_clip_gpu.py; CLIPX_FULL_LAST_BLOCK=1 runs the full block).
      bf16* attc = reinterpret_cast<bf16*>(h->x);
      void* xc = reinterpret_cast<char*>(h->x) + (((size_t)B * w * sizeof(bf16) + 255) & ~(size_t)255);
      { ProfScope ps(h, st, 3, 0); HIPCHK(launch_gather_pooled(h->att, h->xn, ids, attc, xc, B, t.T, w, st, rg ? rg->poolrows : nullptr)); }
      if ((r = run_gemm(h, st, attc, L.out_w, L.out_b, xc, nullptr, 1, B, w, w, EPI_BIAS_RESID_H16))) return r;
      if (!h->ln_fused) { ProfScope ps(h, st, 2, 0); HIPCHK(launch_rowstats(xc, h->rstd, B, w, eps, st, 1, h->cur_flag)); }
      if ((r = run_gemm(h, st, reinterpret_cast<const bf16*>(xc), L.fc1_w, L.fc1_c, h->hbuf, nullptr, 1, B, t.mlp, w, act, h->rstd, true, h->ln_fused ? eps : 0.f))) return r;
      if ((r = run_gemm(h, st, h->hbuf, L.fc2_w, L.fc2_b, xc, nullptr, 1, B, w, t.mlp, EPI_BIAS_RESID_H16))) return r;
      *pooled = true;
      return 0;
    }
    if ((r = run_gemm(h, st, h->att, L.out_w, L.out_b, h->xn, nullptr, 1, M, w, w, EPI_BIAS_RESID_H16))) return r;
    if (!h->ln_fused) { ProfScope ps(h, st, 2, 0); HIPCHK(launch_rowstats(h->xn, h->rstd, M, w, eps, st, 1, h->cur_flag)); }
    if ((r = run_gemm(h, st, h->xn, L.fc1_w, L.fc1_c, h->hbuf, nullptr, 1, M, t.mlp, w, act, h->rstd, true, h->ln_fused ? eps : 0.f))) return r;
    if ((r = run_gemm(h, st, h->hbuf, L.fc2_w, L.fc2_b, h->xn, nullptr, 1, M, w, t.mlp, EPI_BIAS_RESID_H16))) return r;
  }
  *pooled = false;
  return 0;
}

constexpr int GRAPH_MAX_B = 8;      // batches up to this size are replayed from a captured graph
constexpr size_t GRAPH_MAX = 64;    // graphs kept per handle (distinct buffer sets of callers that bring their own)

// Runs `body` (which only launches kernels on `st`) directly, or -- small batch, profiling off -- through a hipGraph captured
// from exactly that launch sequence the first time this (tower, B, buffers, stream) combination is seen.  Any capture /
// instantiate failure switches graphs off for the handle and runs the plain launches: same kernels, same results.
template <class F>
static int run_graphed(clipx_handle* h, hipStream_t st, const clipx_handle::GraphKey& key, int B, F&& body) {
  if (!h->graphs_on || h->prof || B > GRAPH_MAX_B) return body();
  auto it = h->graphs.find(key);
  if (it != h->graphs.end()) {
    HIPCHK(hipGraphLaunch(it->second, st));
    return 0;
  }
  if (h->graphs.size() >= GRAPH_MAX) return body();
  if (h->graph_seen.size() > 4 * GRAPH_MAX) h->graph_seen.clear();
  if (++h->graph_seen[key] < 2) return body();
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    h->graphs_on = false;
    return body();
  }
  const int r = body();
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(st, &g);
  hipGraphExec_t ex = nullptr;
  if (r == 0 && e == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(g);
    h->graphs[key] = ex;
    HIPCHK(hipGraphLaunch(ex, st));
    return 0;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  h->graphs_on = false;
  return body();  // nothing ran during the capture
}

static int vision_chunk_body(clipx_handle* h, hipStream_t st, const void* pix_dev, int B, int fmt, uint16_t* out_f16,
                             float* out_f32);
static int text_chunk_body(clipx_handle* h, hipStream_t st, const int32_t* ids_dev, int B, uint16_t* out_f16, float* out_f32);
static bool stream_is_capturing(hipStream_t st);

// one chunk (B <= max_batch), everything on the device, asynchronous on `st`
static int vision_chunk(clipx_handle* h, hipStream_t st, const void* pix_dev, int B, int fmt, uint16_t* out_f16,
                        float* out_f32) {
  return run_graphed(h, st, clipx_handle::GraphKey(h->single_query ? 16 : 0, B, fmt, pix_dev, out_f16, out_f32, st), B,
                     [&]() { return vision_chunk_body(h, st, pix_dev, B, fmt, out_f16, out_f32); });
}

static int text_chunk(clipx_handle* h, hipStream_t st, const int32_t* ids_dev, int B, uint16_t* out_f16, float* out_f32) {
  return run_graphed(h, st, clipx_handle::GraphKey(h->single_query ? 17 : 1, B, 0, ids_dev, out_f16, out_f32, st), B,
                     [&]() { return text_chunk_body(h, st, ids_dev, B, out_f16, out_f32); });
}

static int vision_chunk_body(clipx_handle* h, hipStream_t st, const void* pix_dev, int B, int fmt, uint16_t* out_f16,
                             float* out_f32) {
  const clipx_model_desc& d = h->desc;
  const Tower& V = h->vis;
  const int M = B * V.T;
  { ProfScope ps(h, st, 3, 0); HIPCHK(launch_im2col(pix_dev, fmt, B, d.image_size, d.patch_size, h->Kp, d.pix_mean, d.pix_std, h->patches, st)); }
  int r = run_gemm(h, st, h->patches, h->conv_w, nullptr, h->x, h->clspos, V.T, M, V.width, h->Kp, EPI_TABLE_F32);
  if (r) return r;
  { ProfScope ps(h, st, 2, 0); HIPCHK(launch_layernorm(h->x, h->ln_pre_w, h->ln_pre_b, h->xn, 2, M, V.width, d.ln_eps, st)); }
  bool pooled = false;
  if ((r = run_layers(h, st, V, B, 0, nullptr, &pooled))) return r;
  const void* xf = pooled ? pooled_rows(h, B, V.width) : h->xn;
  { ProfScope ps(h, st, 3, 0); HIPCHK(launch_tail(xf, nullptr, V.lnf_w, V.lnf_b, V.proj, out_f16, out_f32, reinterpret_cast<float*>(h->qkv), B, pooled ? 1 : V.T, V.width, d.embed_dim, d.ln_eps, st, 1, h->cur_flag)); }
  return 0;
}

// Builds the ragged description of a text batch from its ids (host) and uploads it.  The pooling rule is launch_tail's /
// gather_pooled's: the highest id, first occurrence (torch.argmax of the reference model).
static int ragged_prepare(clipx_handle* h, hipStream_t st, const int32_t* ids_host, int B, Ragged* rg) {
  const int T = h->txt.T;
  const int s = h->rg_next;
  h->rg_next = (s + 1) % clipx_handle::RG_SLOTS;
  if (h->rg_used[s]) HIPCHK(hipEventSynchronize(h->rg_ev[s]));  // the upload that last read this slot's host buffer is done
  int* host = h->rg_host[s];
  int* offs = host, *lens = host + B, *pool = host + 2 * B, *rowmap = host + 3 * B;
  int M = 0;
  double f = 0.0;
  for (int b = 0; b < B; ++b) {
    const int32_t* row = ids_host + (size_t)b * T;
    int best = 0, bv = row[0];
    for (int t = 1; t < T; ++t)
      if (row[t] > bv) { bv = row[t]; best = t; }
    const int len = best + 1;
    offs[b] = M;
    lens[b] = len;
    pool[b] = M + best;
    for (int t = 0; t < len; ++t) rowmap[M + t] = b * T + t;
    M += len;
    f += 4.0 * (double)len * len;
  }
  int* dev = h->rg_dev[s];
  HIPCHK(hipMemcpyAsync(dev, host, ((size_t)3 * B + M) * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(h->rg_ev[s], st));
  h->rg_used[s] = true;
  rg->M = M;
  rg->offs = dev;
  rg->lens = dev + B;
  rg->poolrows = dev + 2 * B;
  rg->rowmap = dev + 3 * B;
  rg->att_flops = f;
  return 0;
}

static int text_chunk_body(clipx_handle* h, hipStream_t st, const int32_t* ids_dev, int B, uint16_t* out_f16, float* out_f32) {
  const clipx_model_desc& d = h->desc;
  const Tower& X = h->txt;
  Ragged rgv{};
  const Ragged* rg = nullptr;
  bool ragged = h->ragged_text && h->pool_last_block && B > GRAPH_MAX_B && X.T <= 128 && X.width / X.heads == 64;
  // A stream that is being captured into a hipGraph: the read-back of device-resident ids needs a synchronisation,
  // and even with the caller's host ids the row map is per-call host data uploaded from a ring slot that later calls overwrite --
  // a replayed graph would upload whatever the slot holds then.  Captured calls run the rectangular tower (same bytes out).
  if (ragged && stream_is_capturing(st)) ragged = false;
  if (ragged) {
    const int32_t* ids_host = h->text_ids_host;
    if (!ids_host) {  // device-resident ids: one synchronisation of `st` to read them (CLIPX_RAGGED_TEXT=0 keeps the call asynchronous)
      HIPCHK(hipMemcpyAsync(h->rg_ids_host, ids_dev, (size_t)B * X.T * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      ids_host = h->rg_ids_host;
    }
    int rr = ragged_prepare(h, st, ids_host, B, &rgv);
    if (rr) return rr;
    rg = &rgv;
  }
  { ProfScope ps(h, st, 3, 0); HIPCHK(launch_text_embed(ids_dev, h->tok_emb, h->txt_pos, nullptr, B, X.T, X.width, d.vocab, st, h->xn, 1, rg ? rg->rowmap : nullptr, rg ? rg->M : 0)); }
  bool pooled = false;
  int r = run_layers(h, st, X, B, 1, ids_dev, &pooled, rg);
  if (r) return r;
  const void* xf = pooled ? pooled_rows(h, B, X.width) : h->xn;
  { ProfScope ps(h, st, 3, 0); HIPCHK(launch_tail(xf, pooled ? nullptr : ids_dev, X.lnf_w, X.lnf_b, X.proj, out_f16, out_f32, reinterpret_cast<float*>(h->qkv), B, pooled ? 1 : X.T, X.width, d.embed_dim, d.ln_eps, st, 1, h->cur_flag)); }
  return 0;
}

static bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}
// A caller that captures `st` into its own hipGraph: the workspace event must not be recorded inside the capture (an event
// recorded there cannot be waited for by a later, non-captured call).  The captured launches are ordered by the graph itself;
// the caller must not replay it concurrently with other calls on this handle (clipx.h).
static int ws_acquire(clipx_handle* h, hipStream_t st) {
  if (h->ev_ws_valid && !stream_is_capturing(st)) HIPCHK(hipStreamWaitEvent(st, h->ev_ws, 0));
  return 0;
}
static int ws_release(clipx_handle* h, hipStream_t st) {
  if (stream_is_capturing(st)) return 0;
  HIPCHK(hipEventRecord(h->ev_ws, st));
  h->ev_ws_valid = true;
  return 0;
}

static size_t pix_bytes_per_image(const clipx_model_desc& d, int fmt) {
  const size_t px = (size_t)3 * d.image_size * d.image_size;
  return fmt == CLIPX_PIX_F32_NCHW ? px * sizeof(float) : px;
}

extern "C" int clipx_encode_image_device(clipx_handle* h, const void* pixels_dev, int B, int pix_fmt, uint16_t* out_f16_dev,
                                         float* out_f32_or_null, void* stream) {
  if (!h || !pixels_dev || !out_f16_dev || B < 0) return fail(CLIPX_E_ARG, "bad encode_image arguments");
  if (pix_fmt != CLIPX_PIX_F32_NCHW && pix_fmt != CLIPX_PIX_U8_NHWC) return fail(CLIPX_E_ARG, "unknown pixel format");
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const size_t ib = pix_bytes_per_image(h->desc, pix_fmt), E = h->desc.embed_dim;
  h->single_query = B == 1;
  h->cur_flag = h->range_flags + clipx_handle::NSLOT;
  if (ws_acquire(h, st)) return CLIPX_E_HIP;
  for (int o = 0; o < B; o += h->max_batch) {
    const int nb = std::min(h->max_batch, B - o);
    int r = vision_chunk(h, st, (const char*)pixels_dev + (size_t)o * ib, nb, pix_fmt, out_f16_dev + (size_t)o * E,
                         out_f32_or_null ? out_f32_or_null + (size_t)o * E : nullptr);
    if (r) return r;
  }
  if (ws_release(h, st)) return CLIPX_E_HIP;
  return CLIPX_OK;
}

static int encode_text_device_impl(clipx_handle* h, const int32_t* ids_dev, const int32_t* ids_host_or_null, int B, uint16_t* out_f16_dev,
                                   float* out_f32_or_null, void* stream);

extern "C" int clipx_encode_text_device(clipx_handle* h, const int32_t* ids_dev, int B, uint16_t* out_f16_dev,
                                        float* out_f32_or_null, void* stream) {
  return encode_text_device_impl(h, ids_dev, nullptr, B, out_f16_dev, out_f32_or_null, stream);
}

extern "C" int clipx_encode_text_device_ids(clipx_handle* h, const int32_t* ids_dev, const int32_t* ids_host, int B, uint16_t* out_f16_dev,
                                            float* out_f32_or_null, void* stream) {
  if (!ids_host) return fail(CLIPX_E_ARG, "ids_host is null (use clipx_encode_text_device)");
  return encode_text_device_impl(h, ids_dev, ids_host, B, out_f16_dev, out_f32_or_null, stream);
}

static int encode_text_device_impl(clipx_handle* h, const int32_t* ids_dev, const int32_t* ids_host_or_null, int B, uint16_t* out_f16_dev,
                                   float* out_f32_or_null, void* stream) {
  if (!h || !ids_dev || !out_f16_dev || B < 0) return fail(CLIPX_E_ARG, "bad encode_text arguments");
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const size_t E = h->desc.embed_dim;
  h->single_query = B == 1;
  h->cur_flag = h->range_flags + clipx_handle::NSLOT;
  if (ws_acquire(h, st)) return CLIPX_E_HIP;
  for (int o = 0; o < B; o += h->max_batch) {
    const int nb = std::min(h->max_batch, B - o);
    // the caller's host copy of the ids (the reader tokenised them there): the ragged row map is built without reading them back
    h->text_ids_host = ids_host_or_null ? ids_host_or_null + (size_t)o * h->desc.ctx_len : nullptr;
    int r = text_chunk(h, st, ids_dev + (size_t)o * h->desc.ctx_len, nb, out_f16_dev + (size_t)o * E,
                       out_f32_or_null ? out_f32_or_null + (size_t)o * E : nullptr);
    h->text_ids_host = nullptr;
    if (r) return r;
  }
  if (ws_release(h, st)) return CLIPX_E_HIP;
  return CLIPX_OK;
}

static const char* kRangeMsg =
    "the residual stream left the range of IEEE fp16 (|x| > 65504 became inf) for at least one row of this batch: these weights "
    "cannot be served by the fp16-stream encoder; the embeddings written for this call must be discarded";

extern "C" int clipx_range_check(clipx_handle* h, void* stream) {
  if (!h) return fail(CLIPX_E_ARG, "handle is null");
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const int i = clipx_handle::NSLOT;
  HIPCHK(hipMemcpyAsync(h->range_host + i, h->range_flags + i, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemsetAsync(h->range_flags + i, 0, sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));
  if (h->range_host[i]) {
    h->range_host[i] = 0;
    return fail(CLIPX_E_RANGE, kRangeMsg);
  }
  return CLIPX_OK;
}

// Host-buffer path.  A chunk (<= host_chunk samples) travels through one staging slot: upload on the copy stream (straight
// from caller memory when that is already page-locked -- the reference's DataLoader collates with pin_memory, reader.py:198
// -- otherwise through the slot's pinned buffer), kernels + download on the compute stream, copy-out to the caller when
// the slot is collected.  The upload of a later chunk runs while the kernels of an earlier one execute; the activation
// workspace is shared, so the compute stream serialises the chunks.  Chunking never changes a row's result (test: bitwise).
struct clipx_ticket {
  clipx_handle* h;
  int slot;
  int nb;
  uint16_t* out16;
  float* out32;
};

enum { KIND_IMAGE = 0, KIND_TEXT = 1 };

// caller holds h->mu.  Returns the slot index (>= 0) or a negative error code.
static int slot_submit(clipx_handle* h, int kind, const char* src, size_t in_bytes_per_item, int nb, int pix_fmt, bool want32) {
  int s = -1;
  for (int i = 0; i < clipx_handle::NSLOT; ++i) {
    const int c = (h->slot_next + i) % clipx_handle::NSLOT;
    if (!h->slot_busy[c]) { s = c; break; }
  }
  if (s < 0) return fail(CLIPX_E_STATE, "all staging slots are in flight: clipx_wait() an earlier ticket first");
  h->slot_next = (s + 1) % clipx_handle::NSLOT;
  hipPointerAttribute_t attr;
  bool pinned = false;
  if (hipPointerGetAttributes(&attr, src) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
  else (void)hipGetLastError();  // plain malloc memory: the query fails, which is the answer
  if (!pinned) {
    memcpy(h->pin_in[s], src, (size_t)nb * in_bytes_per_item);
    src = (const char*)h->pin_in[s];
  }
  const size_t E = h->desc.embed_dim;
  HIPCHK(hipMemcpyAsync(h->dev_in[s], src, (size_t)nb * in_bytes_per_item, hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(hipEventRecord(h->ev_copied[s], h->copy_stream));
  HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copied[s], 0));
  int r = ws_acquire(h, h->stream);
  if (r) return r;
  float* o32 = want32 ? h->dev_out32[s] : nullptr;
  if (kind == KIND_TEXT) h->text_ids_host = reinterpret_cast<const int32_t*>(src);  // the ragged text tower reads the ids on the host
  h->cur_flag = h->range_flags + s;
  r = kind == KIND_IMAGE ? vision_chunk(h, h->stream, h->dev_in[s], nb, pix_fmt, h->dev_out[s], o32)
                         : text_chunk(h, h->stream, (const int32_t*)h->dev_in[s], nb, h->dev_out[s], o32);
  h->text_ids_host = nullptr;
  if (r) return r;
  if ((r = ws_release(h, h->stream))) return r;
  char* po = (char*)h->pin_out[s];
  HIPCHK(hipMemcpyAsync(po, h->dev_out[s], (size_t)nb * E * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
  if (want32) HIPCHK(hipMemcpyAsync(po + h->out_slot_bytes, o32, (size_t)nb * E * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->range_host + s, h->range_flags + s, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemsetAsync(h->range_flags + s, 0, sizeof(int), h->stream));
  HIPCHK(hipEventRecord(h->ev_done[s], h->stream));
  h->slot_busy[s] = true;
  return s;
}

// blocks until the slot's chunk is complete, copies it out, frees the slot.  Caller holds h->mu or owns the ticket.
static int slot_collect(clipx_handle* h, int s, int nb, uint16_t* out16, float* out32) {
  const size_t E = h->desc.embed_dim;
  hipError_t e = hipEventSynchronize(h->ev_done[s]);
  if (e == hipSuccess) {
    const char* po = (const char*)h->pin_out[s];
    if (out16) memcpy(out16, po, (size_t)nb * E * sizeof(uint16_t));
    if (out32) memcpy(out32, po + h->out_slot_bytes, (size_t)nb * E * sizeof(float));
  }
  h->slot_busy[s] = false;
  if (e != hipSuccess) return fail(CLIPX_E_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
  if (h->range_host[s]) {
    h->range_host[s] = 0;
    return fail(CLIPX_E_RANGE, kRangeMsg);
  }
  return 0;
}

// synchronous call: chunks pipelined two deep through the slots
static int host_sync(clipx_handle* h, int kind, const char* in, size_t in_bytes_per_item, int B, int pix_fmt, uint16_t* out16,
                     float* out32) {
  const size_t E = h->desc.embed_dim;
  const int CH = h->host_chunk;
  const int nchunk = (B + CH - 1) / CH;
  h->single_query = B == 1;
  int prev_slot = -1, prev_o = 0, prev_nb = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int o = c * CH, nb = std::min(CH, B - o);
    const int s = slot_submit(h, kind, in + (size_t)o * in_bytes_per_item, in_bytes_per_item, nb, pix_fmt, out32 != nullptr);
    if (s < 0) return s;
    if (prev_slot >= 0) {
      int r = slot_collect(h, prev_slot, prev_nb, out16 ? out16 + (size_t)prev_o * E : nullptr, out32 ? out32 + (size_t)prev_o * E : nullptr);
      if (r) return r;
    }
    prev_slot = s; prev_o = o; prev_nb = nb;
  }
  if (prev_slot >= 0)
    return slot_collect(h, prev_slot, prev_nb, out16 ? out16 + (size_t)prev_o * E : nullptr, out32 ? out32 + (size_t)prev_o * E : nullptr);
  return 0;
}

static int encode_image_host(clipx_handle* h, const void* pixels, int B, int pix_fmt, uint16_t* out16, float* out32) {
  if (!h || (B > 0 && (!pixels || (!out16 && !out32))) || B < 0) return fail(CLIPX_E_ARG, "bad encode_image arguments");
  if (pix_fmt != CLIPX_PIX_F32_NCHW && pix_fmt != CLIPX_PIX_U8_NHWC) return fail(CLIPX_E_ARG, "unknown pixel format");
  if (B == 0) return CLIPX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  return host_sync(h, KIND_IMAGE, (const char*)pixels, pix_bytes_per_image(h->desc, pix_fmt), B, pix_fmt, out16, out32);
}
static int encode_text_host(clipx_handle* h, const int32_t* ids, int B, uint16_t* out16, float* out32) {
  if (!h || (B > 0 && (!ids || (!out16 && !out32))) || B < 0) return fail(CLIPX_E_ARG, "bad encode_text arguments");
  if (B == 0) return CLIPX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  return host_sync(h, KIND_TEXT, (const char*)ids, (size_t)h->desc.ctx_len * sizeof(int32_t), B, 0, out16, out32);
}

extern "C" int clipx_encode_image(clipx_handle* h, const void* pixels, int B, int pix_fmt, uint16_t* out_f16) {
  return encode_image_host(h, pixels, B, pix_fmt, out_f16, nullptr);
}
extern "C" int clipx_encode_text(clipx_handle* h, const int32_t* ids, int B, uint16_t* out_f16) {
  return encode_text_host(h, ids, B, out_f16, nullptr);
}
extern "C" int clipx_encode_image_f32(clipx_handle* h, const void* pixels, int B, int pix_fmt, float* out_f32) {
  return encode_image_host(h, pixels, B, pix_fmt, nullptr, out_f32);
}
extern "C" int clipx_encode_text_f32(clipx_handle* h, const int32_t* ids, int B, float* out_f32) {
  return encode_text_host(h, ids, B, nullptr, out_f32);
}

static int encode_async(clipx_handle* h, int kind, const void* in, size_t in_bytes_per_item, int B, int pix_fmt, uint16_t* out_f16,
                        clipx_ticket** ticket) {
  if (!h || !in || !out_f16 || !ticket || B <= 0) return fail(CLIPX_E_ARG, "bad encode_*_async arguments");
  *ticket = nullptr;
  if (B > h->max_batch) return fail(CLIPX_E_ARG, "an asynchronous call takes at most clipx_max_batch() samples");
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  h->single_query = B == 1;
  const int s = slot_submit(h, kind, (const char*)in, in_bytes_per_item, B, pix_fmt, false);
  if (s < 0) return s;
  *ticket = new clipx_ticket{h, s, B, out_f16, nullptr};
  return CLIPX_OK;
}
extern "C" int clipx_encode_image_async(clipx_handle* h, const void* pixels, int B, int pix_fmt, uint16_t* out_f16,
                                        clipx_ticket** ticket) {
  if (pix_fmt != CLIPX_PIX_F32_NCHW && pix_fmt != CLIPX_PIX_U8_NHWC) return fail(CLIPX_E_ARG, "unknown pixel format");
  return encode_async(h, KIND_IMAGE, pixels, h ? pix_bytes_per_image(h->desc, pix_fmt) : 0, B, pix_fmt, out_f16, ticket);
}
extern "C" int clipx_encode_text_async(clipx_handle* h, const int32_t* ids, int B, uint16_t* out_f16, clipx_ticket** ticket) {
  return encode_async(h, KIND_TEXT, ids, h ? (size_t)h->desc.ctx_len * sizeof(int32_t) : 0, B, 0, out_f16, ticket);
}
extern "C" int clipx_wait(clipx_ticket* t) {
  if (!t) return fail(CLIPX_E_ARG, "ticket is null");
  clipx_handle* h = t->h;
  int r;
  {
    // the event wait runs outside the lock (other threads may submit meanwhile); the slot is this ticket's alone
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipEventSynchronize(h->ev_done[t->slot]);
    std::lock_guard<std::mutex> lk(h->mu);
    r = e == hipSuccess ? slot_collect(h, t->slot, t->nb, t->out16, t->out32)
                        : (h->slot_busy[t->slot] = false, fail(CLIPX_E_HIP, std::string("clipx_wait: ") + hipGetErrorString(e)));
  }
  delete t;
  return r;
}

static int gemm_hook(int device, const void* A_bf16, const void* W_bf16, const float* bias, void* out, int M, int N, int K, int epi,
                     const float* rowscale, void* out16, void* stream, int f16 = 0, float stats_eps = 0.f) {
  if (!A_bf16 || !W_bf16 || !out || M <= 0 || N <= 0 || K <= 0) return fail(CLIPX_E_ARG, "bad gemm arguments");
  if (!((epi >= 0 && epi <= 3) || epi == EPI_BIAS_RESID_H16 || epi == EPI_BIAS_F16) || !bias) return fail(CLIPX_E_ARG, "epi must be 0..3, 6 or 7 and bias non-null");
  if (f16 && epi > 2 && epi != EPI_BIAS_F16) return fail(CLIPX_E_ARG, "fp16 operands go with the 16-bit-output epilogues 0..2 and 7 only");
  if (N % 128 || K % 64) return fail(CLIPX_E_UNSUPPORTED, "N must be a multiple of 128 and K of 64");
  HIPCHK(hipSetDevice(device));
  GemmArgs g{};
  g.A = (const bf16*)A_bf16; g.W = (const bf16*)W_bf16; g.bias = bias; g.out = out; g.table = nullptr; g.T = 1;
  g.M = M; g.N = N; g.K = K; g.epi = epi;
  g.rowscale = rowscale;
  g.out16 = epi == 3 ? (bf16*)out16 : nullptr;
  g.f16 = f16;
  g.stats_eps = stats_eps;
  if (stats_eps > 0.f && !g.rowscale) return fail(CLIPX_E_ARG, "a GEMM that owns its LayerNorm statistics needs the [M] row-scale buffer");
  if (!g.rowscale) {  // bf16-output epilogues scale rows (LayerNorm-folded GEMMs of the encoder); a plain GEMM uses ones
    static std::mutex ones_mu;
    static float* ones[64] = {};
    static int ones_n[64] = {};
    std::lock_guard<std::mutex> lk(ones_mu);
    if (device < 0 || device >= 64) return fail(CLIPX_E_ARG, "bad device");
    if (ones_n[device] < M) {
      if (ones[device]) (void)hipFree(ones[device]);
      ones[device] = nullptr;
      ones_n[device] = 0;
      HIPCHK(hipMalloc((void**)&ones[device], (size_t)M * sizeof(float)));
      HIPCHK(launch_fill_f32(ones[device], 1.f, M, (hipStream_t)stream));
      HIPCHK(hipStreamSynchronize((hipStream_t)stream));
      ones_n[device] = M;
    }
    g.rowscale = ones[device];
  }
  const char* gv = getenv("CLIPX_GEMM_VARIANT");
  g.variant = gv ? std::min(6, std::max(0, atoi(gv))) : 6;
  int ncu = 0;
  HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
  g.n_cu = ncu;
  g.row0 = 0;
  HIPCHK(launch_gemm(g, (hipStream_t)stream));
  return CLIPX_OK;
}

extern "C" int clipx_gemm_bf16_device(int device, const void* A_bf16, const void* W_bf16, const float* bias, void* out,
                                      int M, int N, int K, int epi, void* stream) {
  return gemm_hook(device, A_bf16, W_bf16, bias, out, M, N, K, epi, nullptr, nullptr, stream);
}
extern "C" int clipx_gemm_bf16_ex_device(int device, const void* A_bf16, const void* W_bf16, const float* bias, void* out, int M,
                                         int N, int K, int epi, const float* rowscale_or_null, void* out16_or_null, void* stream) {
  return gemm_hook(device, A_bf16, W_bf16, bias, out, M, N, K, epi, rowscale_or_null, out16_or_null, stream);
}

extern "C" int clipx_gemm_f16_device(int device, const void* A_f16, const void* W_f16, const float* bias, void* out_bf16, int M, int N,
                                     int K, int epi, const float* rowscale_or_null, void* stream) {
  return gemm_hook(device, A_f16, W_f16, bias, out_bf16, M, N, K, epi, rowscale_or_null, nullptr, stream, 1);
}

extern "C" int clipx_gemm_f16_ln_device(int device, const void* A_f16, const void* W_f16, const float* bias, void* out_16bit, int M, int N,
                                        int K, int epi, float* rstd_buf, float eps, void* stream) {
  if (!(eps > 0.f)) return fail(CLIPX_E_ARG, "eps must be positive");
  return gemm_hook(device, A_f16, W_f16, bias, out_16bit, M, N, K, epi, rstd_buf, nullptr, stream, 1, eps);
}

extern "C" int clipx_attention_device(int device, const void* qkv_bf16, void* out_bf16, int B, int T, int H, int causal,
                                      void* stream) {
  if (!qkv_bf16 || !out_bf16 || B <= 0 || T <= 0 || H <= 0) return fail(CLIPX_E_ARG, "bad attention arguments");
  if (T > 288) return fail(CLIPX_E_UNSUPPORTED, "sequence longer than 288 tokens");
  HIPCHK(hipSetDevice(device));
  HIPCHK(launch_attention((const bf16*)qkv_bf16, (bf16*)out_bf16, B, T, H, 64, causal, (hipStream_t)stream));
  return CLIPX_OK;
}

extern "C" int clipx_attention_dh_device(int device, const void* qkv_bf16, void* out_bf16, int B, int T, int H, int dh,
                                         int causal, void* stream) {
  if (!qkv_bf16 || !out_bf16 || B <= 0 || T <= 0 || H <= 0) return fail(CLIPX_E_ARG, "bad attention arguments");
  if (T > 288) return fail(CLIPX_E_UNSUPPORTED, "sequence longer than 288 tokens");
  if (dh != 64 && dh != 80) return fail(CLIPX_E_UNSUPPORTED, "head dimension must be 64 or 80");
  HIPCHK(hipSetDevice(device));
  hipError_t e = launch_attention((const bf16*)qkv_bf16, (bf16*)out_bf16, B, T, H, dh, causal, (hipStream_t)stream);
  if (e == hipErrorInvalidValue) return fail(CLIPX_E_UNSUPPORTED, "no attention kernel for this (T, head dimension)");
  HIPCHK(e);
  return CLIPX_OK;
}

extern "C" int clipx_layernorm_device(int device, const float* x, const float* gamma, const float* beta, void* y,
                                      int out_bf16, int M, int d, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || M <= 0) return fail(CLIPX_E_ARG, "bad layernorm arguments");
  if (d % 256 || d > 2048) return fail(CLIPX_E_UNSUPPORTED, "d must be a multiple of 256, <= 2048");
  HIPCHK(hipSetDevice(device));
  HIPCHK(launch_layernorm(x, gamma, beta, y, out_bf16, M, d, eps, (hipStream_t)stream));
  return CLIPX_OK;
}

extern "C" int clipx_rowstats_device(int device, const void* x16, int is_f16, float* rstd, int M, int d, float eps, void* stream) {
  if (!x16 || !rstd || M <= 0) return fail(CLIPX_E_ARG, "bad rowstats arguments");
  if (d % 256 || d > 2048) return fail(CLIPX_E_UNSUPPORTED, "d must be a multiple of 256, <= 2048");
  HIPCHK(hipSetDevice(device));
  HIPCHK(launch_rowstats(x16, rstd, M, d, eps, (hipStream_t)stream, is_f16 ? 1 : 0, nullptr, is_f16 == 2 ? 1 : 0));
  return CLIPX_OK;
}

extern "C" int clipx_profile_enable(clipx_handle* h, int on) {
  if (!h) return fail(CLIPX_E_ARG, "handle is null");
  std::lock_guard<std::mutex> lk(h->mu);
  h->prof = on == 1 ? 15 : ((on >> 1) & 15);  // 1 = every kind; otherwise bit (kind + 1) selects a kind
  return CLIPX_OK;
}

extern "C" int clipx_profile_get(clipx_handle* h, int kind, int64_t* launches, double* ms, double* flops) {
  if (!h) return fail(CLIPX_E_ARG, "handle is null");
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  int64_t n = 0;
  double t = 0.0, f = 0.0;
  std::vector<ProfEvent> keep;
  for (auto& e : h->prof_events) {
    if (e.kind != kind) {
      keep.push_back(e);
      continue;
    }
    HIPCHK(hipEventSynchronize(e.b));
    float dt = 0.f;
    HIPCHK(hipEventElapsedTime(&dt, e.a, e.b));
    t += dt;
    f += e.flops;
    ++n;
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  h->prof_events.swap(keep);
  if (launches) *launches = n;
  if (ms) *ms = t;
  if (flops) *flops = f;
  return CLIPX_OK;
}
