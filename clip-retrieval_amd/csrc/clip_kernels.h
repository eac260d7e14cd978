// clip_kernels.h -- launchers of the gfx950 CLIP encoder kernels (internal; public ABI: include/clipx.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clipx {

typedef __bf16 bf16;

enum GemmEpilogue {
  EPI_BIAS_BF16 = 0,        // out bf16 [M,N] = acc + bias                       (QKV projection)
  EPI_BIAS_QGELU_BF16 = 1,  // out bf16 = quick_gelu(acc + bias)                 (fc1, OpenAI CLIP)
  EPI_BIAS_GELU_BF16 = 2,   // out bf16 = gelu_erf(acc + bias)                   (fc1, open_clip LAION)
  EPI_BIAS_RESID_F32 = 3,   // out f32 [M,N] += acc + bias   (in place residual) (out_proj, fc2)
  EPI_TABLE_F32 = 4,        // out f32 = acc + table[m % T][n]                   (patch embed + cls/pos)
  EPI_RAW_F32 = 5,          // internal (split-K): out f32 [M,N] = acc, no bias; the reduction kernel applies the epilogue
  EPI_BIAS_RESID_H16 = 6,   // out fp16 [M,N] = fp16(f32(out) + (acc + bias)): the encoder's residual stream lives in IEEE fp16
                            // (out_proj, fc2): a third of the bytes of the f32 read-modify-write + bf16 shadow of EPI 3
  EPI_BIAS_F16 = 7          // out IEEE fp16 [M,N] = acc * rowscale + bias: EPI 0 with three more mantissa bits.  The QKV projection
                            // (round 4): q and k feed the softmax logits, where the bf16 rounding of EPI 0 was the largest single
                            // error of the encoder on weights with large LayerNorm gains (tools/emulate_fp16_stream.py)
};

struct GemmArgs {
  const bf16* A;      // activations [M, K] row-major
  const bf16* W;      // weights     [N, K] row-major (torch Linear layout)
  const float* bias;  // [N] or null
  void* out;          // [M, N] bf16 or f32
  const float* table; // [T, N] (EPI_TABLE_F32)
  int T;
  int M, N, K;        // N % 128 == 0, K % 64 == 0
  int epi;
  int variant;        // 0 = 128x128 register-staged LDS fill, 1 = 128x128 global_load_lds, >= 2 (3 by convention, the
                      // default) = persistent 256x256 kernel (gemm256sp.hip) for the whole 256-row m-tiles, 128x128
                      // (variant 1) for the rest
  int n_cu;           // compute units of the device (variant 3 launches one workgroup per CU); 0 = 256
  int row0;           // EPI_TABLE_F32: table row = (row0 + m) % T (set by the launcher when it splits M)
  const float* rowscale;  // bf16-output epilogues: out = act(acc * rowscale[m] + bias[n]); [M] f32, never null (the LayerNorm
                          // 1/std of the row when the LayerNorm is folded into W, launch_rowstats; ones otherwise)
  bf16* out16;            // EPI_BIAS_RESID_F32: also store the new x row as bf16 here (null: do not)
  float* splitk_ws;       // device scratch for split-K partial products (null: never split), splitk_ws_bytes of it
  size_t splitk_ws_bytes;
  int tail_m0, tail_nb;   // set by launch_gemm for the persistent kernel: rows [tail_m0, tail_m0 + 256) beyond the whole
                          // m-tiles are computed inside the same launch, 32 x (tail_nb x 32) per workgroup (0: none)
  int f16;                // A and W hold IEEE fp16 instead of bf16 (v_mfma_f32_32x32x16_f16, same rate): the LayerNorm-folded
                          // GEMMs, whose A operand is the fp16 residual stream itself.  bf16-output epilogues (0..2) and RAW only.
  float stats_eps;        // > 0 (f16, 16-bit-output epilogues, K = the row length): the GEMM is LayerNorm-folded and rowscale is NOT an
                          // input -- it is the [M] buffer that receives 1 / sqrt(var(A[m, :]) + stats_eps) where a separate pass has to
                          // compute it (launch_gemm runs launch_rowstats on those rows); the rows the 4-wave 256x256 kernel takes get
                          // their statistics from the A fragments inside its K loop instead (gemm256w4.hip, STATS) and are not written
  int* range_flag;        // (stats_eps > 0) raised when a row of A holds inf / NaN (launch_rowstats)
};

// number of 256-row m-tiles variant 3 hands to the 256x256 kernel for an [M, N] output
int gemm256_bulk_mtiles(int M, int N, int n_cu);
// blocks of 32 columns per workgroup when ONE ragged m-tile of 256 rows rides in the persistent launch (0: not possible)
int gemm256_tail_blocks(int N, int K, int n_cu);

hipError_t launch_gemm(const GemmArgs& g, hipStream_t st);

// x f32 [M, d] -> y [M, d]; out_kind 0: f32, 1: bf16, 2: IEEE fp16 (ln_pre of the vision tower writes the fp16 residual
// stream); one wave per row; d % 256 == 0, d <= 2048.  y16 (f32 output only, may be null): also the bf16 rounding of y
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* y, int out_kind, int M,
                            int d, float eps, hipStream_t st, bf16* y16 = nullptr);

// LayerNorm statistics of the 16-bit residual stream rows (f16 != 0: IEEE fp16, else bf16): rstd[m] = 1 / sqrt(var(x16[m, :]) + eps)
// (two-pass, fp32; or, for fp16 rows on request, ONE pass in the canonical order of gemm_common.h ln_rstd_onepass -- the order the
// 4-wave GEMM kernel accumulates the same sums in from its A fragments, so that a row's rstd does not depend on which of the two
// computed it: the CLIPX_LN_FUSED=1 experiment of round 6).  The LayerNorm itself is folded into the GEMM that follows (weights scaled by gamma and row-centred, bias
// absorbs beta: clipx_api.hip fold_layernorm), whose epilogue multiplies by rstd[m].  One wave per row.
// canonical != 0 (fp16 rows): the one-pass form of a GEMM that owns its statistics (GemmArgs.stats_eps)
hipError_t launch_rowstats(const void* x16, float* rstd, int M, int d, float eps, hipStream_t st, int f16 = 0, int* range_flag = nullptr,
                           int canonical = 0);

// LayerNorm-folded weights: Wf[n, k] = r16(W[n,k] gamma[k] - mean_k(W[n,:] gamma)), cf[n] = bias[n] + sum_k beta[k] W[n,k];
// r16 = bf16 rounding, or IEEE fp16 when f16 != 0
hipError_t launch_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, bf16* Wf, float* cf,
                                 int N, int K, hipStream_t st, int f16 = 0);
hipError_t launch_fill_f32(float* p, float v, int64_t n, hipStream_t st);

// pixels -> bf16 patch matrix [B*T, Kp] (row b*T is the all-zero class-token row; k = c*P*P + iy*P + ix)
// fmt 0: f32 NCHW already normalised (the reference's `image_tensor`); fmt 1: u8 NHWC, normalised here
hipError_t launch_im2col(const void* pixels, int fmt, int B, int S, int P, int Kp, const float* mean,
                         const float* stdv, bf16* out, hipStream_t st);

// qkv bf16 [B*T, 3*H*dh] -> out bf16 [B*T, H*dh]; softmax(q k^T / sqrt(dh) [+ causal]) v, head dim dh = 64 or 80
// q_blocks > 0: only the first q_blocks 32-row query blocks are computed (rows past them are left untouched); 0 = all
// offs / lens (both or neither; head dim 64, T <= 128): ragged batch -- sample b owns rows offs[b] .. offs[b] + lens[b] - 1
hipError_t launch_attention(const bf16* qkv, bf16* out, int B, int T, int H, int dh, int causal, hipStream_t st, int q_blocks = 0,
                            const int* offs = nullptr, const int* lens = nullptr);

// ids int32 [B, T] -> tok_emb[id] + pos_emb[t] as x f32 [B*T, d] (may be null) and / or x16 [B*T, d] (may be null; bf16, or
// IEEE fp16 when x16_f16 != 0: the text tower's residual stream)
hipError_t launch_text_embed(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int T,
                             int d, int vocab, hipStream_t st, void* x16 = nullptr, int x16_f16 = 0, const int* rowmap = nullptr,
                             int nrows = 0);

// pooled row (CLS, or argmax(ids) for text) -> LayerNorm -> @ proj^T [E, d] -> / L2 norm -> fp16 [B, E]
// x: the residual stream, f32 [B*T, d] or (x_f16 != 0) IEEE fp16
// scratch: B * E floats of device memory (the un-normalised projection between the two kernels)
// pooled rows (token 0, or the EOT token of `ids`) of att [B*T, d] bf16 and x16 [B*T, d] fp16 -> attc / xc [B, d]
hipError_t launch_gather_pooled(const bf16* att, const void* x16, const int32_t* ids_or_null, bf16* attc, void* xc, int B, int T,
                                int d, hipStream_t st, const int* rows_or_null = nullptr);
hipError_t launch_tail(const void* x, const int32_t* ids_or_null, const float* gamma, const float* beta,
                       const bf16* proj, uint16_t* out_f16, float* out_f32_or_null, float* scratch, int B, int T, int d,
                       int E, float eps, hipStream_t st, int x_f16 = 0, int* range_flag = nullptr);

hipError_t launch_f32_to_bf16(const float* in, bf16* out, int64_t n, hipStream_t st);
// conv weight [width, 3*P*P] f32 -> bf16 [width, Kp] zero padded
hipError_t launch_pad_rows_bf16(const float* in, bf16* out, int rows, int k, int kp, hipStream_t st);

}  // namespace clipx
