// gemm_common.h -- device-side pieces shared by the two bf16 GEMM kernels (clip_kernels.hip: 128x128 tile,
// gemm256sp.hip: persistent 256x256).  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clip_kernels.h"

namespace clipx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int frag_t __attribute__((ext_vector_type(4)));  // one MFMA operand fragment: 8 x 16-bit values, type-agnostic

// one v_mfma_f32_32x32x16 on 16-bit operand fragments: bf16 (F16 = false) or IEEE fp16 (same issue rate)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(frag_t a, frag_t b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ---- MFMA shape (round 4).  CLIPX_MFMA16 = 1: every GEMM kernel of this library multiplies with v_mfma_f32_16x16x32 (16 x 16
// output blocks, 32-deep k-slabs) instead of v_mfma_f32_32x32x16.  Same flops per matrix-pipe cycle, same LDS bytes per flop -- but on
// this power-managed part the 16x16x32 form sustains ~10 % more TFLOP/s: +13 % on register-only loops, +9 % on this library's GEMM
// skeleton (fragment reads + LDS-DMA + barrier; tools/mfma_power_probe.hip, tools/mfma_probe2.hip, profiles/r04i_*, r04j_*): a quarter
// of the accumulator registers are read and written per instruction for half the flops.  It is also the instruction the vendor's
// assembly kernel uses.  All kernels switch together: an output element is the sum of its k-slabs in ascending order of
// 32, whichever kernel computes it, so rows stay bit-identical across kernels, batch chunkings and the ragged tail.
//
// Accumulator layout.  A 32 x 32 output block (32 weight rows n x 32 activation rows m, the weights being the MFMA A operand) is 16
// registers per lane in both forms (one f32x16, or four f32x4 quads):
//   32x32x16  lane (l31 = lane & 31, hb = lane >> 5):  quad g = 0..3 -> m = l31,           n = 8 g + 4 hb + e
//   16x16x32  lane (l15 = lane & 15, q4 = lane >> 4):  quad g = 0..3 -> m = 16 (g >> 1) + l15, n = 16 (g & 1) + 4 q4 + e     (e = 0..3)
// i.e. quad g of the 16x16x32 form is the 16 x 16 sub-block (m-half g >> 1, n-half g & 1).  Fragments: one 16-B LDS read per lane in
// both forms -- 32x32x16: row l31, k-chunk 2 kk + hb of the 16-deep step kk; 16x16x32: row l15 (+ 16 per half), k-chunk 4 s + q4 of
// the 32-deep slab s.
#ifndef CLIPX_MFMA16
#define CLIPX_MFMA16 1
#endif

template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(frag_t a, frag_t b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// In the 16x16x32 form the four quads of a 32 x 32 block are four separate f32x4 variables (each MFMA then accumulates in place;
// packed into one f32x16 with shuffles the register allocator let the accumulators wander and spilled around tile boundaries).
// one 32 x 32 block over one 32-deep k-slab: wf[j2] = weight rows 16 j2 .. + 16, af[h2] = activation rows 16 h2 .. + 16
template <bool F16>
__device__ __forceinline__ void mfma_block16(f32x4 (&q)[4], frag_t wf0, frag_t wf1, frag_t af0, frag_t af1) {
  q[0] = mfma_16x16x32<F16>(wf0, af0, q[0]);
  q[1] = mfma_16x16x32<F16>(wf1, af0, q[1]);
  q[2] = mfma_16x16x32<F16>(wf0, af1, q[2]);
  q[3] = mfma_16x16x32<F16>(wf1, af1, q[3]);
}
// the (m, n) of quad g inside its 32 x 32 block for this lane (n = first of four consecutive columns)
__device__ __forceinline__ int quad_m(int g, int lane) { return CLIPX_MFMA16 ? 16 * (g >> 1) + (lane & 15) : (lane & 31); }
__device__ __forceinline__ int quad_n(int g, int lane) { return CLIPX_MFMA16 ? 16 * (g & 1) + 4 * (lane >> 4) : 8 * g + 4 * (lane >> 5); }

// ---- LayerNorm statistics of an fp16 row, one pass, in ONE canonical order (round 6).  The folded GEMMs (QKV, fc1) multiply their
// accumulators by the row's 1 / sqrt(var + eps).  The 4-wave 256x256 kernel takes the two sums it needs from the A fragments it
// holds anyway (gemm256w4.hip, STATS: v_dot2_f32_f16 in the MFMA shadow) instead of a separate pass over the stream (48 launches and
// 1.1 - 1.4 ms per ViT-L/14 step in rounds 2 - 5); rows that reach another kernel get them from rowstats_f16_kernel.  Both follow:
//   lane part q4 = 0..3 of a row:  S1[q4], S2[q4] accumulate the 16-B chunks c = q4, q4 + 4, q4 + 8, .. of the row in ascending
//   order, the four dwords of a chunk in ascending order, one v_dot2_f32_f16 each (x . 1 resp. x . x added to the running sum);
//   row sums = (S[q4] + S[q4 ^ 1]) + (S[q4 ^ 2] + S[q4 ^ 3])   (the xor-16 / xor-32 butterfly of lanes l15 + 16 q4);
//   rstd = ln_rstd_onepass(S1, S2, 1 / d, eps).
// Same instructions in the same order = the same bits whichever kernel computes a row.
__device__ __forceinline__ float ln_rstd_onepass(float s1, float s2, float inv_d, float eps) {
  const float mean = s1 * inv_d;
  const float ex2 = s2 * inv_d;
  float var = __builtin_fmaf(-mean, mean, ex2);
  var = var > 0.f ? var : 0.f;
  return 1.f / sqrtf(var + eps);
}
#define CLIPX_DOT2_SQ(acc, w) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(w))
#define CLIPX_DOT2_SUM(acc, w, ones) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "s"(ones))

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// v * sigmoid(1.702 v) on the hardware exp2 and reciprocal (1 ulp each): 5 VALU instructions instead of the ~14 of
// exp + IEEE division -- the activation epilogue of fc1 is VALU time the matrix pipe waits for (128 outputs per lane).
__device__ __forceinline__ float quick_gelu(float v) {
  return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}
// exact-erf GELU (open_clip LAION towers): erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below the bf16 step of
// the output by four orders of magnitude) on v_rcp_f32 / v_exp_f32: ~14 VALU instructions instead of erff()'s ~30.
__device__ __forceinline__ float gelu_erf(float v) {
  const float z = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erf_abs = __builtin_fmaf(-p * t, e, 1.f);  // erf(|v| / sqrt 2)
  return 0.5f * v + 0.5f * fabsf(v) * erf_abs;           // 0.5 v (1 + sign(v) erf(|v|/sqrt 2))
}

// One accumulator quad of the transposed-product layout both kernels use (MFMA A operand = weight rows, B operand
// = activation rows): the lane owns output row m and four consecutive columns n..n+3.
// bf16-output epilogues: out = act(acc * rowscale[m] + bias[n]) -- rowscale is the row's LayerNorm 1/std when the GEMM
// consumes the raw residual stream with the LayerNorm folded into its weights (clipx_api.hip: fold_layernorm), 1 otherwise.
// The multiply-add is ONE fma in both kernels, so a row's result does not depend on which kernel produced it.
// f32 residual epilogue: x += acc + bias, and (out16 != null) the bf16 copy of the new x row that the next folded GEMM reads.
// fp16 residual epilogue (EPI_BIAS_RESID_H16): x16 = fp16(f32(x16) + (acc + bias)), in place -- the same association.
template <int EPI>
__device__ __forceinline__ void gemm_store_quad(float4 v, int m, int n, int N, const float* __restrict__ bias,
                                                void* __restrict__ outp, const float* __restrict__ table, int T,
                                                int row0, const float* __restrict__ rowscale, bf16* __restrict__ out16) {
  // OUT_BF16: the 16-bit-output epilogues (bf16, or IEEE fp16 for EPI_BIAS_F16)
  constexpr bool OUT_BF16 = EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_F16;
  if (EPI == EPI_RAW_F32) {  // a split-K partial product: the accumulators as they are
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * N + n) = v;
    return;
  }
  if (EPI != EPI_TABLE_F32) {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
    if (OUT_BF16) {
      const float r = rowscale[m];
      v.x = __builtin_fmaf(v.x, r, b4.x); v.y = __builtin_fmaf(v.y, r, b4.y);
      v.z = __builtin_fmaf(v.z, r, b4.z); v.w = __builtin_fmaf(v.w, r, b4.w);
    } else {
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    }
  }
  if (EPI == EPI_BIAS_QGELU_BF16) { v.x = quick_gelu(v.x); v.y = quick_gelu(v.y); v.z = quick_gelu(v.z); v.w = quick_gelu(v.w); }
  if (EPI == EPI_BIAS_GELU_BF16) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
  if (EPI == EPI_BIAS_F16) {
    // the fma above and this conversion stay two instructions (f32 result, then round to fp16): left alone, hipcc may merge them into
    // v_fma_mix{lo,hi}_f16 -- one rounding instead of two -- in one inlined copy and not in another, and a row would then depend on
    // the kernel that produced it (seen in round 4: 1 ulp on ~2e-5 of the outputs of the 32-row tail strips)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    f16x4 o;
    o[0] = (_Float16)v.x; o[1] = (_Float16)v.y; o[2] = (_Float16)v.z; o[3] = (_Float16)v.w;
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp) + (size_t)m * N + n) = o;
  } else if (OUT_BF16) {
    bf16x4 o;
    o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(outp) + (size_t)m * N + n) = o;
  } else if (EPI == EPI_BIAS_RESID_F32) {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * N + n);
    float4 o = *p;
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *p = o;
    if (out16) {
      bf16x4 h;
      h[0] = (bf16)o.x; h[1] = (bf16)o.y; h[2] = (bf16)o.z; h[3] = (bf16)o.w;
      *reinterpret_cast<bf16x4*>(out16 + (size_t)m * N + n) = h;
    }
  } else if (EPI == EPI_BIAS_RESID_H16) {
    f16x4* p = reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp) + (size_t)m * N + n);
    const f16x4 x4 = *p;
    f16x4 h;
    h[0] = (_Float16)((float)x4[0] + v.x); h[1] = (_Float16)((float)x4[1] + v.y);
    h[2] = (_Float16)((float)x4[2] + v.z); h[3] = (_Float16)((float)x4[3] + v.w);
    *p = h;
  } else {  // EPI_TABLE_F32
    const float4 t4 = *reinterpret_cast<const float4*>(table + (size_t)((m + row0) % T) * N + n);
    v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * N + n) = v;
  }
}

// bulk-tile launcher of gemm256sp.hip: rows [0, g.M) must be a multiple of 256, N % 256 == 0, K % 128 == 0
hipError_t launch_gemm256sp(const GemmArgs& g, int n_cu, hipStream_t st);
// the 4-wave form of the same tile (gemm256w4.hip): 16-bit-output epilogues and the fp16 in-place residual, K >= 256
bool gemm256w4_supports(const GemmArgs& g);
bool gemm256w4_fuses_stats(const GemmArgs& g);  // GemmArgs.stats_eps > 0 and the form has the STATS kernel
hipError_t launch_gemm256w4(const GemmArgs& g, int n_cu, hipStream_t st);

}  // namespace clipx
