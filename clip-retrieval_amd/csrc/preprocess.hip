// preprocess.hip -- the geometric half of CLIP's image transform on the GPU, bit-identical to Pillow.
//
// Stands in for what the reference's DataLoader workers do to every decoded image before the encoder sees it
// (clip_retrieval/clip_inference/reader.py:83,87 `self.image_transform(image)`; third party: CLIP's `_transform` =
// torchvision Resize(n_px, BICUBIC) + CenterCrop(n_px) on a PIL image, i.e. Pillow's ImagingResample, 8 bits per channel):
//   * geometry: shorter side -> S, the long side int(S * long / short); crop offsets int(round((dim - S) / 2));
//   * resample: per output coordinate a window [xmin, xmin + xmax) of source pixels with bicubic weights (a = -0.5, support 2
//     x the down-scale), computed in double precision, normalised, converted to fixed point with 22 fractional bits; the
//     horizontal pass writes uint8 ((sum + 2^21) >> 22, clipped), the vertical pass runs over that (Resample.c).
// The weights are computed on the host in the very arithmetic of Pillow's precompute_coeffs / normalize_coeffs_8bpc (a few KB
// per image); the pixel work -- integer multiply-adds over bytes -- runs here.  A pass Pillow skips (size unchanged) becomes the
// identity weight 1 << 22, which reproduces the byte exactly.  Only what the centre crop keeps is computed.
//
// Kernel: one workgroup per (band of output rows, image).  Phase 1 resamples the source rows the band's vertical windows need
// horizontally into LDS (uint8 [rows, S, 3]); phase 2 resamples those vertically and writes the uint8 NHWC crop that
// clipx_encode_image_device takes as CLIPX_PIX_U8_NHWC.  HBM traffic = the decoded source once + the crop once.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/clipx.h"

extern "C" const char* clipx_last_error(void);
// (sets the thread-local message of clipx_api.hip)
extern "C" int clipx_set_error(int code, const char* msg);

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int PP_LDS_BYTES = 96 * 1024;  // LDS budget of a band's horizontally resampled rows (S = 224: 146 rows)

struct ImgDesc {      // one per image, in the coefficient buffer
  long long src_off;  // byte offset of the image in the packed source
  int h, w;           // decoded size
  int br;             // output rows per band
  int kh, kv;         // taps per output column / row (ksize of the two passes)
  int hb_off, hk_off; // int32 offsets (from the start of the coefficient buffer): horizontal bounds [S][2], weights [S][kh]
  int vb_off, vk_off; // vertical bounds [S][2], weights [S][kv]
};

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_crop_kernel(const unsigned char* __restrict__ src, const int* __restrict__ cb,
                                                        const ImgDesc* __restrict__ descs, int S,
                                                        unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tmp[];  // [rows][S][3]
  const ImgDesc d = descs[blockIdx.y];
  const int yy0 = blockIdx.x * d.br;
  if (yy0 >= S) return;
  const int yy1 = min(S, yy0 + d.br);
  const int* hb = cb + d.hb_off;
  const int* hk = cb + d.hk_off;
  const int* vb = cb + d.vb_off;
  const int* vk = cb + d.vk_off;
  const int r0 = vb[2 * yy0];                                   // first source row any window of this band touches
  const int r1 = vb[2 * (yy1 - 1)] + vb[2 * (yy1 - 1) + 1];     // one past the last (windows move monotonically)
  const int nrows = r1 - r0;
  const unsigned char* img = src + d.src_off;
  // ---- phase 1: horizontal pass of rows [r0, r1), output columns = the cropped S columns
  for (int idx = threadIdx.x; idx < nrows * S; idx += 256) {
    const int row = idx / S, xx = idx - row * S;
    const int xmin = hb[2 * xx], xmax = hb[2 * xx + 1];
    const unsigned char* p = img + ((size_t)(r0 + row) * d.w + xmin) * 3;
    const int* k = hk + xx * d.kh;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int kw = k[x];
      s0 += (int)p[3 * x] * kw;
      s1 += (int)p[3 * x + 1] * kw;
      s2 += (int)p[3 * x + 2] * kw;
    }
    unsigned char* t = tmp + (size_t)idx * 3;
    t[0] = (unsigned char)clip8(s0);
    t[1] = (unsigned char)clip8(s1);
    t[2] = (unsigned char)clip8(s2);
  }
  __syncthreads();
  // ---- phase 2: vertical pass over the LDS rows
  unsigned char* o = out + (size_t)blockIdx.y * S * S * 3;
  for (int idx = threadIdx.x; idx < (yy1 - yy0) * S; idx += 256) {
    const int y = idx / S, xx = idx - y * S;
    const int yy = yy0 + y;
    const int ymin = vb[2 * yy], ymax = vb[2 * yy + 1];
    const unsigned char* t = tmp + ((size_t)(ymin - r0) * S + xx) * 3;
    const int* k = vk + yy * d.kv;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int j = 0; j < ymax; ++j) {
      const int kw = k[j];
      s0 += (int)t[0] * kw;
      s1 += (int)t[1] * kw;
      s2 += (int)t[2] * kw;
      t += (size_t)S * 3;
    }
    unsigned char* q = o + ((size_t)yy * S + xx) * 3;
    q[0] = (unsigned char)clip8(s0);
    q[1] = (unsigned char)clip8(s1);
    q[2] = (unsigned char)clip8(s2);
  }
}

// ---- host: Pillow's coefficient arithmetic (Resample.c: bicubic_filter, precompute_coeffs, normalize_coeffs_8bpc)
double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// weights of output coordinates [o0, o0 + n) of an axis resampled from in_size to out_size; identity when the sizes agree
void axis_coeffs(int in_size, int out_size, int o0, int n, int* ksize_out, std::vector<int>& bounds, std::vector<int>& kk) {
  if (in_size == out_size) {  // Pillow skips the pass: weight 1.0 on the pixel itself reproduces the byte
    *ksize_out = 1;
    bounds.resize((size_t)2 * n);
    kk.assign((size_t)n, 1 << PRECISION_BITS);
    for (int i = 0; i < n; ++i) { bounds[2 * i] = o0 + i; bounds[2 * i + 1] = 1; }
    return;
  }
  const double scale = (double)((float)in_size - 0.f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  *ksize_out = ksize;
  bounds.resize((size_t)2 * n);
  kk.assign((size_t)n * ksize, 0);
  std::vector<double> k((size_t)ksize);
  for (int i = 0; i < n; ++i) {
    const int xx = o0 + i;
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      kk[(size_t)i * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xmax;
  }
}

// device-side coefficient buffers: a small ring per device, each slot guarded by the event of the launch that last read it
struct CoefSlot {
  void* dev = nullptr;
  void* host = nullptr;  // page-locked staging of the same size: the upload is a true asynchronous copy
  size_t bytes = 0;
  hipEvent_t ev = nullptr;
  bool used = false;
};
struct CoefRing {
  CoefSlot slot[4];
  int next = 0;
};
std::mutex g_mu;
CoefRing g_ring[64];

int fail(int code, const std::string& m) { return clipx_set_error(code, m.c_str()); }

}  // namespace

#define PPCHK(expr)                                                                                                \
  do {                                                                                                             \
    hipError_t _e = (expr);                                                                                        \
    if (_e != hipSuccess) return fail(_e == hipErrorOutOfMemory ? CLIPX_E_NOMEM : CLIPX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" int clipx_resize_crop_u8_device(int device, const void* src_dev, const int64_t* offsets, const int32_t* hw, int B, int S,
                                           void* out_dev, void* stream) {
  if (!src_dev || !offsets || !hw || !out_dev || B < 0 || S <= 0 || S > 1024) return fail(CLIPX_E_ARG, "bad resize_crop arguments");
  if (B == 0) return CLIPX_OK;
  if (device < 0 || device >= 64) return fail(CLIPX_E_ARG, "bad device");
  // ---- geometry + weights of every image (host, Pillow's arithmetic)
  std::vector<int> cb;
  std::vector<ImgDesc> descs((size_t)B);
  static_assert(sizeof(ImgDesc) % 8 == 0, "descriptors are read as an array at the head of the buffer");
  const size_t desc_ints = (size_t)B * sizeof(ImgDesc) / 4;
  cb.resize(desc_ints);
  int max_bands = 1, max_rows = 1;
  const int rows_max = PP_LDS_BYTES / (S * 3);
  std::vector<int> hbv, hkv, vbv, vkv;
  for (int i = 0; i < B; ++i) {
    const int h = hw[2 * i], w = hw[2 * i + 1];
    if (h <= 0 || w <= 0) return fail(CLIPX_E_ARG, "image with a non-positive size");
    int nw, nh;
    if (w <= h) { nw = S; nh = (int)((double)S * h / w); }   // torchvision Resize(int): long side = int(S * long / short)
    else { nw = (int)((double)S * w / h); nh = S; }
    const int left = (int)nearbyint((nw - S) / 2.0), top = (int)nearbyint((nh - S) / 2.0);  // Python round(): half to even
    ImgDesc& d = descs[i];
    d.src_off = offsets[i];
    d.h = h;
    d.w = w;
    axis_coeffs(w, nw, left, S, &d.kh, hbv, hkv);
    axis_coeffs(h, nh, top, S, &d.kv, vbv, vkv);
    // rows per band: 16 where the band's vertical windows fit the LDS budget, fewer for large down-scales
    int br = 16, need = 0;
    for (;;) {
      need = 0;
      for (int y0 = 0; y0 < S; y0 += br) {
        const int y1 = std::min(S, y0 + br) - 1;
        need = std::max(need, vbv[2 * y1] + vbv[2 * y1 + 1] - vbv[2 * y0]);
      }
      if (need <= rows_max) break;
      if (br == 1) return fail(CLIPX_E_UNSUPPORTED, "down-scale too large for the GPU resample (one output row needs more source rows than fit the LDS)");
      br /= 2;
    }
    max_rows = std::max(max_rows, need);
    d.br = br;
    max_bands = std::max(max_bands, (S + br - 1) / br);
    d.hb_off = (int)cb.size(); cb.insert(cb.end(), hbv.begin(), hbv.end());
    d.hk_off = (int)cb.size(); cb.insert(cb.end(), hkv.begin(), hkv.end());
    d.vb_off = (int)cb.size(); cb.insert(cb.end(), vbv.begin(), vbv.end());
    d.vk_off = (int)cb.size(); cb.insert(cb.end(), vkv.begin(), vkv.end());
  }
  memcpy(cb.data(), descs.data(), (size_t)B * sizeof(ImgDesc));
  const size_t bytes = cb.size() * sizeof(int);

  PPCHK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  CoefSlot* sl;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    CoefRing& r = g_ring[device];
    sl = &r.slot[r.next];
    r.next = (r.next + 1) & 3;
    if (sl->used) PPCHK(hipEventSynchronize(sl->ev));  // the launch that last read this slot has finished
    if (!sl->ev) PPCHK(hipEventCreateWithFlags(&sl->ev, hipEventDisableTiming));
    if (sl->bytes < bytes) {
      if (sl->dev) (void)hipFree(sl->dev);
      if (sl->host) (void)hipHostFree(sl->host);
      sl->dev = sl->host = nullptr;
      sl->bytes = 0;
      const size_t want = std::max(bytes, (size_t)1 << 20);
      PPCHK(hipMalloc(&sl->dev, want));
      PPCHK(hipHostMalloc(&sl->host, want, hipHostMallocDefault));
      sl->bytes = want;
    }
    memcpy(sl->host, cb.data(), bytes);
    PPCHK(hipMemcpyAsync(sl->dev, sl->host, bytes, hipMemcpyHostToDevice, st));
    const size_t smem = (size_t)max_rows * S * 3;  // what the largest band of this batch needs (small for mild down-scales: more workgroups per CU)
    PPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_crop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(resize_crop_kernel, dim3(max_bands, B), dim3(256), smem, st, (const unsigned char*)src_dev, (const int*)sl->dev,
                       (const ImgDesc*)sl->dev, S, (unsigned char*)out_dev);
    PPCHK(hipGetLastError());
    PPCHK(hipEventRecord(sl->ev, st));
    sl->used = true;
  }
  return CLIPX_OK;
}
