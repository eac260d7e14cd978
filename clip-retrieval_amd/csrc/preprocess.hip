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
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/clipx.h"

extern "C" const char* clipx_last_error(void);
// (sets the thread-local message of clipx_api.hip)
extern "C" int clipx_set_error(int code, const char* msg);

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int PP_LDS_BYTES = 96 * 1024;  // LDS budget of a band's horizontally resampled rows (S = 224: 146 rows)
constexpr int PP_STAGE_BYTES = 48 * 1024;  // LDS budget of the staged source rows

struct ImgDesc {      // one per image, at the head of the coefficient buffer
  long long src_off;  // byte offset of the image in the packed source
  int h, w;           // decoded size
  int br;             // output rows per band
  int kh, kv;         // taps per output column / row (ksize of the two passes)
  int hb_off, hk_off; // int32 offsets (from the start of the coefficient buffer): horizontal bounds [S][2], weights [S][kh]
  int vb_off, vk_off; // vertical bounds [S][2], weights [S][kv]
  int c0, ncols;      // source columns the cropped output columns touch: [c0, c0 + ncols)
  int pitch, R;       // staging: bytes per staged source row (16-aligned, room for the alignment shift), rows per chunk
  int tmp_bytes;      // LDS bytes of the horizontally resampled rows of the largest band (16-aligned)
  int pad;
};

constexpr int STAGE_ROWS = 8;  // source rows resampled horizontally per pass: one weight load feeds STAGE_ROWS x 3 multiply-adds

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  // Opaque to the optimiser on purpose.  ROCm 7.2's LLVM fuses two of these shift-and-saturate results that are OR-ed into a
  // dword into one v_ashr_pk_u8_i32 and then treats the upper 16 bits of its destination as zero; on gfx950 the instruction
  // leaves them as they were, so bytes 2 and 3 of the packed dword picked up whatever the register held before (measured: the
  // filter weight).  Kept as separate v_med3 results, the pack is plain shifts and ORs.
  asm volatile("" : "+v"(v));
  return v;
}

// LDS: [tmp: horizontally resampled rows of the band, uint8 [rows][S][3]] [stg: STAGE_ROWS raw source rows] [vks: the band's
// vertical weights].  Source bytes reach the multiply-adds through LDS only: 16-byte global loads stage them, ds_read_u8 feeds
// the taps (a tap window is an arbitrary, unaligned byte range of the row).
__global__ __launch_bounds__(256) void resize_crop_kernel(const unsigned char* __restrict__ src, long long src_bytes,
                                                        const int* __restrict__ cb, const ImgDesc* __restrict__ descs, int S,
                                                        unsigned char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const ImgDesc d = descs[blockIdx.y];
  const int yy0 = blockIdx.x * d.br;
  if (yy0 >= S) return;
  const int yy1 = min(S, yy0 + d.br);
  const int tid = threadIdx.x;
  const int* hb = cb + d.hb_off;
  const int* hk = cb + d.hk_off;
  const int* vb = cb + d.vb_off;
  const int* vk = cb + d.vk_off;
  unsigned char* tmp = lds;
  unsigned char* stg = lds + d.tmp_bytes;
  int* vks = reinterpret_cast<int*>(stg + (size_t)d.R * d.pitch);
  const int r0 = vb[2 * yy0];                                   // first source row any window of this band touches
  const int r1 = vb[2 * (yy1 - 1)] + vb[2 * (yy1 - 1) + 1];     // one past the last (windows move monotonically)
  const int nrows = r1 - r0;
  for (int i = tid; i < (yy1 - yy0) * d.kv; i += 256) vks[i] = vk[yy0 * d.kv + i];
  const int rowbytes = S * 3;
  const int nvec = d.pitch >> 4;
  // ---- phase 1: horizontal pass of source rows [r0, r1) in chunks of R rows, output columns = the cropped S columns
  for (int rb = 0; rb < nrows; rb += d.R) {
    const int nr = min(d.R, nrows - rb);
    // stage: row r of the chunk = source bytes [a, a + ncols * 3) with a = src_off + ((r0 + rb + r) * w + c0) * 3, copied from the
    // 16-byte boundary below a (the shift a & 15 is re-derived by the readers); vectors that cross the ends of the packed source
    // are fetched byte by byte
    for (int i = tid; i < nr * nvec; i += 256) {
      const int r = i / nvec, v = i - r * nvec;
      const long long a = d.src_off + ((long long)(r0 + rb + r) * d.w + d.c0) * 3;
      const long long g = (a & ~15LL) + 16LL * v;
      uint4 val;
      if (g >= 0 && g + 16 <= src_bytes && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        val = *reinterpret_cast<const uint4*>(src + g);
      } else {
        unsigned char b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = (g + j >= 0 && g + j < src_bytes) ? src[g + j] : 0;
        val = *reinterpret_cast<const uint4*>(b);
      }
      *reinterpret_cast<uint4*>(stg + (size_t)r * d.pitch + 16 * v) = val;
    }
    __syncthreads();
    for (int xx = tid; xx < S; xx += 256) {
      const int xmin = hb[2 * xx], xmax = hb[2 * xx + 1];
      const int* k = hk + xx * d.kh;
      int acc[STAGE_ROWS][3];
      const unsigned char* p[STAGE_ROWS];
#pragma unroll
      for (int r = 0; r < STAGE_ROWS; ++r) {
        acc[r][0] = acc[r][1] = acc[r][2] = 1 << (PRECISION_BITS - 1);
        const long long a = d.src_off + ((long long)(r0 + rb + min(r, nr - 1)) * d.w + d.c0) * 3;
        p[r] = stg + (size_t)min(r, nr - 1) * d.pitch + (int)(a & 15) + (xmin - d.c0) * 3;
      }
      for (int x = 0; x < xmax; ++x) {
        const int kw = k[x];
#pragma unroll
        for (int r = 0; r < STAGE_ROWS; ++r) {
          acc[r][0] += (int)p[r][3 * x] * kw;
          acc[r][1] += (int)p[r][3 * x + 1] * kw;
          acc[r][2] += (int)p[r][3 * x + 2] * kw;
        }
      }
#pragma unroll
      for (int r = 0; r < STAGE_ROWS; ++r)
        if (r < nr) {
          unsigned char* t = tmp + ((size_t)(rb + r) * S + xx) * 3;
          t[0] = (unsigned char)clip8(acc[r][0]);
          t[1] = (unsigned char)clip8(acc[r][1]);
          t[2] = (unsigned char)clip8(acc[r][2]);
        }
    }
    __syncthreads();
  }
  // ---- phase 2: vertical pass over the LDS rows; four output bytes per item where the rows are whole dwords
  unsigned char* o = out + (size_t)blockIdx.y * S * rowbytes;
  if ((rowbytes & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
    const int n4 = rowbytes >> 2;
    for (int idx = tid; idx < (yy1 - yy0) * n4; idx += 256) {
      const int y = idx / n4, q = idx - y * n4;
      const int yy = yy0 + y;
      const int ymin = vb[2 * yy], ymax = vb[2 * yy + 1];
      const unsigned char* t = tmp + (size_t)(ymin - r0) * rowbytes + 4 * q;
      const int* k = vks + y * d.kv;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
      for (int j = 0; j < ymax; ++j) {
        const int kw = k[j];
        const unsigned wv = *reinterpret_cast<const unsigned*>(t);
        s0 += (int)(wv & 255u) * kw;
        s1 += (int)((wv >> 8) & 255u) * kw;
        s2 += (int)((wv >> 16) & 255u) * kw;
        s3 += (int)(wv >> 24) * kw;
        t += rowbytes;
      }
      const unsigned res = (unsigned)clip8(s0) | ((unsigned)clip8(s1) << 8) | ((unsigned)clip8(s2) << 16) | ((unsigned)clip8(s3) << 24);
      *reinterpret_cast<unsigned*>(o + (size_t)yy * rowbytes + 4 * q) = res;
    }
  } else {
    for (int idx = tid; idx < (yy1 - yy0) * rowbytes; idx += 256) {
      const int y = idx / rowbytes, q = idx - y * rowbytes;
      const int yy = yy0 + y;
      const int ymin = vb[2 * yy], ymax = vb[2 * yy + 1];
      const unsigned char* t = tmp + (size_t)(ymin - r0) * rowbytes + q;
      const int* k = vks + y * d.kv;
      int s0 = 1 << (PRECISION_BITS - 1);
      for (int j = 0; j < ymax; ++j) {
        s0 += (int)t[0] * k[j];
        t += rowbytes;
      }
      o[(size_t)yy * rowbytes + q] = (unsigned char)clip8(s0);
    }
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

// Datasets repeat sizes (img2dataset writes 256 x 256, or one side 256): the weights of an axis are kept per (in, out, first
// output coordinate, count).  A few KB each; the table is dropped when it reaches 4096 entries.
struct AxisKey {
  int in_size, out_size, o0, n;
  bool operator==(const AxisKey& o) const { return in_size == o.in_size && out_size == o.out_size && o0 == o.o0 && n == o.n; }
};
struct AxisKeyHash {
  size_t operator()(const AxisKey& k) const {
    uint64_t h = ((uint64_t)(uint32_t)k.in_size << 32) ^ (uint32_t)k.out_size;
    h = (h ^ ((uint64_t)(uint32_t)k.o0 << 20) ^ (uint64_t)(uint32_t)k.n) * 0x9E3779B97F4A7C15ull;
    return (size_t)(h ^ (h >> 29));
  }
};
struct AxisCoeffs {
  int ksize = 0;
  std::vector<int> bounds, kk;
};
std::mutex g_axis_mu;
std::unordered_map<AxisKey, std::shared_ptr<const AxisCoeffs>, AxisKeyHash> g_axis_cache;

std::shared_ptr<const AxisCoeffs> axis_coeffs_cached(int in_size, int out_size, int o0, int n) {
  const AxisKey key{in_size, out_size, o0, n};
  {
    std::lock_guard<std::mutex> lk(g_axis_mu);
    auto it = g_axis_cache.find(key);
    if (it != g_axis_cache.end()) return it->second;
  }
  auto c = std::make_shared<AxisCoeffs>();
  axis_coeffs(in_size, out_size, o0, n, &c->ksize, c->bounds, c->kk);
  std::lock_guard<std::mutex> lk(g_axis_mu);
  if (g_axis_cache.size() >= 4096) g_axis_cache.clear();
  g_axis_cache.emplace(key, c);
  return c;
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
  std::unordered_map<const AxisCoeffs*, std::pair<int, int>> placed;
  std::vector<std::shared_ptr<const AxisCoeffs>> held;  // (the cache may be cleared by another thread mid-call)
  int max_bands = 1;
  size_t max_lds = 0;
  long long src_bytes = 0;
  const int rows_max = PP_LDS_BYTES / (S * 3);
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
    const auto hc = axis_coeffs_cached(w, nw, left, S), vc = axis_coeffs_cached(h, nh, top, S);
    const std::vector<int>&hbv = hc->bounds, &hkv = hc->kk, &vbv = vc->bounds, &vkv = vc->kk;
    held.push_back(hc);
    held.push_back(vc);
    d.kh = hc->ksize;
    d.kv = vc->ksize;
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
    d.br = br;
    d.tmp_bytes = ((need * S * 3) + 15) & ~15;
    d.c0 = hbv[0];
    d.ncols = hbv[2 * (S - 1)] + hbv[2 * (S - 1) + 1] - d.c0;
    d.pitch = ((d.ncols * 3 + 15 + 15) / 16) * 16;  // the row starts up to 15 bytes into its first 16-byte vector
    if (d.pitch > PP_STAGE_BYTES) return fail(CLIPX_E_UNSUPPORTED, "image too wide for the GPU resample (one source row does not fit the LDS staging)");
    d.R = std::max(1, std::min(STAGE_ROWS, PP_STAGE_BYTES / d.pitch));
    d.pad = 0;
    max_lds = std::max(max_lds, (size_t)d.tmp_bytes + (size_t)d.R * d.pitch + (size_t)br * d.kv * 4);
    max_bands = std::max(max_bands, (S + br - 1) / br);
    src_bytes = std::max(src_bytes, (long long)offsets[i] + (long long)h * w * 3);
    if (offsets[i] < 0) return fail(CLIPX_E_ARG, "negative image offset");
    // one copy of an axis' tables per call, however many images of the batch share it
    auto place = [&](const std::shared_ptr<const AxisCoeffs>& c, int* b_off, int* k_off) {
      auto it = placed.find(c.get());
      if (it == placed.end()) {
        const int bo = (int)cb.size();
        cb.insert(cb.end(), c->bounds.begin(), c->bounds.end());
        const int ko = (int)cb.size();
        cb.insert(cb.end(), c->kk.begin(), c->kk.end());
        it = placed.emplace(c.get(), std::make_pair(bo, ko)).first;
      }
      *b_off = it->second.first;
      *k_off = it->second.second;
    };
    place(hc, &d.hb_off, &d.hk_off);
    place(vc, &d.vb_off, &d.vk_off);
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
    const size_t smem = max_lds;  // what the largest band of this batch needs (small for mild down-scales: more workgroups per CU)
    PPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_crop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(resize_crop_kernel, dim3(max_bands, B), dim3(256), smem, st, (const unsigned char*)src_dev, src_bytes,
                       (const int*)sl->dev, (const ImgDesc*)sl->dev, S, (unsigned char*)out_dev);
    PPCHK(hipGetLastError());
    PPCHK(hipEventRecord(sl->ev, st));
    sl->used = true;
  }
  return CLIPX_OK;
}
