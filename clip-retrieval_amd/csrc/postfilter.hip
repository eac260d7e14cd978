// postfilter.hip -- the safety head of the serve path's post filter on the GPU.
//
// Stands in for `safety_model.predict(embeddings, batch_size)` of clip_retrieval/clip_back.py:315-325 when the model is the
// H14 detector (clip_retrieval/h14_nsfw_model.py:10-50): a stack of fp32 Linear layers with ReLU between them
// (1024 -> 1024 -> 2048 -> 1024 -> 256 -> 128 -> 16 -> 1; Dropout is the identity in eval mode), applied to the k result
// embeddings of one request.  Any Linear / ReLU stack is accepted.
//
// fp32 throughout -- the reference runs it in fp32 on the CPU and thresholds the raw output at 0.5, so no reduced-precision
// operand: plain FMA, f32 accumulate in k order (deterministic).  One kernel per layer: 64 x 64 output tile per workgroup,
// 4 x 4 outputs per thread, operands staged through LDS 16 k at a time; bias and ReLU fused.  33 GFLOP for 3 000 rows of the
// H14 stack: about a millisecond, against tens of milliseconds for torch on the host cores the request thread shares.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/knnx.h"

extern "C" int knnx_set_error(int code, const char* msg);

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// Y[n, N] = act(X[n, K] @ W[N, K]^T + b);  W row-major like torch.nn.Linear.weight
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ Y, int n, int N, int K,
                                                        int relu) {
  __shared__ float sX[TK][TM + 4];  // [k][row]: a thread reads 4 consecutive rows
  __shared__ float sW[TK][TN + 4];  // [k][col]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += TK) {
    // 64 x 16 elements of each operand: thread t loads element (r = t / 4 + 0, k = 4 (t % 4) .. +3) as one 16-byte vector
    {
      const int r = tid >> 2, kq = (tid & 3) * 4;
      const int gr = row0 + r, gc = col0 + r;
      // unconditional loads from clamped coordinates, masked afterwards: behind per-element tests hipcc waits for every load
      // on its own (the pattern found in the hit re-scoring kernel)
      float xv[4], wv[4];
      const size_t xr = (size_t)(gr < n ? gr : n - 1) * K, wr = (size_t)(gc < N ? gc : N - 1) * K;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + kq + e, kc = k < K ? k : K - 1;
        xv[e] = X[xr + kc];
        wv[e] = W[wr + kc];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool kin = k0 + kq + e < K;
        xv[e] = (gr < n && kin) ? xv[e] : 0.f;
        wv[e] = (gc < N && kin) ? wv[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sX[kq + e][r] = xv[e];
        sW[kq + e][r] = wv[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sX[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = sW[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], w[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = col0 + tx * 4 + j;
      if (c >= N) continue;
      float v = acc[i][j] + (b ? b[c] : 0.f);
      if (relu) v = v > 0.f ? v : 0.f;
      Y[(size_t)r * N + c] = v;
    }
  }
}

int fail(int code, const std::string& m) { return knnx_set_error(code, m.c_str()); }

}  // namespace

struct knnx_mlp {
  int device = 0;
  int n_layers = 0;
  std::vector<int> dims;           // n_layers + 1
  std::vector<float*> w, b;        // device
  std::vector<unsigned char> relu;
  float* act[2] = {nullptr, nullptr};  // ping-pong activations [cap, max_dim]
  size_t cap_rows = 0;
  int max_dim = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
};

#define MLPCHK(expr)                                                                                                              \
  do {                                                                                                                            \
    hipError_t _e = (expr);                                                                                                       \
    if (_e != hipSuccess) return fail(_e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" int knnx_mlp_destroy(knnx_mlp* m) {
  if (!m) return KNNX_OK;
  (void)hipSetDevice(m->device);
  for (float* p : m->w) (void)hipFree(p);
  for (float* p : m->b) (void)hipFree(p);
  (void)hipFree(m->act[0]);
  (void)hipFree(m->act[1]);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return KNNX_OK;
}

static int mlp_create_impl(knnx_mlp* m, const float* const* weights, const float* const* biases) {
  MLPCHK(hipSetDevice(m->device));
  MLPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  for (int l = 0; l < m->n_layers; ++l) {
    const size_t nw = (size_t)m->dims[l + 1] * m->dims[l];
    float* dw = nullptr;
    MLPCHK(hipMalloc(&dw, nw * sizeof(float)));
    m->w.push_back(dw);
    MLPCHK(hipMemcpy(dw, weights[l], nw * sizeof(float), hipMemcpyHostToDevice));
    float* db = nullptr;
    if (biases && biases[l]) {
      MLPCHK(hipMalloc(&db, (size_t)m->dims[l + 1] * sizeof(float)));
      m->b.push_back(db);
      MLPCHK(hipMemcpy(db, biases[l], (size_t)m->dims[l + 1] * sizeof(float), hipMemcpyHostToDevice));
    } else {
      m->b.push_back(nullptr);
    }
  }
  return KNNX_OK;
}

extern "C" int knnx_mlp_create(int device, int n_layers, const int32_t* dims, const float* const* weights, const float* const* biases,
                               const uint8_t* relu, knnx_mlp** out) {
  if (!out || !dims || !weights || n_layers <= 0 || n_layers > 64) return fail(KNNX_E_ARG, "bad mlp_create arguments");
  for (int l = 0; l <= n_layers; ++l)
    if (dims[l] <= 0 || dims[l] > (1 << 16)) return fail(KNNX_E_ARG, "mlp layer width out of range");
  for (int l = 0; l < n_layers; ++l)
    if (!weights[l]) return fail(KNNX_E_ARG, "null mlp weight");
  knnx_mlp* m = new knnx_mlp();
  m->device = device;
  m->n_layers = n_layers;
  m->dims.assign(dims, dims + n_layers + 1);
  m->relu.assign(n_layers, 0);
  for (int l = 0; l < n_layers; ++l) m->relu[l] = relu ? relu[l] : (l + 1 < n_layers ? 1 : 0);
  for (int l = 0; l <= n_layers; ++l) m->max_dim = std::max(m->max_dim, (int)dims[l]);
  const int r = mlp_create_impl(m, weights, biases);
  if (r) {
    knnx_mlp_destroy(m);
    return r;
  }
  *out = m;
  return KNNX_OK;
}

extern "C" int knnx_mlp_forward(knnx_mlp* m, const float* x_host, int n, float* y_host) {
  if (!m || n < 0 || (n > 0 && (!x_host || !y_host))) return fail(KNNX_E_ARG, "bad mlp_forward arguments");
  if (n == 0) return KNNX_OK;
  std::lock_guard<std::mutex> lk(m->mu);
  MLPCHK(hipSetDevice(m->device));
  if ((size_t)n > m->cap_rows) {
    (void)hipFree(m->act[0]);
    (void)hipFree(m->act[1]);
    m->act[0] = m->act[1] = nullptr;
    m->cap_rows = 0;
    const size_t rows = std::max<size_t>((size_t)n, 4096);
    MLPCHK(hipMalloc(&m->act[0], rows * m->max_dim * sizeof(float)));
    MLPCHK(hipMalloc(&m->act[1], rows * m->max_dim * sizeof(float)));
    m->cap_rows = rows;
  }
  MLPCHK(hipMemcpyAsync(m->act[0], x_host, (size_t)n * m->dims[0] * sizeof(float), hipMemcpyHostToDevice, m->stream));
  int cur = 0;
  for (int l = 0; l < m->n_layers; ++l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    hipLaunchKernelGGL(linear_f32_kernel, dim3((N + TN - 1) / TN, (n + TM - 1) / TM), dim3(256), 0, m->stream, m->act[cur], m->w[l],
                       m->b[l], m->act[cur ^ 1], n, N, K, (int)m->relu[l]);
    MLPCHK(hipGetLastError());
    cur ^= 1;
  }
  MLPCHK(hipMemcpyAsync(y_host, m->act[cur], (size_t)n * m->dims[m->n_layers] * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  MLPCHK(hipStreamSynchronize(m->stream));
  return KNNX_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Dedup links of a batch of requests (clip_back.py:290-309, fused into the coalesced search of knnx_api.hip).
// The reference, per request: faiss.IndexFlatIP over the k normalised result vectors, range_search(same vectors, 0.94), connected
// components.  Here the product: for request b (one workgroup), rows = its k reconstructed f32 vectors; sim(i, j) =
// dot(r_i, r_j) / (|r_i| |r_j|) in f32 (a wave per pair, lanes over d, fixed shuffle tree), every pair i < j with sim > thr is
// appended to the request's pair list.  k <= 64, so at most 2 016 pairs are examined per request: ~1.6 MFLOP at d = 1024.
// ids < 0 (the -1 padding of a short answer) take no part.  npairs counts every link, also past `cap` (the caller then
// falls back to the range scan for that request).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void dedup_pairs_kernel(const float* __restrict__ rows, const int64_t* __restrict__ ids, int k, int d,
                                                          float thr, const unsigned char* __restrict__ want, int32_t* __restrict__ pairs,
                                                          int cap, int* __restrict__ npairs) {
  const int b = blockIdx.x;
  if (want && !want[b]) return;
  __shared__ float inv_norm[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* R = rows + (size_t)b * k * d;
  for (int i = w; i < k; i += 4) {
    float s = 0.f;
    if (ids[(size_t)b * k + i] >= 0)
      for (int c = lane; c < d; c += 64) { const float v = R[(size_t)i * d + c]; s = __builtin_fmaf(v, v, s); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) inv_norm[i] = s > 0.f ? 1.f / sqrtf(s) : 0.f;  // a zero vector keeps norm 1 in the reference: its products are 0 either way
  }
  __syncthreads();
  const int np = k * (k - 1) / 2;
  for (int p = w; p < np; p += 4) {
    // pair index -> (i, j), i < j, row-major over the strict upper triangle
    int i = 0, rem = p;
    while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const float ni = inv_norm[i], nj = inv_norm[j];
    if (ni == 0.f || nj == 0.f) continue;  // padding / zero rows (wave-uniform)
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s = __builtin_fmaf(R[(size_t)i * d + c] * ni, R[(size_t)j * d + c] * nj, s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && s > thr) {
      const int at = atomicAdd(npairs + b, 1);
      if (at < cap) pairs[(size_t)b * cap + at] = (i << 16) | j;
    }
  }
}
}  // namespace

// rows f32 [m, k, d] and ids [m, k] on the device; pairs [m, cap] (i << 16 | j), npairs [m] (zeroed by the caller)
hipError_t knnx_launch_dedup_pairs(const float* rows, const int64_t* ids, int m, int k, int d, float thr, const unsigned char* want_or_null,
                                   int32_t* pairs, int cap, int* npairs, hipStream_t st) {
  if (m <= 0) return hipSuccess;
  if (k < 1 || k > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dedup_pairs_kernel, dim3(m), dim3(256), 0, st, rows, ids, k, d, thr, want_or_null, pairs, cap, npairs);
  return hipGetLastError();
}
