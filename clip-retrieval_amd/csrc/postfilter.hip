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
