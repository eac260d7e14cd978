// attn_bench.cpp -- times the library's attention kernel (clipx_attention_dh_device, include/clipx.h) without Python and
// checks a few (batch, head) pairs against an fp32 CPU softmax(QK^T/sqrt(dh))V of the same IEEE fp16 inputs (round 4: q, k, v are fp16; the output is bf16).
//   hipcc -O2 -o tools/attn_bench tools/attn_bench.cpp -ldl ;  tools/attn_bench [B T H dh causal]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
typedef int (*attn_fn)(int, const void*, void*, int, int, int, int, int, void*);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }
int main(int argc, char** argv) {
  std::string self = argv[0];
  std::string dir = self.substr(0, self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/'));
  void* h = dlopen(((dir.empty() ? std::string(".") : dir) + (getenv("CLIPX_LIB") ? std::string("/../clip-retrieval_amd/lib/") + getenv("CLIPX_LIB") : std::string("/../clip-retrieval_amd/lib/libclipx.so"))).c_str(), RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  attn_fn attn = (attn_fn)dlsym(h, "clipx_attention_dh_device");
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 257, H = argc > 3 ? atoi(argv[3]) : 16;
  const int dh = argc > 4 ? atoi(argv[4]) : 64, causal = argc > 5 ? atoi(argv[5]) : 0;
  const size_t ld = (size_t)3 * H * dh, nq = (size_t)B * T * ld, no = (size_t)B * T * H * dh;
  std::vector<uint16_t> hq(nq), ho(no);
  unsigned r = 777u;
  for (auto& v : hq) { r = r * 1664525u + 1013904223u; v = f2h((((int)(r >> 9) & 0xffff) / 32768.f - 1.f) * 1.5f); }
  void *dq, *dout;
  CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dout, no * 2));
  CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (attn(0, dq, dout, B, T, H, dh, causal, st)) { fprintf(stderr, "attention call failed\n"); return 2; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ho.data(), dout, no * 2, hipMemcpyDeviceToHost));
  double maxerr = 0;
  const int pairs[3][2] = {{0, 0}, {B / 2, H - 1}, {B - 1, H / 2}};
  std::vector<float> s(T);
  for (auto& ph : pairs) {
    const int b = ph[0], hh = ph[1];
    for (int q = 0; q < T; ++q) {
      const uint16_t* qp = &hq[((size_t)b * T + q) * ld + hh * dh];
      float mx = -1e30f;
      const int kend = causal ? q + 1 : T;
      for (int k = 0; k < kend; ++k) {
        const uint16_t* kp = &hq[((size_t)b * T + k) * ld + H * dh + hh * dh];
        float a = 0;
        for (int d = 0; d < dh; ++d) a += h2f(qp[d]) * h2f(kp[d]);
        s[k] = a / sqrtf((float)dh);
        mx = std::max(mx, s[k]);
      }
      double sum = 0;
      for (int k = 0; k < kend; ++k) { s[k] = expf(s[k] - mx); sum += s[k]; }
      for (int d = 0; d < dh; ++d) {
        double o = 0;
        for (int k = 0; k < kend; ++k) o += s[k] * h2f(hq[((size_t)b * T + k) * ld + 2 * H * dh + hh * dh + d]);
        o /= sum;
        maxerr = std::max(maxerr, fabs(o - bf2f(ho[((size_t)b * T + q) * (H * dh) + hh * dh + d])));
      }
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int i = 0; i < 12; ++i) {
    CK(hipEventRecord(e0, st)); attn(0, dq, dout, B, T, H, dh, causal, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double fl = 4.0 * B * H * (double)T * T * dh * (causal ? 0.5 : 1.0);
  printf("attention B=%d T=%d H=%d dh=%d causal=%d cfg=%s: median %.1f us (%.1f TFLOP/s), min %.1f us; max |err| vs fp32 CPU on 3 heads %.4g\n", B, T, H, dh,
         causal, getenv("CLIPX_ATTN_CFG") ? getenv("CLIPX_ATTN_CFG") : "-", ts[ts.size() / 2] * 1e3, fl / (ts[ts.size() / 2] * 1e-3) / 1e12, ts[0] * 1e3, maxerr);
  typedef int (*dbg_fn)(long long*, int);
  dbg_fn dbgf = (dbg_fn)dlsym(h, "clipx_dbg_attn_phase");
  if (dbgf && getenv("CLIPX_ATTN_CFG") && atoi(getenv("CLIPX_ATTN_CFG")) == 9) {
    const int nw = std::min(B * H, 2048) * 3;
    std::vector<long long> ph((size_t)nw * 4);
    if (!dbgf(ph.data(), nw * 4)) {
      double s4[4] = {0, 0, 0, 0};
      for (int i = 0; i < nw; ++i) for (int j = 0; j < 4; ++j) s4[j] += ph[(size_t)i * 4 + j] / (double)nw;
      printf("  phases, shader cycles per wave (3 query blocks): staging %.0f | S + max %.0f | exp + PV %.0f | store %.0f\n", s4[0], s4[1], s4[2], s4[3]);
    }
  }
  if (dbgf && getenv("CLIPX_ATTN_PK_TIMER") && atoi(getenv("CLIPX_ATTN_PK_TIMER"))) {
    // persistent kernel (attention_pk_kernel): per wave index, averaged over the workgroups, shader cycles of the LAST launch
    const int nwg = std::min(B * H, 256);
    std::vector<long long> ph((size_t)nwg * 6 * 4);
    if (!dbgf(ph.data(), nwg * 6 * 4)) {
      for (int w = 0; w < 6; ++w) {
        double s4[4] = {0, 0, 0, 0};
        for (int g = 0; g < nwg; ++g) for (int j = 0; j < 4; ++j) s4[j] += ph[((size_t)g * 6 + w) * 4 + j] / (double)nwg;
        const double pairs = (double)B * H / nwg;
        printf("  wave %d, shader cycles per pair: wait + barrier %.0f | DMA issue %.0f | S %.0f | exp + PV + out %.0f | sum %.0f\n", w,
               s4[0] / pairs, s4[1] / pairs, s4[2] / pairs, s4[3] / pairs, (s4[0] + s4[1] + s4[2] + s4[3]) / pairs);
      }
    }
  }
  return maxerr < 0.02 ? 0 : 1;
}
