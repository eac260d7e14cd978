// gemm_bench.cpp -- times the library's bf16 GEMM (clipx_gemm_bf16_device, include/clipx.h) without Python: a gpurun call
// with this binary costs ~20 s of box time instead of minutes, so kernel A/B runs are cheap.
//
//   hipcc -O2 -o tools/gemm_bench tools/gemm_bench.cpp -ldl
//   tools/gemm_bench [-r reps] M,N,K,epi [M,N,K,epi ...] -- variant[:dbg[:flags]] [variant[:dbg[:flags]] ...]
//   epi 0..3 as in include/clipx.h; 6 = fp16 in-place residual (the encoder's out_proj / fc2); 16 / 17 / 18 = epi 0 / 1 / 2 with
//   IEEE fp16 operands (clipx_gemm_f16_device: the encoder's QKV / fc1)
//
// Every configuration (CLIPX_GEMM_VARIANT / _DBG / _FLAGS, read by the launcher at each call) runs interleaved with the
// others; dbg == 0 results are compared bit-for-bit with variant 0 (the 128x128 kernel accumulates in the same order).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef int (*gemm_fn)(int, const void*, const void*, const float*, void*, int, int, int, int, void*);
typedef int (*gemm_ex_fn)(int, const void*, const void*, const float*, void*, int, int, int, int, const float*, void*, void*);
static gemm_ex_fn g_ex = nullptr;
static void* g_shadow = nullptr;  // GEMM_BENCH_SHADOW=1: epi 3 also writes the bf16 shadow (clipx_gemm_bf16_ex_device)
typedef int (*gemm_f16_fn)(int, const void*, const void*, const float*, void*, int, int, int, int, const float*, void*);
static gemm_f16_fn g_f16 = nullptr;
typedef const char* (*err_fn)(void);

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

struct Cfg {
  std::string name, v, d, f;
};

static void set_cfg(const Cfg& c) {
  setenv("CLIPX_GEMM_VARIANT", c.v.c_str(), 1);
  if (c.d.empty()) unsetenv("CLIPX_GEMM_DBG"); else setenv("CLIPX_GEMM_DBG", c.d.c_str(), 1);
  if (c.f.empty()) unsetenv("CLIPX_GEMM_FLAGS"); else setenv("CLIPX_GEMM_FLAGS", c.f.c_str(), 1);
}

int main(int argc, char** argv) {
  std::string self = argv[0];
  std::string dir = self.substr(0, self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/'));
  std::string lib = (dir.empty() ? std::string(".") : dir) + "/../clip-retrieval_amd/lib/" + (getenv("CLIPX_LIB") ? getenv("CLIPX_LIB") : "libclipx.so");  // CLIPX_LIB: A/B of two builds
  void* h = dlopen(lib.c_str(), RTLD_NOW);
  if (!h) {
    fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror());
    return 2;
  }
  gemm_fn gemm0 = (gemm_fn)dlsym(h, "clipx_gemm_bf16_device");
  g_ex = (gemm_ex_fn)dlsym(h, "clipx_gemm_bf16_ex_device");
  g_f16 = (gemm_f16_fn)dlsym(h, "clipx_gemm_f16_device");
  const bool want_shadow = getenv("GEMM_BENCH_SHADOW") && g_ex;
  auto gemm = [&](int dev, const void* A, const void* W, const float* b, void* o, int M, int N, int K, int epi, void* st) -> int {
    if (epi >= 16) return g_f16 ? g_f16(dev, A, W, b, o, M, N, K, epi - 16, nullptr, st) : -99;
    if (epi == 6) return g_ex ? g_ex(dev, A, W, b, o, M, N, K, 6, nullptr, nullptr, st) : -99;
    if (want_shadow && epi == 3 && g_shadow) return g_ex(dev, A, W, b, o, M, N, K, epi, nullptr, g_shadow, st);
    return gemm0(dev, A, W, b, o, M, N, K, epi, st);
  };
  err_fn lasterr = (err_fn)dlsym(h, "clipx_last_error");
  if (!gemm0) return 2;
  int reps = 10, burst = 0;
  std::vector<std::vector<int>> shapes;
  std::vector<Cfg> cfgs;
  bool after = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-r" && i + 1 < argc) { reps = atoi(argv[++i]); continue; }
    if (a == "-b" && i + 1 < argc) { burst = atoi(argv[++i]); continue; }  // also time `burst` launches back to back per configuration
    if (a == "--") { after = true; continue; }
    if (!after) {
      std::vector<int> s;
      char* p = argv[i];
      while (*p) { s.push_back((int)strtol(p, &p, 10)); if (*p == ',') ++p; }
      if (s.size() == 3) s.push_back(0);
      if (s.size() != 4) { fprintf(stderr, "bad shape %s\n", argv[i]); return 2; }
      shapes.push_back(s);
    } else {
      Cfg c;
      c.name = a;
      size_t p1 = a.find(':');
      c.v = a.substr(0, p1);
      if (p1 != std::string::npos) {
        size_t p2 = a.find(':', p1 + 1);
        c.d = a.substr(p1 + 1, p2 == std::string::npos ? std::string::npos : p2 - p1 - 1);
        if (p2 != std::string::npos) c.f = a.substr(p2 + 1);
      }
      cfgs.push_back(c);
    }
  }
  if (shapes.empty()) shapes.push_back({65792, 3072, 1024, 0});
  if (cfgs.empty()) cfgs.push_back({"3", "3", "", ""});
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2], epi = s[3];
    const bool f32out = epi == 3;
    const size_t nA = (size_t)M * K, nW = (size_t)N * K, nO = (size_t)M * N;
    std::vector<uint16_t> hA(nA), hW(nW);
    std::vector<float> hb(N);
    unsigned r = 12345u + M + 3 * N + 7 * K;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((int)(r >> 9) & 0xffff) / 32768.f - 1.f; };
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); };
    auto hf = [](float f) { _Float16 hv = (_Float16)f; uint16_t u; memcpy(&u, &hv, 2); return u; };
    const bool f16ops = epi >= 16;
    for (auto& v : hA) v = f16ops ? hf(rnd()) : bf(rnd());
    for (auto& v : hW) v = f16ops ? hf(rnd() * 0.05f) : bf(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    void *dA, *dW, *dO, *dRef, *dInit;
    float* db;
    const size_t ob = nO * (f32out ? 4 : 2);
    CK(hipMalloc(&dA, nA * 2)); CK(hipMalloc(&dW, nW * 2)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dO, ob)); CK(hipMalloc(&dRef, ob)); CK(hipMalloc(&dInit, ob));
    if (want_shadow && f32out) CK(hipMalloc(&g_shadow, nO * 2));
    CK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    {  // residual input (epi 3 / 6 accumulate into out): small deterministic values
      std::vector<float> hi(f32out ? nO : 1);
      for (auto& v : hi) v = rnd();
      CK(hipMemset(dInit, 0, ob));
      if (f32out) CK(hipMemcpy(dInit, hi.data(), ob, hipMemcpyHostToDevice));
      if (epi == 6) {
        std::vector<uint16_t> h16(nO);
        for (auto& v : h16) v = hf(rnd() * 4.f);
        CK(hipMemcpy(dInit, h16.data(), ob, hipMemcpyHostToDevice));
      }
    }
    // reference = variant 0
    Cfg ref{"0", "0", "", ""};
    set_cfg(ref);
    CK(hipMemcpyAsync(dRef, dInit, ob, hipMemcpyDeviceToDevice, st));
    if (gemm(0, dA, dW, db, dRef, M, N, K, epi, st)) { fprintf(stderr, "gemm: %s\n", lasterr()); return 2; }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned char> href(ob), hout(ob);
    CK(hipMemcpy(href.data(), dRef, ob, hipMemcpyDeviceToHost));
    printf("shape M=%d N=%d K=%d epi=%d  (%.1f GFLOP)\n", M, N, K, epi, 2.0 * M * N * K / 1e9);
    std::vector<std::vector<float>> times(cfgs.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nchk = getenv("GEMM_BENCH_CHECKS") ? atoi(getenv("GEMM_BENCH_CHECKS")) : 1;
    for (size_t c = 0; c < cfgs.size(); ++c) {  // correctness (nchk independent launches) + warm-up
      set_cfg(cfgs[c]);
      for (int k = 0; k < nchk; ++k) {
        CK(hipMemcpyAsync(dO, dInit, ob, hipMemcpyDeviceToDevice, st));
        if (gemm(0, dA, dW, db, dO, M, N, K, epi, st)) { fprintf(stderr, "gemm %s: %s\n", cfgs[c].name.c_str(), lasterr()); return 2; }
        CK(hipStreamSynchronize(st));
        if (cfgs[c].d.empty() || cfgs[c].d == "0") {
          CK(hipMemcpy(hout.data(), dO, ob, hipMemcpyDeviceToHost));
          size_t bad = 0, first_bad = 0;
          for (size_t i = 0; i < ob; ++i) if (hout[i] != href[i]) { if (!bad) first_bad = i; ++bad; }
          if (bad) {
            const size_t es = f32out ? 4 : 2, e = first_bad / es;
            printf("  cfg %-10s vs variant 0: MISMATCH (%zu differing bytes; first at row %zu col %zu)\n", cfgs[c].name.c_str(), bad, e / N, e % N);
          } else {
            printf("  cfg %-10s vs variant 0: bitwise equal\n", cfgs[c].name.c_str());
          }
        }
      }
    }
    for (int rep = 0; rep < reps; ++rep)
      for (size_t c = 0; c < cfgs.size(); ++c) {
        set_cfg(cfgs[c]);
        CK(hipEventRecord(e0, st));
        gemm(0, dA, dW, db, dO, M, N, K, epi, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[c].push_back(ms);
      }
    std::vector<float> sustained(cfgs.size(), 0.f);
    if (burst > 0)
      for (size_t c = 0; c < cfgs.size(); ++c) {  // the encoder's regime: the part stays at its power limit (median of 3 bursts)
        set_cfg(cfgs[c]);
        float b3[3];
        for (int k = 0; k < 3; ++k) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < burst; ++i) gemm(0, dA, dW, db, dO, M, N, K, epi, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&b3[k], e0, e1));
        }
        std::sort(b3, b3 + 3);
        sustained[c] = b3[1] / burst;
      }
    typedef int (*dbg_fn)(long long*, int);
    dbg_fn dbgf8 = (dbg_fn)dlsym(h, "clipx_dbg_phase_cycles"), dbgf4 = (dbg_fn)dlsym(h, "clipx_dbg_phase_cycles_w4");
    for (size_t c = 0; c < cfgs.size(); ++c) {
      dbg_fn dbgf = cfgs[c].v == "6" ? dbgf4 : dbgf8;  // variant 6 = the 4-wave kernel (gemm256w4.hip): [1] = first K-tile after an epilogue, [3] / [5] = the others
      if (!dbgf) continue;
      if (cfgs[c].d != "16" && cfgs[c].d != "19" && cfgs[c].d != "20" && cfgs[c].d != "21" && cfgs[c].d != "22" && cfgs[c].d != "23" && cfgs[c].d != "25") continue;
      set_cfg(cfgs[c]);
      gemm(0, dA, dW, db, dO, M, N, K, epi, st);
      CK(hipStreamSynchronize(st));
      std::vector<long long> ph(256 * 8);
      if (dbgf(ph.data(), 256 * 8)) continue;
      double s8[8] = {0};
      for (int b = 0; b < 256; ++b) for (int i = 0; i < 8; ++i) s8[i] += ph[b * 8 + i] / 256.0;
      const double tiles = s8[6] > 0 ? s8[6] : 1;
      printf("  cfg %-10s phases (cycles, mean over blocks): epilogue %.0f per tile | first pair %.0f + %.0f | steady K-tile %.0f | last pair %.0f | kernel %.0f cycles, %.1f tiles\n",
             cfgs[c].name.c_str(), s8[0] / tiles, s8[1] / tiles, s8[2] / tiles, s8[5] > 0 ? s8[3] / s8[5] : 0.0, s8[4] / tiles, s8[7], tiles);
    }
    for (size_t c = 0; c < cfgs.size(); ++c) {
      std::sort(times[c].begin(), times[c].end());
      const float med = times[c][times[c].size() / 2], mn = times[c][0];
      printf("  cfg %-10s median %.4f ms  %7.1f TF   (min %.4f ms %7.1f TF)", cfgs[c].name.c_str(), med,
             2.0 * M * N * K / (med * 1e-3) / 1e12, mn, 2.0 * M * N * K / (mn * 1e-3) / 1e12);
      if (burst > 0) printf("   sustained x%d: %.4f ms %7.1f TF", burst, sustained[c], 2.0 * M * N * K / (sustained[c] * 1e-3) / 1e12);
      printf("\n");
    }
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dO)); CK(hipFree(dRef)); CK(hipFree(dInit));
    if (g_shadow) { CK(hipFree(g_shadow)); g_shadow = nullptr; }
  }
  return 0;
}
