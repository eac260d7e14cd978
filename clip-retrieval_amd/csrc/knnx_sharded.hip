// knnx_sharded.hip -- ONE process, several GPUs: a row-sharded index behind one faiss-shaped handle.
//
// `KnnService` holds one index object per modality inside a single Flask process and calls it from its request threads
// (reference clip_retrieval/clip_back.py:343-362, 781-782, 1018).  BASELINE config 5 ("IVF-Flat 1 B x 1024 sharded over
// 8 GPUs, served via clip_back KnnService") therefore needs the shards of all 8 devices behind ONE object in ONE
// process -- SURVEY 8(b) `knnx_create(n_devices, devices, ...)` / 8(e).  This file is that object:
//   * shard g = rows [lo_g, hi_g) of the global row order on devices[g] (a flat or IVF-Flat knnx_index with id_base = lo_g);
//   * search: the queries go to every device (B * d * 4 bytes each), every device scans its shard on its own stream --
//     all devices run concurrently, the host thread only enqueues --, the per-shard top-k (B * k * 12 bytes) are copied
//     peer-to-peer over xGMI into one buffer on devices[0] (hipMemcpyPeerAsync: the exchange is a few KB, latency-bound,
//     a direct copy per shard is the one-step exchange SURVEY 8(e) asks for), and the same merge kernel that follows the
//     RCCL all-gather of the one-process-per-GPU path (knnx_merge_topk_device) produces the final top-k;
//     -- or, OPT-IN with KNNX_SHARDS_RCCL=1, when every shard sits on its own device and librccl.so can be loaded (round 5; SURVEY 8(e): "RCCL ncclAllGather of
//     B * k * 12 B per rank inside one process (ncclCommInitAll)"), by ONE grouped all-gather over RCCL: a communicator per device
//     from ncclCommInitAll, ncclGroupStart; per device ncclAllGather(D) + ncclAllGather(I) on its own stream; ncclGroupEnd -- every
//     device then holds all P lists and devices[0] merges.  RCCL is loaded with dlopen on first use (the library has no link-time
//     dependency on it and a process that does not ask for it never loads it); the default is the peer copies, KNNX_SHARDS_RCCL=1 takes
//     RCCL (also for a single shard: the one-GPU test of this path).  knnx_shards_exchange() says which one is in use.  NEVER RUN ON
//     MORE THAN ONE GPU by its author (one-GPU test boxes): correct by construction and by the single-device communicator test only --
//     which is why it is opt-in, and why an exchange that returns an error hands the batch (and the index, from then on) to the peer
//     copies instead of failing the search.
//   * reconstruct: ids are routed to the owning shard by row range.
// Built on the public entry points of include/knnx.h only.  No CPU arithmetic: the host routes ids and pointers.

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <float.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/knnx.h"
#include "knn_kernels.h"

extern "C" int knnx_set_error(int code, const char* msg);  // knnx_api.hip: sets the thread-local message, returns code

namespace {

// ---- RCCL through dlopen: the five entry points the exchange needs (rccl.h: ncclResult_t = int, ncclSuccess = 0; ncclFloat32 = 7,
// ncclInt64 = 4 in ncclDataType_t -- the values of rccl.h / nccl.h, checked by the single-device test)
struct Rccl {
  void* so = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int RCCL_INT64 = 4, RCCL_FLOAT32 = 7;
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) return;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.so, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
  });
  return r;
}

struct Shard {
  knnx_index* ix = nullptr;
  int device = 0;
  int64_t lo = 0, hi = 0;      // global row range (hi fixed by reserve(); rows present = knnx_ntotal)
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  float* q = nullptr;          // [cap_q, d]
  float* D = nullptr;          // [cap_q, 64]
  int64_t* I = nullptr;
  void* comm = nullptr;        // RCCL communicator of this device (rank = shard index), or null
  float* agD = nullptr;        // RCCL: all P lists as this device receives them, [P, cap_q, 64]
  int64_t* agI = nullptr;
};

}  // namespace

struct knnx_shards {
  int d = 0;
  std::vector<Shard> sh;
  std::mutex mu;
  bool reserved = false;
  int cap_q = 1024;            // queries per exchange round
  // on devices[0]
  hipStream_t st0 = nullptr;
  float* gD = nullptr;         // [P, cap_q, 64]
  int64_t* gI = nullptr;
  float* mD = nullptr;         // [cap_q, 64]
  int64_t* mI = nullptr;
  void* pin = nullptr;         // pinned: queries | D | I
  size_t pin_q = 0, pin_d = 0;
  bool use_rccl = false;       // the per-shard top-k lists travel by one grouped ncclAllGather instead of peer copies
};

#define SHIP(expr)                                                                                                  \
  do {                                                                                                              \
    hipError_t _e = (expr);                                                                                         \
    if (_e != hipSuccess)                                                                                           \
      return knnx_set_error(_e == hipErrorOutOfMemory ? KNNX_E_NOMEM : KNNX_E_HIP, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
  } while (0)

// give the RCCL exchange up: communicators destroyed, receive buffers freed, the peer-copy form from now on
static void shards_drop_rccl(knnx_shards* s) {
  s->use_rccl = false;
  for (auto& h : s->sh) {
    (void)hipSetDevice(h.device);
    if (h.comm && rccl().ok) (void)rccl().CommDestroy(h.comm);
    if (h.agD) (void)hipFree(h.agD);
    if (h.agI) (void)hipFree(h.agI);
    h.comm = nullptr;
    h.agD = nullptr;
    h.agI = nullptr;
  }
  if (!s->sh.empty()) (void)hipSetDevice(s->sh[0].device);
}

static int shards_finish_setup(knnx_shards* s) {
  const int P = (int)s->sh.size();
  for (int g = 0; g < P; ++g) {
    Shard& h = s->sh[g];
    SHIP(hipSetDevice(h.device));
    SHIP(hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking));
    SHIP(hipEventCreateWithFlags(&h.ev, hipEventDisableTiming));
    SHIP(hipMalloc(&h.q, (size_t)s->cap_q * s->d * sizeof(float)));
    SHIP(hipMalloc(&h.D, (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(float)));
    SHIP(hipMalloc(&h.I, (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(int64_t)));
    for (int o = 0; o < P; ++o) {  // peer access both ways where the topology offers it; the copies work without it
      if (s->sh[o].device == h.device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, h.device, s->sh[o].device) == hipSuccess && can) {
        hipError_t e = hipDeviceEnablePeerAccess(s->sh[o].device, 0);
        if (e != hipSuccess) (void)hipGetLastError();  // already enabled
      }
    }
  }
  SHIP(hipSetDevice(s->sh[0].device));
  SHIP(hipStreamCreateWithFlags(&s->st0, hipStreamNonBlocking));
  SHIP(hipMalloc(&s->gD, (size_t)P * s->cap_q * KNNX_MAX_K_FAST * sizeof(float)));
  SHIP(hipMalloc(&s->gI, (size_t)P * s->cap_q * KNNX_MAX_K_FAST * sizeof(int64_t)));
  SHIP(hipMalloc(&s->mD, (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(float)));
  SHIP(hipMalloc(&s->mI, (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(int64_t)));
  s->pin_q = (size_t)s->cap_q * s->d * sizeof(float);
  s->pin_d = (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(float);
  SHIP(hipHostMalloc(&s->pin, s->pin_q + s->pin_d + (size_t)s->cap_q * KNNX_MAX_K_FAST * sizeof(int64_t), hipHostMallocDefault));
  // ---- RCCL exchange: every shard on its own device (a communicator cannot hold one GPU twice).  OPT-IN (KNNX_SHARDS_RCCL=1) until it
  // has run on a box with more than one GPU: the peer-copy form is the default, and a failing exchange falls back to it.
  const char* env = getenv("KNNX_SHARDS_RCCL");
  const int want = env ? atoi(env) : 0;
  bool distinct = true;
  for (int g = 0; g < P; ++g)
    for (int o = 0; o < g; ++o) distinct = distinct && s->sh[o].device != s->sh[g].device;
  if (want == 1 && distinct && rccl().ok) {
    std::vector<int> devs(P);
    std::vector<void*> comms(P, nullptr);
    for (int g = 0; g < P; ++g) devs[g] = s->sh[g].device;
    if (rccl().CommInitAll(comms.data(), P, devs.data()) == 0) {
      bool mem = true;
      for (int g = 0; g < P; ++g) s->sh[g].comm = comms[g];  // all of them, so that a failure below can destroy all of them
      for (int g = 0; g < P && mem; ++g) {
        Shard& h = s->sh[g];
        mem = hipSetDevice(h.device) == hipSuccess &&
              hipMalloc(&h.agD, (size_t)P * s->cap_q * KNNX_MAX_K_FAST * sizeof(float)) == hipSuccess &&
              hipMalloc(&h.agI, (size_t)P * s->cap_q * KNNX_MAX_K_FAST * sizeof(int64_t)) == hipSuccess;
      }
      s->use_rccl = mem;
      if (!mem) {
        (void)hipGetLastError();
        shards_drop_rccl(s);
      }
    }
    (void)hipSetDevice(s->sh[0].device);
  }
  return KNNX_OK;
}

extern "C" int knnx_shards_create(int n_shards, const int* devices, int d, int metric, knnx_shards** out) {
  if (!out || !devices || n_shards <= 0 || n_shards > 64) return knnx_set_error(KNNX_E_ARG, "bad shards_create arguments");
  *out = nullptr;
  knnx_shards* s = new knnx_shards();
  s->d = d;
  s->sh.resize(n_shards);
  for (int g = 0; g < n_shards; ++g) {
    s->sh[g].device = devices[g];
    int r = knnx_create(devices[g], d, metric, &s->sh[g].ix);
    if (r) {
      knnx_shards_destroy(s);
      return r;
    }
  }
  int r = shards_finish_setup(s);
  if (r) {
    knnx_shards_destroy(s);
    return r;
  }
  *out = s;
  return KNNX_OK;
}

extern "C" int knnx_shards_adopt(int n_shards, knnx_index* const* shards, const int* devices, const int64_t* row_lo,
                                 knnx_shards** out) {
  if (!out || !shards || !devices || !row_lo || n_shards <= 0 || n_shards > 64) return knnx_set_error(KNNX_E_ARG, "bad shards_adopt arguments");
  *out = nullptr;
  const int d = knnx_dim(shards[0]);
  for (int g = 0; g < n_shards; ++g)
    if (!shards[g] || knnx_dim(shards[g]) != d) return knnx_set_error(KNNX_E_ARG, "shards disagree on the dimension");
  knnx_shards* s = new knnx_shards();
  s->d = d;
  s->sh.resize(n_shards);
  for (int g = 0; g < n_shards; ++g) {
    s->sh[g].ix = shards[g];
    s->sh[g].device = devices[g];
    s->sh[g].lo = row_lo[g];
    s->sh[g].hi = row_lo[g] + knnx_ntotal(shards[g]);
  }
  s->reserved = true;
  int r = shards_finish_setup(s);
  if (r) {
    for (auto& h : s->sh) h.ix = nullptr;  // the caller still owns them on failure
    knnx_shards_destroy(s);
    return r;
  }
  *out = s;
  return KNNX_OK;
}

extern "C" void knnx_shards_destroy(knnx_shards* s) {
  if (!s) return;
  for (auto& h : s->sh) {
    (void)hipSetDevice(h.device);
    if (h.st) (void)hipStreamSynchronize(h.st);
    if (h.q) (void)hipFree(h.q);
    if (h.D) (void)hipFree(h.D);
    if (h.I) (void)hipFree(h.I);
    if (h.comm && rccl().ok) (void)rccl().CommDestroy(h.comm);
    if (h.agD) (void)hipFree(h.agD);
    if (h.agI) (void)hipFree(h.agI);
    if (h.ev) (void)hipEventDestroy(h.ev);
    if (h.st) (void)hipStreamDestroy(h.st);
    if (h.ix) knnx_destroy(h.ix);
  }
  if (!s->sh.empty()) (void)hipSetDevice(s->sh[0].device);
  if (s->st0) (void)hipStreamSynchronize(s->st0);
  if (s->gD) (void)hipFree(s->gD);
  if (s->gI) (void)hipFree(s->gI);
  if (s->mD) (void)hipFree(s->mD);
  if (s->mI) (void)hipFree(s->mI);
  if (s->pin) (void)hipHostFree(s->pin);
  if (s->st0) (void)hipStreamDestroy(s->st0);
  delete s;
}

extern "C" int knnx_shards_count(const knnx_shards* s) { return s ? (int)s->sh.size() : 0; }
extern "C" int knnx_shards_exchange(const knnx_shards* s) { return s ? (s->use_rccl ? 1 : 0) : -1; }
extern "C" knnx_index* knnx_shards_get(knnx_shards* s, int g) { return (s && g >= 0 && g < (int)s->sh.size()) ? s->sh[g].ix : nullptr; }

extern "C" int64_t knnx_shards_ntotal(const knnx_shards* s) {
  int64_t n = 0;
  if (s) for (auto& h : s->sh) n += knnx_ntotal(h.ix);
  return n;
}

extern "C" int knnx_shards_reserve(knnx_shards* s, int64_t total_rows) {
  if (!s || total_rows < 0) return knnx_set_error(KNNX_E_ARG, "bad shards_reserve arguments");
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->reserved) return knnx_set_error(KNNX_E_STATE, "row ranges are already fixed");
  const int P = (int)s->sh.size();
  for (int g = 0; g < P; ++g) {
    Shard& h = s->sh[g];
    h.lo = total_rows * g / P;
    h.hi = total_rows * (g + 1) / P;
    int r = knnx_set_id_base(h.ix, h.lo);
    if (!r) r = knnx_reserve(h.ix, h.hi - h.lo);
    if (r) return r;
  }
  s->reserved = true;
  return KNNX_OK;
}

static int shards_add(knnx_shards* s, const void* rows, int64_t n, bool f32) {
  if (!s || (n > 0 && !rows) || n < 0) return knnx_set_error(KNNX_E_ARG, "bad shards_add arguments");
  std::lock_guard<std::mutex> lk(s->mu);
  if (!s->reserved) return knnx_set_error(KNNX_E_STATE, "call knnx_shards_reserve(total_rows) before add: it fixes the row range of every shard");
  const size_t esz = (f32 ? 4 : 2) * (size_t)s->d;
  int64_t done = 0;
  for (auto& h : s->sh) {
    if (done == n) break;
    const int64_t room = (h.hi - h.lo) - knnx_ntotal(h.ix);
    if (room <= 0) continue;
    const int64_t m = std::min(room, n - done);
    const char* src = (const char*)rows + (size_t)done * esz;
    int r = f32 ? knnx_add_f32(h.ix, (const float*)src, m) : knnx_add_f16(h.ix, (const uint16_t*)src, m);
    if (r) return r;
    done += m;
  }
  if (done != n) return knnx_set_error(KNNX_E_NOMEM, "more rows added than knnx_shards_reserve announced");
  return KNNX_OK;
}
extern "C" int knnx_shards_add_f16(knnx_shards* s, const uint16_t* rows, int64_t n) { return shards_add(s, rows, n, false); }
extern "C" int knnx_shards_add_f32(knnx_shards* s, const float* rows, int64_t n) { return shards_add(s, rows, n, true); }

extern "C" int knnx_shards_synth_fill(knnx_shards* s, int64_t rows_per_shard, uint64_t seed) {
  if (!s || rows_per_shard < 0) return knnx_set_error(KNNX_E_ARG, "bad shards_synth_fill arguments");
  std::lock_guard<std::mutex> lk(s->mu);
  int64_t lo = 0;
  for (size_t g = 0; g < s->sh.size(); ++g) {
    Shard& h = s->sh[g];
    int r = knnx_set_id_base(h.ix, lo);
    if (!r) r = knnx_synth_fill(h.ix, rows_per_shard, seed + g);
    if (r) return r;
    h.lo = lo;
    h.hi = lo + rows_per_shard;
    lo = h.hi;
  }
  s->reserved = true;
  return KNNX_OK;
}

// k <= 64: scans on every device concurrently, peer copies to devices[0], merge there
static int shards_search_fast(knnx_shards* s, const float* q, int n, int k, float* D, int64_t* I) {
  const int P = (int)s->sh.size(), d = s->d;
  char* pin = (char*)s->pin;
  for (int o = 0; o < n; o += s->cap_q) {
    const int nb = std::min(s->cap_q, n - o);
    memcpy(pin, q + (size_t)o * d, (size_t)nb * d * sizeof(float));
    for (int g = 0; g < P; ++g) {
      Shard& h = s->sh[g];
      SHIP(hipSetDevice(h.device));
      SHIP(hipMemcpyAsync(h.q, pin, (size_t)nb * d * sizeof(float), hipMemcpyHostToDevice, h.st));
      int r = knnx_search_device(h.ix, h.q, nb, k, h.D, h.I, h.st);
      if (r) return r;
    }
    const float* partD = s->gD;
    const int64_t* partI = s->gI;
    bool gathered = false;
    if (s->use_rccl) {
      // ONE grouped exchange: every device contributes its nb * k (score, id) pairs -- nb * k * 12 bytes -- and receives all P lists,
      // rank-major = shard-major = ascending id order, the layout the merge kernel takes
      const size_t cnt = (size_t)nb * k;
      int rc = rccl().GroupStart();
      for (int g = 0; g < P && rc == 0; ++g) {
        Shard& h = s->sh[g];
        rc = rccl().AllGather(h.D, h.agD, cnt, RCCL_FLOAT32, h.comm, h.st);
        if (rc == 0) rc = rccl().AllGather(h.I, h.agI, cnt, RCCL_INT64, h.comm, h.st);
      }
      const int rc2 = rccl().GroupEnd();
      if (rc == 0) rc = rc2;
      if (rc == 0) {
        SHIP(hipSetDevice(s->sh[0].device));
        SHIP(hipEventRecord(s->sh[0].ev, s->sh[0].st));
        SHIP(hipStreamWaitEvent(s->st0, s->sh[0].ev, 0));
        partD = s->sh[0].agD;
        partI = s->sh[0].agI;
        gathered = true;
      } else {
        // the exchange failed: this index answers through peer copies from now on, starting with this very batch (the per-shard
        // lists are still in h.D / h.I, ordered on h.st)
        for (auto& h : s->sh) {
          (void)hipSetDevice(h.device);
          (void)hipStreamSynchronize(h.st);
        }
        (void)hipGetLastError();
        shards_drop_rccl(s);
      }
    }
    if (!gathered) {
      for (int g = 0; g < P; ++g) {
        Shard& h = s->sh[g];
        const size_t cnt = (size_t)nb * k;
        SHIP(hipSetDevice(h.device));
        if (h.device == s->sh[0].device) {
          SHIP(hipMemcpyAsync(s->gD + (size_t)g * cnt, h.D, cnt * sizeof(float), hipMemcpyDeviceToDevice, h.st));
          SHIP(hipMemcpyAsync(s->gI + (size_t)g * cnt, h.I, cnt * sizeof(int64_t), hipMemcpyDeviceToDevice, h.st));
        } else {
          SHIP(hipMemcpyPeerAsync(s->gD + (size_t)g * cnt, s->sh[0].device, h.D, h.device, cnt * sizeof(float), h.st));
          SHIP(hipMemcpyPeerAsync(s->gI + (size_t)g * cnt, s->sh[0].device, h.I, h.device, cnt * sizeof(int64_t), h.st));
        }
        SHIP(hipEventRecord(h.ev, h.st));
      }
      SHIP(hipSetDevice(s->sh[0].device));
      for (int g = 0; g < P; ++g) SHIP(hipStreamWaitEvent(s->st0, s->sh[g].ev, 0));
    }
    int r = knnx_merge_topk_device(s->sh[0].device, partD, partI, P, nb, k, s->mD, s->mI, s->st0);
    if (r) return r;
    SHIP(hipMemcpyAsync(pin + s->pin_q, s->mD, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, s->st0));
    SHIP(hipMemcpyAsync(pin + s->pin_q + s->pin_d, s->mI, (size_t)nb * k * sizeof(int64_t), hipMemcpyDeviceToHost, s->st0));
    SHIP(hipStreamSynchronize(s->st0));
    memcpy(D + (size_t)o * k, pin + s->pin_q, (size_t)nb * k * sizeof(float));
    memcpy(I + (size_t)o * k, pin + s->pin_q + s->pin_d, (size_t)nb * k * sizeof(int64_t));
  }
  return KNNX_OK;
}

// k > 64 (front-end num_result_ids = 3000): every shard answers through its own large-k path (host buffers), the P sorted
// lists go to devices[0] and a P-way merge of sorted lists runs there (knnx_merge_sorted_device)
static int shards_search_large(knnx_shards* s, const float* q, int n, int k, float* D, int64_t* I) {
  const int P = (int)s->sh.size();
  const size_t cnt = (size_t)n * k;
  std::vector<float> hD(P * cnt);
  std::vector<int64_t> hI(P * cnt);
  for (int g = 0; g < P; ++g) {
    int r = knnx_search(s->sh[g].ix, q, n, k, hD.data() + g * cnt, hI.data() + g * cnt, nullptr);
    if (r) return r;
  }
  SHIP(hipSetDevice(s->sh[0].device));
  float *dD = nullptr, *oD = nullptr;
  int64_t *dI = nullptr, *oI = nullptr;
  SHIP(hipMalloc(&dD, P * cnt * sizeof(float)));
  hipError_t e = hipMalloc(&dI, P * cnt * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&oD, cnt * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&oI, cnt * sizeof(int64_t));
  if (e == hipSuccess) e = hipMemcpyAsync(dD, hD.data(), P * cnt * sizeof(float), hipMemcpyHostToDevice, s->st0);
  if (e == hipSuccess) e = hipMemcpyAsync(dI, hI.data(), P * cnt * sizeof(int64_t), hipMemcpyHostToDevice, s->st0);
  if (e == hipSuccess) e = knnx::launch_merge_sorted(dD, dI, P, n, k, oD, oI, s->st0);
  if (e == hipSuccess) e = hipMemcpyAsync(D, oD, cnt * sizeof(float), hipMemcpyDeviceToHost, s->st0);
  if (e == hipSuccess) e = hipMemcpyAsync(I, oI, cnt * sizeof(int64_t), hipMemcpyDeviceToHost, s->st0);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st0);
  (void)hipFree(dD);
  if (dI) (void)hipFree(dI);
  if (oD) (void)hipFree(oD);
  if (oI) (void)hipFree(oI);
  if (e != hipSuccess) return knnx_set_error(KNNX_E_HIP, (std::string("sharded large-k merge: ") + hipGetErrorString(e)).c_str());
  return KNNX_OK;
}

extern "C" int knnx_shards_reconstruct(knnx_shards* s, const int64_t* ids, int64_t n, float* out) {
  if (!s || (n > 0 && (!ids || !out)) || n < 0) return knnx_set_error(KNNX_E_ARG, "bad shards_reconstruct arguments");
  const int d = s->d;
  std::vector<int64_t> sub;
  std::vector<int64_t> pos;
  std::vector<float> buf;
  std::vector<char> seen((size_t)n, 0);
  for (auto& h : s->sh) {
    sub.clear();
    pos.clear();
    const int64_t hi = h.lo + knnx_ntotal(h.ix);
    for (int64_t i = 0; i < n; ++i)
      if (ids[i] >= h.lo && ids[i] < hi) {
        sub.push_back(ids[i]);
        pos.push_back(i);
        seen[(size_t)i] = 1;
      }
    if (sub.empty()) continue;
    buf.resize(sub.size() * (size_t)d);
    int r = knnx_reconstruct(h.ix, sub.data(), (int64_t)sub.size(), buf.data());
    if (r) return r;
    for (size_t j = 0; j < sub.size(); ++j) memcpy(out + (size_t)pos[j] * d, buf.data() + j * d, (size_t)d * sizeof(float));
  }
  for (int64_t i = 0; i < n; ++i)
    if (!seen[(size_t)i]) memset(out + (size_t)i * d, 0xFF, (size_t)d * sizeof(float));  // id -1 / unknown: faiss fills 0xFF
  return KNNX_OK;
}

extern "C" int knnx_shards_search(knnx_shards* s, const float* q, int n, int k, float* D, int64_t* I, float* R) {
  if (!s || (n > 0 && (!q || !D || !I)) || n < 0 || k <= 0) return knnx_set_error(KNNX_E_ARG, "bad shards_search arguments");
  if (k > KNNX_MAX_K) return knnx_set_error(KNNX_E_UNSUPPORTED, "k > 131072 is not implemented");
  if (n == 0) return KNNX_OK;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    int r = k <= KNNX_MAX_K_FAST ? shards_search_fast(s, q, n, k, D, I) : shards_search_large(s, q, n, k, D, I);
    if (r) return r;
  }
  if (R) return knnx_shards_reconstruct(s, I, (int64_t)n * k, R);
  return KNNX_OK;
}

// range_search over the shards: shard order = ascending id order, so concatenating the per-shard hit lists of a query
// keeps the "ids ascending inside each query" contract.  Same two-call protocol as knnx_range_search.
extern "C" int knnx_shards_range_search(knnx_shards* s, const float* q, int n, float thresh, int64_t* lims, float* D, int64_t* I) {
  if (!s || !lims || (n > 0 && !q) || n < 0) return knnx_set_error(KNNX_E_ARG, "bad shards_range_search arguments");
  if ((D == nullptr) != (I == nullptr)) return knnx_set_error(KNNX_E_ARG, "D and I must both be null or both be set");
  std::lock_guard<std::mutex> lk(s->mu);
  const int P = (int)s->sh.size();
  std::vector<std::vector<int64_t>> sl(P, std::vector<int64_t>((size_t)n + 1, 0));
  for (int g = 0; g < P; ++g) {
    int r = knnx_range_search(s->sh[g].ix, q, n, thresh, sl[g].data(), nullptr, nullptr);
    if (r) return r;
  }
  if (!D) {
    lims[0] = 0;
    for (int i = 0; i < n; ++i) {
      int64_t c = 0;
      for (int g = 0; g < P; ++g) c += sl[g][i + 1] - sl[g][i];
      lims[i + 1] = lims[i] + c;
    }
    return KNNX_OK;
  }
  std::vector<float> sd;
  std::vector<int64_t> si;
  std::vector<int64_t> fill((size_t)n, 0);
  for (int g = 0; g < P; ++g) {
    const int64_t tot = sl[g][n];
    if (tot == 0) continue;
    sd.resize((size_t)tot);
    si.resize((size_t)tot);
    int r = knnx_range_search(s->sh[g].ix, q, n, thresh, sl[g].data(), sd.data(), si.data());
    if (r) return r;
    for (int i = 0; i < n; ++i) {
      const int64_t c = sl[g][i + 1] - sl[g][i];
      if (lims[i] + fill[i] + c > lims[i + 1]) return knnx_set_error(KNNX_E_STATE, "lims do not match this query/threshold");
      memcpy(D + lims[i] + fill[i], sd.data() + sl[g][i], (size_t)c * sizeof(float));
      memcpy(I + lims[i] + fill[i], si.data() + sl[g][i], (size_t)c * sizeof(int64_t));
      fill[i] += c;
    }
  }
  return KNNX_OK;
}
