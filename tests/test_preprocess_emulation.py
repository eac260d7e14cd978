"""The resize / crop kernel of csrc/preprocess.hip executed on the CPU: the kernel source is compiled for the host behind a small
shim (threadIdx / blockIdx as thread-locals, one std::thread per lane, __syncthreads = a pthread barrier, the dynamic LDS a
heap buffer) together with the library's own host-side coefficient code, and must reproduce Pillow byte for byte.  The GPU
tests (tests/test_preprocess_gpu.py) hold the real thing to the same bar; this one keeps the kernel's index arithmetic under
test where there is no GPU (it is also how the v_ashr_pk_u8_i32 mis-selection was told apart from a logic error: DESIGN 4.5)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
from PIL import Image

from clip_retrieval_amd.reader import clip_preprocess_u8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "clip-retrieval_amd", "csrc", "preprocess.hip")

SHIM = r'''
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <pthread.h>
struct uint4 { unsigned x, y, z, w; };
struct dim3i { int x, y; };
static thread_local dim3i threadIdx, blockIdx;
static pthread_barrier_t g_bar;
static unsigned char* g_lds;
#define __syncthreads() pthread_barrier_wait(&g_bar)
#define __restrict__
#define CLIPX_E_ARG 1
#define CLIPX_E_UNSUPPORTED 2
#define CLIPX_OK 0
using std::max;
using std::min;
namespace {
%(device)s
int fail(int c, const std::string& m) { fprintf(stderr, "emulated clipx_resize_crop_u8_device: %%s\n", m.c_str()); return c; }
}
extern "C" int emu(const unsigned char* src_dev, const int64_t* offsets, const int32_t* hw, int B, int S, unsigned char* out_dev) {
%(host)s
  std::vector<unsigned char> ldsbuf(max_lds + 64);
  g_lds = (unsigned char*)(((uintptr_t)ldsbuf.data() + 15) & ~(uintptr_t)15);
  pthread_barrier_init(&g_bar, nullptr, 256);
  for (int by = 0; by < B; ++by)
    for (int bx = 0; bx < max_bands; ++bx) {
      std::vector<std::thread> th;
      for (int t = 0; t < 256; ++t)
        th.emplace_back([&, t] {
          threadIdx.x = t; blockIdx.x = bx; blockIdx.y = by;
          resize_crop_kernel(src_dev, src_bytes, cb.data(), (const ImgDesc*)cb.data(), S, out_dev);
        });
      for (auto& x : th) x.join();
    }
  pthread_barrier_destroy(&g_bar);
  return 0;
}
'''


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    s = open(SRC).read()
    device = s[s.index("constexpr int PRECISION_BITS"):s.index("// device-side coefficient buffers")]
    device = (device.replace("__global__ __launch_bounds__(256) ", "").replace("__device__ __forceinline__", "static inline")
              .replace("extern __shared__ __attribute__((aligned(16))) unsigned char lds[];", "unsigned char* lds = g_lds;")
              .replace('asm volatile("" : "+v"(v));', ""))
    host = s[s.index("  // ---- geometry + weights of every image"):s.index("  PPCHK(hipSetDevice(device));")]
    d = tmp_path_factory.mktemp("emu")
    (d / "emu.cpp").write_text(SHIM % {"device": device, "host": host})
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(d / "emu.so"), str(d / "emu.cpp"), "-lpthread"], check=True)
    return C.CDLL(str(d / "emu.so"))


def _run(lib, imgs, S, misalign=0):
    hw = np.asarray([im.shape[:2] for im in imgs], dtype=np.int32)
    nb = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    off = np.zeros(len(imgs), dtype=np.int64)
    np.cumsum(nb[:-1], out=off[1:])
    flat = np.concatenate([im.reshape(-1) for im in imgs])
    buf = np.zeros(flat.size + 64, dtype=np.uint8)
    o = (-buf.ctypes.data) % 16 + misalign
    a = buf[o:o + flat.size]
    a[:] = flat
    out = np.zeros((len(imgs), S, S, 3), dtype=np.uint8)
    rc = lib.emu(C.c_void_p(a.ctypes.data), C.c_void_p(off.ctypes.data), C.c_void_p(hw.ctypes.data), len(imgs), S, C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


@pytest.mark.parametrize("S,misalign", [(224, 0), (224, 1), (64, 0)])
def test_emulated_kernel_equals_pillow(emu, S, misalign):
    rng = np.random.default_rng(S + misalign)
    sizes = [(224, 224), (256, 256), (256, 256), (300, 451), (97, 131), (17, 400), (225, 223), (1, 1), (2, 900), (640, 480)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    got = _run(emu, imgs, S, misalign)
    for im, g in zip(imgs, got):
        want = np.asarray(clip_preprocess_u8(Image.fromarray(im), size=S))
        assert np.array_equal(g, want), (im.shape[:2], S, misalign, int((g != want).sum()))
