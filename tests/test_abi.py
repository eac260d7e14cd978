"""The C-ABI library loads and exports every symbol include/*.h declares (no compute: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    prefix = header.split(".")[0]
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["knnx.h", "clipx.h"])
def test_every_declared_symbol_is_exported_and_typed(lib, header):
    from clip_retrieval_amd._lib import SIGNATURES

    names = declared_functions(header)
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
        assert n in SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    prefix = header.split(".")[0]
    extra = [n for n in SIGNATURES if n.startswith(prefix + "_") and n not in names]
    assert not extra, f"_lib.py binds symbols the header does not declare: {extra}"


def test_headers_compile_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "knnx.h"\n#include "clipx.h"\nint main(void){clipx_model_desc d; (void)d; return KNNX_OK + CLIPX_OK;}\n')
    rc = os.system(f"gcc -std=c99 -Wall -Werror -I{ROOT}/include -c {src} -o {tmp_path}/t.o")
    assert rc == 0


def test_desc_struct_layout_matches_header(lib):
    from clip_retrieval_amd._lib import ClipxModelDesc

    assert C.sizeof(ClipxModelDesc) == 15 * 4 + 6 * 4  # 14 ints + 1 float + 2 x float[3]


def test_blob_floats_matches_parameter_count(lib):
    from clip_retrieval_amd.encoder import ARCHS, blob_floats, random_blob

    # ViT-L/14: the published 427,616,513 parameters minus the scalar logit_scale (unused by encode_*)
    n = blob_floats(ARCHS["ViT-L/14"])
    assert n == 427_616_512, n
    tiny = ARCHS["ViT-B/32"]
    assert random_blob(tiny, 0).size == blob_floats(tiny)


def test_calls_fail_loudly_without_a_gpu(lib):
    """No CPU fallback: creating an index / encoder on a box without a HIP device is an error."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from clip_retrieval_amd import HipLibraryError
    from clip_retrieval_amd.knn import Mi355xIndex

    with pytest.raises(HipLibraryError):
        Mi355xIndex(768)


def test_host_merge_is_exact(lib):
    import numpy as np
    from clip_retrieval_amd.distributed import merge_topk_host
    from oracle.knn_oracle import merge_topk

    rng = np.random.default_rng(0)
    P, n, k = 8, 5, 40
    D = np.sort(rng.standard_normal((P, n, k)).astype(np.float32), axis=-1)[..., ::-1].copy()
    D[2, :, 10:] = D[3, :, 10:]  # exact score ties across shards -> id order decides
    I = rng.permutation(P * n * k).reshape(P, n, k).astype(np.int64)
    I[5, 1, 30:] = -1  # a short list
    D[5, 1, 30:] = np.float32(-3.4028234663852886e38)
    Dm, Im = merge_topk_host(D, I, k)
    Do, Io = merge_topk(D, I, k)
    assert np.array_equal(Im, Io) and np.array_equal(Dm, Do)


def test_inline_asm_vmem_hazards():
    """The GEMM kernels issue their LDS-DMA loads, fragment reads and epilogue stores from inline asm, which hipcc neither
    counts nor guards: lint the generated gfx950 code for VALU-written SGPRs read too early by a VMEM instruction, for
    fragment registers touched while their inline-asm ds_read is still in flight, and for VGPR spills in the persistent
    kernel (a scratch reload waits vmcnt(0), i.e. for every store before it).  tools/check_isa.py."""
    import os
    import shutil
    import subprocess
    import sys

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "clip-retrieval_amd", "csrc")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_isa.py"), os.path.join(csrc, "gemm256sp.hip"),
                        os.path.join(csrc, "clip_kernels.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
