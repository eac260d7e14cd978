"""GPU parity tests of the encode half: HIP kernels through the C ABI vs the fp32 CPU oracle.
Bar (north_star): cosine(oracle fp32, ours) >= 1 - 1e-3 per sample.  Per-kernel tests use a plain torch
fp32 reference of the same op (floating-point kernels), with tolerances stated inline."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
COS_BAR = 1.0 - 1e-3  # north_star; the encoder-vs-oracle tests below use oracle.parity_gate: raw cosine >= 1 - 1e-4, centred
                      # cosine >= 0.99 and "the nearest oracle row is my own row" (a row-swapped output fails all three)


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def lib():
    import clip_retrieval_amd

    return clip_retrieval_amd.load_library()


# ------------------------------------------------------------------------------------------ per-kernel
@pytest.mark.parametrize("variant", [0, 1, 3, 6])
@pytest.mark.parametrize("M,N,K", [(257, 128, 64), (514, 1024, 1024), (1000, 2304, 768), (130, 768, 3072), (65, 128, 640),
                                   (33357, 512, 128), (65792, 1024, 1024), (2048, 4096, 1024), (19712, 768, 3072), (16384, 256, 256),
                                   (16640, 256, 384), (8192, 1024, 640), (65536, 512, 2048)])
def test_gemm_epilogues(lib, variant, M, N, K):
    """out = A W^T + b with bf16 operands: reference is the fp32 matmul of the SAME bf16-rounded operands,
    so only accumulation order differs (tol 2e-3 * |row| scale for bf16 outputs = 1 bf16 ulp + sum noise).
    variant 3 = persistent 8-wave 256x256 kernel for the whole m-tiles + 128x128 kernel for the peeled rows; variant 6 (default) =
    the 4-wave 256x256 kernel (gemm256w4.hip) for the forms it has (16-bit outputs, K >= 256), variant 3 otherwise;
    (33357,512,128): 260 tiles -> two tiles per workgroup with K = one iteration; (65792,1024,1024): the ViT-L/14
    bs=256 out_proj shape (4 tiles per workgroup + 256 peeled rows)."""
    from clip_retrieval_amd._lib import check

    if variant < 2 and M > 20000:
        pytest.skip("large shapes exercise the persistent kernel's multi-tile stream only")

    os.environ["CLIPX_GEMM_VARIANT"] = str(variant)
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    # asymmetric, row/col dependent structure so a transposed or permuted tile cannot pass
    A[:, 0] += torch.arange(M, device="cuda").to(torch.bfloat16) * 0.01
    W[:, 1] += torch.arange(N, device="cuda").to(torch.bfloat16) * 0.003
    bias = torch.randn(N, generator=g, device="cuda")
    ref = A.float() @ W.float().T + bias
    for epi in (0, 1, 2, 3):
        if epi == 3:
            out = torch.randn(M, N, generator=g, device="cuda")
            want = out + ref
        else:
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            want = ref if epi == 0 else (ref * torch.sigmoid(1.702 * ref) if epi == 1 else torch.nn.functional.gelu(ref))
        check(lib, lib.clipx_gemm_bf16_device(0, _ptr(A), _ptr(W), _ptr(bias), _ptr(out), M, N, K, epi, None), "clipx")
        torch.cuda.synchronize()
        err = (out.float() - want).abs()
        tol = 1e-3 + (4e-3 * want.abs() if epi != 3 else 1e-4 * want.abs())
        bad = (err > tol).nonzero()
        assert bad.numel() == 0, (f"variant={variant} epi={epi} M={M} N={N} K={K}: {bad.shape[0]} bad, "
                                  f"first {bad[:4].tolist()} max err {err.max().item():.4g}")
    os.environ.pop("CLIPX_GEMM_VARIANT")


@pytest.mark.parametrize("d", [512, 768, 1024, 1280])
def test_layernorm(lib, d):
    from clip_retrieval_amd._lib import check

    M = 1031
    g = torch.Generator().manual_seed(d)
    x = (torch.randn(M, d, generator=g) * 3 + 0.7).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
    want = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    y32 = torch.empty_like(x)
    check(lib, lib.clipx_layernorm_device(0, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y32), 0, M, d, C.c_float(1e-5), None), "clipx")
    y16 = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
    check(lib, lib.clipx_layernorm_device(0, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y16), 1, M, d, C.c_float(1e-5), None), "clipx")
    torch.cuda.synchronize()
    assert (y32 - want).abs().max() < 2e-5
    assert (y16.float() - want).abs().max() < 0.02  # one bf16 ulp at |y| <= 4


@pytest.mark.parametrize("B,T,H,dh,causal", [(2, 257, 16, 64, 0), (3, 77, 12, 64, 1), (2, 50, 12, 64, 0), (1, 197, 12, 64, 0),
                                              (2, 77, 8, 64, 1), (1, 1, 1, 64, 0), (2, 257, 16, 80, 0), (2, 77, 4, 80, 1),
                                              (1, 33, 2, 80, 0)])
def test_attention(lib, B, T, H, dh, causal):
    """softmax(q k^T / sqrt(dh) [+causal]) v per head; reference in fp32 on the same IEEE fp16 inputs (round 4: q, k, v and P
    are fp16 operands).  The output is bf16: tol 1e-2 absolute on O(1) values.  dh = 80 is the ViT-H/14 image tower."""
    from clip_retrieval_amd._lib import check

    g = torch.Generator().manual_seed(T * 31 + H)
    qkv = (torch.randn(B * T, 3 * H * dh, generator=g)).to(torch.float16).cuda()
    qkv[:, : H * dh] *= 2.0  # sharper softmax
    out = torch.empty(B * T, H * dh, dtype=torch.bfloat16, device="cuda")
    if dh == 64:
        check(lib, lib.clipx_attention_device(0, _ptr(qkv), _ptr(out), B, T, H, causal, None), "clipx")
    else:
        check(lib, lib.clipx_attention_dh_device(0, _ptr(qkv), _ptr(out), B, T, H, dh, causal, None), "clipx")
    torch.cuda.synchronize()
    q, k, v = qkv.float().view(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    if causal:
        s = s + torch.full((T, T), float("-inf"), device="cuda").triu_(1)
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * dh)
    err = (out.float() - want).abs()
    assert err.max() < 2e-2, f"B={B} T={T} H={H} dh={dh} causal={causal}: max err {err.max().item():.4g} at {err.argmax().item()}"
    assert err.mean() < 2e-3


# ------------------------------------------------------------------------------------------ whole encoder
def _product_arch(arch):
    from clip_retrieval_amd.encoder import ClipArch

    return ClipArch(**{k: getattr(arch, k) for k in ClipArch.__dataclass_fields__})


@pytest.fixture(scope="module", params=["tiny-B/32", "tiny-L/14", "tiny-H/14"])
def tiny(request):
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import ARCHS, HFClipOracle

    arch = ARCHS[request.param]
    oracle = HFClipOracle(arch, seed=0)
    enc = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    yield request.param, arch, oracle, enc
    enc.close()


@pytest.mark.parametrize("B", [1, 2, 5])
def test_encoder_parity_vs_oracle(tiny, B):
    from oracle.clip_oracle import mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    u8 = synth_pixels_u8(B, seed=B)
    pix = normalise_u8_nhwc(u8)
    ids = synth_tokens(B, seed=10 + B)
    want_i16, want_i32 = mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))
    want_t16, want_t32 = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
    got_i = enc.encode_image(pix)
    got_t = enc.encode_text(ids)
    assert got_i.dtype == np.float16 and got_i.shape == (B, arch.embed_dim) and got_i.flags["C_CONTIGUOUS"]
    parity_gate(got_i, want_i32, f"{name} image")
    parity_gate(got_t, want_t32, f"{name} text")
    assert np.allclose(np.linalg.norm(got_i.astype(np.float32), axis=1), 1, atol=2e-3)
    assert np.abs(got_i.astype(np.float32) - want_i16.astype(np.float32)).max() < 0.02
    # the raw-uint8 entry point normalises on the device and must land on the same embedding
    got_u8 = enc.encode_image(u8)
    assert _cos(got_u8, got_i).min() > 1 - 1e-4


def test_parity_gate_rejects_swapped_stale_and_input_independent_rows(tiny):
    """the old gate (raw cosine >= 0.999 on i.i.d.-noise images, whose oracle embeddings have cosine
    0.998 with EACH OTHER) could not tell one row from another.  With the structured images and parity_gate, the HIP
    encoder's own outputs fail as soon as two rows are exchanged, a row is stale, or the output ignores its input."""
    from oracle.clip_oracle import mapper_semantics, normalise_u8_nhwc, parity_gate, parity_report, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    B = 6
    pix, ids = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=61)), synth_tokens(B, arch.ctx_len, arch.vocab, seed=62)
    _, wi = mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))
    _, wt = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
    gi, gt = enc.encode_image(pix), enc.encode_text(ids)
    rep = parity_gate(gi, wi, f"{name} image")
    parity_gate(gt, wt, f"{name} text")
    assert rep["other"].max() < 0.95, f"the inputs must separate the rows: nearest wrong row at cosine {rep['other']}"
    for got, want in ((gi, wi), (gt, wt)):
        swapped = got.copy()
        swapped[[1, 4]] = swapped[[4, 1]]
        stale = got.copy()
        stale[3] = stale[2]
        const = np.repeat(got.astype(np.float32).mean(0, keepdims=True), B, 0)
        for wrong in (swapped, stale, const, np.roll(got, 1, axis=0)):
            with pytest.raises(AssertionError):
                parity_gate(wrong, want, "negative")
        assert (parity_report(swapped, want)["nearest"][[1, 4]] == [4, 1]).all()


def test_device_path_and_fp16_rounding(tiny):
    """Device-buffer entry points: the fp16 result is exactly the RNE rounding of the fp32 normalised row."""
    from oracle.clip_oracle import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    B = 3
    pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(B, seed=77))).cuda()
    ids = torch.from_numpy(synth_tokens(B, seed=78)).cuda()
    o16 = torch.empty(B, arch.embed_dim, dtype=torch.float16, device="cuda")
    o32 = torch.empty(B, arch.embed_dim, dtype=torch.float32, device="cuda")
    enc.encode_image_device(pix.data_ptr(), B, 0, o16.data_ptr(), o32.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(o32.to(torch.float16), o16)
    assert torch.allclose(o32.norm(dim=-1), torch.ones(B, device="cuda"), atol=1e-5)
    assert np.array_equal(o16.cpu().numpy(), enc.encode_image(pix.cpu().numpy()))  # host path == device path, bitwise
    enc.encode_text_device(ids.data_ptr(), B, o16.data_ptr(), o32.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(o32.to(torch.float16), o16)
    assert np.array_equal(o16.cpu().numpy(), enc.encode_text(ids.cpu().numpy()))


def test_chunked_batches_equal_unchunked(tiny):
    """B larger than the workspace batch goes through the two-slot pinned pipeline in chunks."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    pix, ids = normalise_u8_nhwc(synth_pixels_u8(7, seed=5)), synth_tokens(7, seed=6)
    os.environ["CLIPX_MAX_BATCH"] = "2"
    try:
        small = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    finally:
        os.environ.pop("CLIPX_MAX_BATCH")
    assert small.max_batch == 2
    assert np.array_equal(small.encode_image(pix), enc.encode_image(pix))
    assert np.array_equal(small.encode_text(ids), enc.encode_text(ids))
    small.close()


def test_last_block_on_the_pooled_rows_equals_the_full_block(tiny):
    """The encoder runs the last block's out-proj / LayerNorm 2 / MLP on the pooled rows only (token 0, the EOT token): the
    embedding reads nothing else of that block.  CLIPX_FULL_LAST_BLOCK=1 runs the block on every row; batches of two or more
    must give the same bytes (a GEMM row does not depend on the rows it travels with), the split-K single-query path the same
    embedding up to f32 summation order."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    os.environ["CLIPX_FULL_LAST_BLOCK"] = "1"
    try:
        full = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    finally:
        os.environ.pop("CLIPX_FULL_LAST_BLOCK")
    for B in (2, 5, 9, 33):
        pix, ids = normalise_u8_nhwc(synth_pixels_u8(B, seed=40 + B)), synth_tokens(B, seed=50 + B)
        ids[0, :] = 0
        ids[0, 3] = arch.vocab - 1  # EOT early in the caption: the pooled row is not the last one
        assert np.array_equal(enc.encode_image(pix), full.encode_image(pix)), f"{name} image B={B}"
        assert np.array_equal(enc.encode_text(ids), full.encode_text(ids)), f"{name} text B={B}"
    pix, ids = normalise_u8_nhwc(synth_pixels_u8(1, seed=3)), synth_tokens(1, seed=4)
    assert _cos(enc.encode_image(pix), full.encode_image(pix).astype(np.float32)).min() > 1 - 1e-5
    assert _cos(enc.encode_text(ids), full.encode_text(ids).astype(np.float32)).min() > 1 - 1e-5
    full.close()


def test_ragged_text_tower_equals_the_rectangular_one(tiny):
    """The text transformer is causal and the embedding is read at the EOT token, so rows after a caption's EOT are never read:
    batches above the hipGraph sizes run every layer on sum(eot + 1) rows.  Same kernels on a subset of the rows -- and a row
    does not depend on the rows it travels with -- so the embeddings must be the same BYTES as with CLIPX_RAGGED_TEXT=0, through
    host and device pointers, for captions of every length (EOT first, EOT last, no early maximum), and match the oracle."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import mapper_semantics, synth_tokens

    name, arch, oracle, enc = tiny
    os.environ["CLIPX_RAGGED_TEXT"] = "0"
    try:
        rect = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    finally:
        os.environ.pop("CLIPX_RAGGED_TEXT")
    for B in (9, 33, 70):
        ids = synth_tokens(B, seed=60 + B)
        ids[0, :] = 5
        ids[0, 0] = arch.vocab - 1          # EOT at position 0: one row
        ids[1, :] = 7
        ids[1, -1] = arch.vocab - 1         # EOT at the last position: all rows
        ids[2, :] = np.arange(arch.ctx_len) % 11 + 1
        ids[2, 20] = 400                    # no EOT token at all: the pooling rule is "highest id, first occurrence"
        ids[2, 40] = 400
        a, b = enc.encode_text(ids), rect.encode_text(ids)
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"{name} B={B}: ragged != rectangular (host ids)"
        dev_ids = torch.from_numpy(ids).cuda()
        o16 = torch.empty(B, arch.embed_dim, dtype=torch.float16, device="cuda")
        enc.encode_text_device(dev_ids.data_ptr(), B, o16.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(o16.cpu().numpy().view(np.uint16), b.view(np.uint16)), f"{name} B={B}: ragged != rectangular (device ids)"
        # device ids + the caller's host copy (clipx_encode_text_device_ids): no read-back, so the call also works on a stream that
        # is being captured into a hipGraph -- where the plain device entry falls back to the rectangular tower by itself
        o16.zero_()
        enc.encode_text_device(dev_ids.data_ptr(), B, o16.data_ptr(), None, torch.cuda.current_stream().cuda_stream, ids_host=ids)
        torch.cuda.synchronize()
        assert np.array_equal(o16.cpu().numpy().view(np.uint16), b.view(np.uint16)), f"{name} B={B}: ragged != rectangular (device + host ids)"
        for hint in (None, ids):
            side = torch.cuda.Stream()
            o2 = torch.empty_like(o16)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                enc.encode_text_device(dev_ids.data_ptr(), B, o2.data_ptr(), None, side.cuda_stream, ids_host=hint)  # warm-up outside the capture
                side.synchronize()
                o2.zero_()
                with torch.cuda.graph(g, stream=side):
                    enc.encode_text_device(dev_ids.data_ptr(), B, o2.data_ptr(), None, side.cuda_stream, ids_host=hint)
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(o2.cpu().numpy().view(np.uint16), b.view(np.uint16)), f"{name} B={B}: captured call (hint={'host ids' if hint is not None else 'none'})"
        _, want = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
        assert _cos(a, want).min() >= COS_BAR
    rect.close()


def test_small_batches_replayed_from_graphs_equal_the_plain_launches(tiny):
    """Batches of <= 8 (the B = 1 query encode of clip_back.py:207-255) are captured into a hipGraph per (tower, B, buffers)
    and replayed; a batch of 12 takes the plain launches.  Rows do not depend on the batch they travel in (bitwise), so
    six calls of B = 2 -- first a capture per staging slot, then replays with NEW inputs in the same buffers -- must
    reproduce the rows of the one plain call, and so must B = 1 calls through device buffers."""
    name, arch, oracle, enc = tiny
    from oracle.clip_oracle import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    pix, ids = normalise_u8_nhwc(synth_pixels_u8(12, seed=21)), synth_tokens(12, seed=22)
    want_i, want_t = enc.encode_image(pix), enc.encode_text(ids)
    for rep in range(2):
        got_i = np.concatenate([enc.encode_image(pix[o:o + 2]) for o in range(0, 12, 2)])
        got_t = np.concatenate([enc.encode_text(ids[o:o + 2]) for o in range(0, 12, 2)])
        assert np.array_equal(got_i, want_i) and np.array_equal(got_t, want_t), f"{name} pass {rep}"
    dpix = torch.from_numpy(pix).cuda()
    dids = torch.from_numpy(ids).cuda()
    out = torch.empty(1, arch.embed_dim, dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # B = 1 API calls (the query path) additionally split the K loops of their GEMMs over more workgroups: the same arithmetic
    # in another f32 summation order, so they are compared with the batch rows by cosine, and with THEMSELVES (first call =
    # capture, later calls = replays that read new contents from the same buffers) bit for bit
    first = {}
    for i in (3, 7, 3):
        one = dpix[i:i + 1].clone()
        for _ in range(2):
            one.copy_(dpix[i:i + 1])
            torch.cuda.synchronize()
            enc.encode_image_device(one.data_ptr(), 1, 0, out.data_ptr(), None, st)
            torch.cuda.synchronize()
            got = out.cpu().numpy()[0]
            assert _cos(got[None], want_i[i][None]).min() > 1 - 1e-5
            assert np.array_equal(first.setdefault(i, got.copy()), got)
    assert enc.graphs_cached() >= 3, "small batches did not go through captured graphs"
    tok = dids[0:1].clone()
    for i in (5, 1, 5, 9):
        tok.copy_(dids[i:i + 1])
        torch.cuda.synchronize()
        enc.encode_text_device(tok.data_ptr(), 1, out.data_ptr(), None, st)
        torch.cuda.synchronize()
        got = out.cpu().numpy()[0]
        assert _cos(got[None], want_t[i][None]).min() > 1 - 1e-5
        assert np.array_equal(first.setdefault(100 + i, got.copy()), got)


def test_mapper_vs_reference_clipmapper_golden(tiny):
    """The HIP ClipMapper against fp16 embeddings produced by the reference's own ClipMapper code on the CPU
    (tests/golden/make_golden_mapper.py): per-sample cosine >= 1 - 1e-3 (north_star bar), same shapes and dtype."""
    from clip_retrieval_amd.encoder import register_encoder
    from clip_retrieval_amd.mapper import ClipMapper
    from oracle.clip_oracle import normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_mapper_" + name.replace("/", "-") + ".npz"))
    B = int(g["batch"])
    register_encoder("golden-" + name, enc)
    m = ClipMapper(True, True, True, False, "registered:golden-" + name, False, "", warmup_batch_size=1)
    pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=int(g["pixel_seed"])))
    ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=int(g["token_seed"]))
    out = m({"image_tensor": torch.from_numpy(pix), "text_tokens": torch.from_numpy(ids), "image_filename": ["a"] * B,
             "text": ["b"] * B, "metadata": ["{}"] * B})
    for key in ("image_embs", "text_embs"):
        assert out[key].dtype == np.float16 and out[key].shape == g[key].shape
        parity_gate(out[key], g[key].astype(np.float32), f"{name} {key} vs the reference ClipMapper's fp16 rows")
        assert np.abs(out[key].astype(np.float32) - g[key].astype(np.float32)).max() < 2e-2


def test_clip_mapper_drop_in(tiny):
    """The reference's mapper test (tests/test_clip_inference/test_mapper.py:20-38: row count + float16) on batches
    of 2 and 1, plus the value check the reference never had."""
    from clip_retrieval_amd.encoder import register_encoder
    from clip_retrieval_amd.mapper import ClipMapper
    from oracle.clip_oracle import mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    register_encoder("tiny-under-test", enc)
    mapper = ClipMapper(enable_image=True, enable_text=True, enable_metadata=True, use_mclip=False,
                        clip_model="registered:tiny-under-test", use_jit=True, mclip_model="", warmup_batch_size=1)
    for B in (2, 1):
        pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(B, seed=20 + B)))
        ids = torch.from_numpy(synth_tokens(B, seed=30 + B)).long()
        item = {"image_tensor": pix, "text_tokens": ids, "image_filename": [f"{i}.jpg" for i in range(B)],
                "text": [f"t{i}" for i in range(B)], "metadata": ["{}"] * B}
        out = mapper(item)
        assert set(out) == {"image_embs", "text_embs", "image_filename", "text", "metadata"}
        assert out["image_embs"].shape[0] == B and out["image_embs"].dtype == np.float16
        assert out["text_embs"].shape[0] == B and out["text_embs"].dtype == np.float16
        assert out["image_filename"] == item["image_filename"] and out["text"] == item["text"]
        _, w32 = mapper_semantics(oracle.encode_image(pix))
        parity_gate(out["image_embs"], w32, "mapper image")
        _, w32 = mapper_semantics(oracle.encode_text(ids))
        parity_gate(out["text_embs"], w32, "mapper text")
    off = ClipMapper(True, False, False, False, "registered:tiny-under-test", True, "", warmup_batch_size=0)
    o = off({"image_tensor": pix, "image_filename": ["a"]})
    assert o["text_embs"] is None and o["metadata"] is None and o["image_embs"].shape == (1, arch.embed_dim)
    with pytest.raises(NotImplementedError):
        ClipMapper(True, True, False, True, "registered:tiny-under-test", True, "x")


def test_full_depth_vit_l14_parity():
    """The BASELINE config's model at full depth (24 + 12 layers), small batch: the oracle needs ~1 s per image."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    arch = ARCHS["ViT-L/14"]
    oracle = HFClipOracle(arch, seed=0)
    enc = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    pix = normalise_u8_nhwc(synth_pixels_u8(3, seed=1))
    ids = synth_tokens(6, seed=2)
    _, wi = mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))
    _, wt = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
    gi, gt = enc.encode_image(pix), enc.encode_text(ids)
    enc.close()
    parity_gate(gi, wi, "ViT-L/14 image")
    for wrong, want in ((gi[::-1], wi), (gt[::-1], wt), (np.repeat(gi[:1], len(gi), 0), wi)):  # swapped / stale rows must fail
        with pytest.raises(AssertionError):
            parity_gate(wrong, want, "negative")
    parity_gate(gt, wt, "ViT-L/14 text")


def test_full_depth_vit_h14_parity():
    """open_clip ViT-H/14 (BASELINE config 5's query encoder: erf GELU, 80-wide image heads, 32 + 24 layers) at FULL depth,
    B = 2 -- round 1 only compared the 2-layer tiny-H/14."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    arch = ARCHS["ViT-H/14"]
    oracle = HFClipOracle(arch, seed=0)
    enc = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    pix = normalise_u8_nhwc(synth_pixels_u8(2, seed=3))
    ids = synth_tokens(2, seed=4)
    _, wi = mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))
    _, wt = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
    gi, gt = enc.encode_image(pix), enc.encode_text(ids)
    enc.close()
    parity_gate(gi, wi, "ViT-H/14 image")
    for wrong, want in ((gi[::-1], wi), (gt[::-1], wt), (np.repeat(gi[:1], len(gi), 0), wi)):  # swapped / stale rows must fail
        with pytest.raises(AssertionError):
            parity_gate(wrong, want, "negative")
    parity_gate(gt, wt, "ViT-H/14 text")


@pytest.mark.parametrize("name,B,outliers", [("tiny-L/14", 4, (300.0, -300.0)), ("tiny-H/14", 3, (2000.0,)), ("tiny-H/14", 4, (300.0, -300.0)),
                                             ("ViT-L/14", 2, (300.0, -300.0)), ("ViT-H/14", 2, (2000.0,))])
def test_parity_with_trained_like_weight_statistics(name, B, outliers):
    """Random init gives benign activations; trained CLIP does not (massive-activation channels, wide LayerNorm gains, peaked
    softmax).  VERDICT r3 weak #4 asked for published magnitudes: two channels at +-300 (ViT-L/14), one at 2 000 (ViT-H/14, a
    bf16-trained checkpoint), LayerNorm gains up to 30 x, FULL depth for both.  The LayerNorm fold multiplies the UN-normalised
    fp16 stream by mean-centred fp16 weights, the residual adds round to fp16: both are exercised at those magnitudes here.
    Bar: the north-star 1e-3 on the raw cosine, nearest-oracle-row == own row, centred cosine >= 0.9 (these weights push every
    embedding onto a common direction: the closest WRONG row sits at 0.97 .. 0.995), and no range flag.
    Round 4 made q, k, v IEEE fp16 for this test: with bf16 q / k the 30 x gains cost tiny-H/14 1 - cos = 2e-3
    (tools/emulate_fp16_stream.py --exact qkv_bf16)."""
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import ARCHS, NORTH_STAR_BAR, HFClipOracle, mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    arch = ARCHS[name]
    oracle = HFClipOracle(arch, seed=5)
    oracle.make_trained_like(seed=5, outliers=outliers, gain=30.0)
    enc = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=11))
    ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=12)
    _, wi = mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))
    _, wt = mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))
    gi, gt = enc.encode_image(pix), enc.encode_text(ids)  # a range overflow would raise ResidualStreamOverflow here
    enc.close()
    ri = parity_gate(gi, wi, f"{name} image", bar=NORTH_STAR_BAR, centred_bar=0.9)
    rt = parity_gate(gt, wt, f"{name} text", bar=NORTH_STAR_BAR, centred_bar=0.9)
    print(f"{name} outliers {outliers}: image 1-cos {1 - ri['cos'].min():.2e} centred {ri['centred'].min():.4f}; text 1-cos {1 - rt['cos'].min():.2e}")
    for wrong, want in ((gi[::-1], wi), (gt[::-1], wt)):
        with pytest.raises(AssertionError):
            parity_gate(wrong, want, "negative", bar=NORTH_STAR_BAR, centred_bar=0.9)


@pytest.mark.parametrize("where", ["fc2_bias", "token_embedding", "ln_pre"])
def test_fp16_stream_overflow_is_reported_never_hidden(where):
    """The residual stream lives in IEEE fp16 (65 504).  A model that exceeds it must get CLIPX_E_RANGE, never a finite-looking wrong
    embedding: LayerNorm of a row holding inf has rstd = 0, i.e. a perfectly finite output.  Weights are
    planted so that the stream overflows (a 70 000 bias on the image tower's first MLP output / a 70 000 token embedding / a
    ln_pre gain of 1e6) and EVERY way into the encoder is checked: the host calls, tickets + clipx_wait, the *_device calls +
    clipx_range_check, the B = 1 hipGraph replays; the unaffected tower keeps working and the flag clears after it is read."""
    import clip_retrieval_amd
    from clip_retrieval_amd import ResidualStreamOverflow
    from clip_retrieval_amd.encoder import ClipEncoder
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics, normalise_u8_nhwc, parity_gate, synth_pixels_u8, synth_tokens

    arch = ARCHS["tiny-L/14"]
    oracle = HFClipOracle(arch, seed=0)
    sd = oracle.model.state_dict()
    with torch.no_grad():
        if where == "fc2_bias":
            sd["vision_model.encoder.layers.0.mlp.fc2.bias"][7] += 70000.0
        elif where == "ln_pre":
            sd["vision_model.pre_layrnorm.weight"][3] = 1.0e6
        else:
            sd["text_model.embeddings.token_embedding.weight"][:, 11] += 70000.0
    bad_img, bad_txt = where != "token_embedding", where == "token_embedding"
    enc = ClipEncoder(_product_arch(arch), oracle.export_blob(), 0)
    for B in (1, 3, 12):  # 1, 3: replayed from hipGraphs after the second call; 12: plain launches, ragged text
        pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=70 + B))
        ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=80 + B)
        for rep in range(3):
            for bad, call, arg in ((bad_img, enc.encode_image, pix), (bad_txt, enc.encode_text, ids)):
                if bad:
                    with pytest.raises(ResidualStreamOverflow):
                        call(arg)
                else:
                    call(arg)
        # tickets
        hi, ht = enc.submit_image(pix), enc.submit_text(ids)
        for bad, h in ((bad_img, hi), (bad_txt, ht)):
            if bad:
                with pytest.raises(ResidualStreamOverflow):
                    enc.collect(h)
            else:
                enc.collect(h)
        # device entry points + clipx_range_check
        dp, di = torch.from_numpy(pix).cuda(), torch.from_numpy(ids).cuda()
        o = torch.empty(B, arch.embed_dim, dtype=torch.float16, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        enc.check_range(st)  # nothing queued through the device entry points yet: clean
        for bad, fn in ((bad_img, lambda: enc.encode_image_device(dp.data_ptr(), B, 0, o.data_ptr(), None, st)),
                        (bad_txt, lambda: enc.encode_text_device(di.data_ptr(), B, o.data_ptr(), None, st))):
            fn()
            if bad:
                with pytest.raises(ResidualStreamOverflow):
                    enc.check_range(st)
            enc.check_range(st)  # read once, cleared
    # the tower whose weights are sane is still right, row for row
    pix = normalise_u8_nhwc(synth_pixels_u8(4, arch.image_size, seed=7))
    ids = synth_tokens(4, arch.ctx_len, arch.vocab, seed=8)
    if bad_img:
        parity_gate(enc.encode_text(ids), mapper_semantics(oracle.encode_text(torch.from_numpy(ids)))[1], "text tower beside an overflowing image tower")
    else:
        parity_gate(enc.encode_image(pix), mapper_semantics(oracle.encode_image(torch.from_numpy(pix)))[1], "image tower beside an overflowing text tower")
    enc.close()
    assert clip_retrieval_amd.load_library().clipx_last_error() is not None


def test_load_clip_facade_query_path(tiny, tmp_path):
    """`load_clip` -> (model, preprocess, tokenizer) as clip_back.py:865-868 / worker.py:52-57 use it: the B = 1 query
    encodes of KnnService.compute_query (clip_back.py:227-246) return FP32 unit-norm torch features (not fp16-rounded),
    equal to the oracle within the cosine bar; preprocess is a real callable; the tokenizer fails loudly without the
    merges file and works when one is supplied."""
    from PIL import Image

    from clip_retrieval_amd.encoder import load_clip, register_encoder
    from oracle.clip_oracle import mapper_semantics, normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    register_encoder("facade-" + name, enc)
    model, preprocess, tokenizer = load_clip("registered:facade-" + name, use_jit=False, warmup_batch_size=1, clip_cache_path=str(tmp_path))
    pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(1, seed=9)))
    ids = torch.from_numpy(synth_tokens(1, seed=8)).long()
    fi, ft = model.encode_image(pix), model.encode_text(ids)
    assert fi.dtype == torch.float32 and ft.dtype == torch.float32 and tuple(fi.shape) == (1, arch.embed_dim)
    assert torch.allclose(fi.norm(dim=-1), torch.ones(1), atol=1e-5)
    assert not torch.equal(fi, fi.half().float()), "features must not be fp16-rounded (the service feeds fp32 to the index)"
    assert torch.equal(fi.half(), torch.from_numpy(enc.encode_image(pix)))  # the fp16 path is the rounding of the same row
    _, wi = mapper_semantics(oracle.encode_image(pix))
    _, wt = mapper_semantics(oracle.encode_text(ids))
    assert _cos(fi.numpy(), wi).min() >= COS_BAR and _cos(ft.numpy(), wt).min() >= COS_BAR
    img = preprocess(Image.fromarray(synth_pixels_u8(1, seed=1)[0]))
    assert tuple(img.shape) == (3, arch.image_size, arch.image_size) and img.dtype == torch.float32
    with pytest.raises(FileNotFoundError):
        tokenizer(["a photo"])
    with pytest.raises(IndexError):
        enc.encode_text(np.full((1, arch.ctx_len), arch.vocab, dtype=np.int64))  # the reference's embedding raises too


def test_async_tickets_equal_the_synchronous_path(tiny):
    """clipx_encode_*_async + clipx_wait: four tickets in flight (image, text, image, text of two batches -- what the
    pipelined Runner submits) return bit-identical rows to the synchronous calls; a fifth ticket is refused; inputs may
    be page-locked (uploaded in place) or pageable."""
    from clip_retrieval_amd import HipLibraryError
    from oracle.clip_oracle import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    name, arch, oracle, enc = tiny
    pixs = [normalise_u8_nhwc(synth_pixels_u8(B, seed=40 + B)) for B in (3, 2)]
    idss = [synth_tokens(B, seed=50 + B) for B in (3, 2)]
    want = [(enc.encode_image(p), enc.encode_text(i)) for p, i in zip(pixs, idss)]
    pinned = torch.from_numpy(pixs[0]).pin_memory()
    hs = [enc.submit_image(pinned), enc.submit_text(idss[0]), enc.submit_image(pixs[1]), enc.submit_text(idss[1])]
    with pytest.raises(HipLibraryError):
        enc.submit_text(idss[1])
    got = [enc.collect(h) for h in hs]
    assert np.array_equal(got[0], want[0][0]) and np.array_equal(got[1], want[0][1])
    assert np.array_equal(got[2], want[1][0]) and np.array_equal(got[3], want[1][1])
    with pytest.raises(HipLibraryError):
        enc.collect(hs[0])
    h = enc.submit_text(idss[1])  # slots are free again
    assert np.array_equal(enc.collect(h), want[1][1])


def test_pipelined_runner_on_the_gpu(tiny, tmp_path):
    """Runner + FilesReader + ClipMapper (submit/collect on real tickets) + NumpyWriter over an image folder: the files
    equal those of the serial call path bit for bit."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from reader_fixture import make_folder

    from clip_retrieval_amd.encoder import register_encoder
    from clip_retrieval_amd.mapper import ClipMapper
    from clip_retrieval_amd.reader import FilesReader, HashTokenizer, clip_preprocess
    from clip_retrieval_amd.runner import NullLogger, Runner
    from clip_retrieval_amd.writer import NumpyWriter

    name, arch, oracle, enc = tiny
    folder = make_folder(str(tmp_path / "data"))
    register_encoder("runner-" + name, enc)
    prep = lambda im: clip_preprocess(im, arch.image_size)
    tok = HashTokenizer(arch.ctx_len, arch.vocab)
    outs = {}
    for mode in ("pipelined", "serial"):
        def mapper_builder(mode=mode):
            m = ClipMapper(True, True, False, False, "registered:runner-" + name, False, "", warmup_batch_size=1)
            return m if mode == "pipelined" else m.__call__
        out = tmp_path / mode
        Runner(lambda s: FilesReader(s, prep, tok, folder, 4, 2, enable_text=True, enable_image=True),
               mapper_builder, lambda i, out=out: NumpyWriter(i, str(out), True, True, False, 1), NullLogger, 1)(0)
        outs[mode] = (np.load(out / "img_emb" / "img_emb_0.npy"), np.load(out / "text_emb" / "text_emb_0.npy"))
    assert outs["serial"][0].shape[0] == 9
    assert np.array_equal(outs["pipelined"][0], outs["serial"][0]) and np.array_equal(outs["pipelined"][1], outs["serial"][1])


@pytest.mark.parametrize("M,N,K", [(16384, 1024, 1024), (65792, 1024, 4096), (19712, 768, 768), (1000, 1024, 1024)])
def test_gemm_layernorm_fold_hooks_both_kernels(lib, M, N, K):
    """The two extras of the LayerNorm-folded layers, on shapes that reach the persistent 256x256 kernel (the encoder parity
    tests run batches whose GEMMs all fit the 128x128 kernel): (1) residual epilogue: the bf16 shadow must be EXACTLY the
    bf16 rounding of the f32 rows it writes, and both kernels must agree bit for bit; (2) bf16 epilogues: out = act(acc *
    rowscale[m] + bias[n]) against torch fp32, and bit-identical between the kernels."""
    from clip_retrieval_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    x0 = torch.randn(M, N, generator=g, device="cuda")
    rs = torch.rand(M, generator=g, device="cuda") + 0.5
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for variant in (3, 1, 6):
        os.environ["CLIPX_GEMM_VARIANT"] = str(variant)
        x = x0.clone()
        x16 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        check(lib, lib.clipx_gemm_bf16_ex_device(0, _ptr(A), _ptr(W), _ptr(bias), _ptr(x), M, N, K, 3, None, _ptr(x16), C.c_void_p(st)), "clipx")
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        check(lib, lib.clipx_gemm_bf16_ex_device(0, _ptr(A), _ptr(W), _ptr(bias), _ptr(y), M, N, K, 0, _ptr(rs), None, C.c_void_p(st)), "clipx")
        torch.cuda.synchronize()
        bad = (x16.view(torch.int16) != x.to(torch.bfloat16).view(torch.int16)).nonzero()
        assert bad.numel() == 0, f"variant {variant}: shadow != bf16(x) at {bad[:5].tolist()} ({bad.shape[0]} elements)"
        outs[variant] = (x, x16, y)
    os.environ.pop("CLIPX_GEMM_VARIANT")
    for other in (1, 6):
        for a, b in zip(outs[3], outs[other]):
            assert torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16), b.view(torch.int32 if b.dtype == torch.float32 else torch.int16))
    ref = (A.float() @ W.float().T)
    want = ref * rs[:, None] + bias
    assert (outs[3][2].float() - want).abs().max() <= 2e-2 * max(1.0, float(want.abs().max()))
    assert torch.allclose(outs[3][0], x0 + ref + bias, atol=2e-3 * float(ref.abs().max()))


def test_large_batch_persistent_kernel_path_equals_128_path():
    """A 2-layer ViT-L/14-width model at batch 64 and 256: here the GEMMs run on the persistent 256x256 kernel (LayerNorm
    fold, shadow stores, raster 2) -- batch sizes the CPU oracle cannot follow in a test.  The same encoder forced onto the
    128x128 kernel (CLIPX_GEMM_VARIANT=1), which the oracle tests above validate, must give bit-identical embeddings."""
    from clip_retrieval_amd.encoder import ARCHS, ClipArch, ClipEncoder, random_blob
    from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    base = ARCHS["ViT-L/14"]
    arch = ClipArch(**{**{k: getattr(base, k) for k in ClipArch.__dataclass_fields__}, "v_layers": 2, "t_layers": 2})
    blob = random_blob(arch, seed=0)
    enc = ClipEncoder(arch, blob, 0)
    os.environ["CLIPX_GEMM_VARIANT"] = "1"
    try:
        ref = ClipEncoder(arch, blob, 0)
    finally:
        os.environ.pop("CLIPX_GEMM_VARIANT")
    os.environ["CLIPX_FULL_LAST_BLOCK"] = "1"
    try:
        full = ClipEncoder(arch, blob, 0)  # every row through the last block (the default pools it: see the test above)
    finally:
        os.environ.pop("CLIPX_FULL_LAST_BLOCK")
    for B in (64, 256):
        pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1))
        ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=2)
        a, b, c = enc.encode_image(pix), ref.encode_image(pix), full.encode_image(pix)
        assert not np.isnan(a.astype(np.float32)).any() and np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"image B={B}"
        assert np.array_equal(a.view(np.uint16), c.view(np.uint16)), f"image B={B}: pooled last block != full last block"
        a, b, c = enc.encode_text(ids), ref.encode_text(ids), full.encode_text(ids)
        assert not np.isnan(a.astype(np.float32)).any() and np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"text B={B}"
        assert np.array_equal(a.view(np.uint16), c.view(np.uint16)), f"text B={B}: pooled last block != full last block"
    enc.close()
    ref.close()
    full.close()


@pytest.mark.parametrize("M,N,K", [(65792, 3072, 1024), (10547, 2304, 768), (16384, 4096, 1024), (256, 4096, 1024), (1000, 1024, 1024),
                                   (33024, 1280, 1280)])
def test_gemm_with_layernorm_statistics_inside(lib, M, N, K):
    """Round 6, an experiment kept behind CLIPX_LN_FUSED=1 (it measured SLOWER: profiles/r06o_ab_ln_fused.log): the
    LayerNorm-folded GEMMs (QKV, fc1) take the row scale 1 / sqrt(var(x) + eps) from the A fragments of their own K loop
    (gemm256w4.hip, STATS) for the rows the 4-wave kernel multiplies; the other rows (ragged tail, small M, the other kernels) get
    it from the one-pass statistics kernel, which sums in the same canonical order.  (1) the statistics pass against torch;
    (2) the GEMM with the statistics inside equals -- bit for bit -- the GEMM given the pass's row scales, with every kernel variant;
    (3) against torch fp32 on the same operands."""
    from clip_retrieval_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(M + 5 * N)
    Ah = (torch.randn(M, K, generator=g, device="cuda") * 2.0 + 0.3).to(torch.float16)  # un-normalised rows with a mean
    Ah[:, 7] += 60.0                                                                     # and a 'massive activation' channel
    Ah[:, 1] += (torch.arange(M, device="cuda") % 113).to(torch.float16) * 0.02
    Wh = (torch.randn(N, K, generator=g, device="cuda") * 0.04).to(torch.float16)
    bias = torch.randn(N, generator=g, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    eps = 1e-5
    rstd = torch.empty(M, device="cuda")
    check(lib, lib.clipx_rowstats_device(0, _ptr(Ah), 2, _ptr(rstd), M, K, C.c_float(eps), C.c_void_p(st)), "clipx")  # 2: the canonical one-pass form
    torch.cuda.synchronize()
    want_r = 1.0 / torch.sqrt(Ah.double().var(dim=1, unbiased=False) + eps)
    assert torch.allclose(rstd.double(), want_r, rtol=2e-5, atol=0), float(((rstd.double() - want_r) / want_r).abs().max())
    ref = (Ah.float() @ Wh.float().T) * want_r.float()[:, None] + bias
    outs = {}
    for variant in (6, 3, 1):
        os.environ["CLIPX_GEMM_VARIANT"] = str(variant)
        for epi in (7, 1):
            dt = torch.float16 if epi == 7 else torch.bfloat16
            y_in = torch.empty(M, N, device="cuda", dtype=dt)
            scratch = torch.full((M,), float("nan"), device="cuda")
            check(lib, lib.clipx_gemm_f16_ln_device(0, _ptr(Ah), _ptr(Wh), _ptr(bias), _ptr(y_in), M, N, K, epi, _ptr(scratch), C.c_float(eps),
                                                    C.c_void_p(st)), "clipx")
            y_ex = torch.empty(M, N, device="cuda", dtype=dt)
            check(lib, lib.clipx_gemm_f16_device(0, _ptr(Ah), _ptr(Wh), _ptr(bias), _ptr(y_ex), M, N, K, epi, _ptr(rstd), C.c_void_p(st)), "clipx")
            torch.cuda.synchronize()
            bad = (y_in.view(torch.int16) != y_ex.view(torch.int16)).nonzero()
            assert bad.numel() == 0, f"variant {variant} epi {epi}: statistics inside != statistics pass at {bad[:5].tolist()} ({bad.shape[0]} elements)"
            outs[(variant, epi)] = y_in
            if variant == 6 and M >= 16384:  # the rows the 4-wave kernel took were never written by a pass
                assert bool(torch.isnan(scratch[0])) and bool(torch.isnan(scratch[255]))
    os.environ.pop("CLIPX_GEMM_VARIANT")
    for epi in (7, 1):
        for variant in (3, 1):
            assert torch.equal(outs[(6, epi)].view(torch.int16), outs[(variant, epi)].view(torch.int16)), (variant, epi)
        w = ref if epi == 7 else ref * torch.sigmoid(1.702 * ref)
        e = (outs[(6, epi)].float() - w).abs()
        t = (3e-4 + 6e-4 * w.abs()) if epi == 7 else (2e-3 + 4e-3 * w.abs())
        assert (e <= t).all(), f"epi {epi}: max err {float(e.max()):.4g}"


@pytest.mark.parametrize("M,N,K", [(16384, 1024, 1024), (65792, 1024, 4096), (19712, 768, 768), (1000, 1024, 1024), (257, 1280, 1280)])
def test_gemm_fp16_residual_stream_hooks_both_kernels(lib, M, N, K):
    """Round 3: the residual stream lives in IEEE fp16.  (1) epi 6, out16 = fp16(f32(out16) + acc + bias) in place (bf16
    operands): against torch fp32 on the same operands to half an fp16 ulp + accumulation noise, and bit-identical between the
    persistent 256x256 kernel and the 128x128 kernel; (2) fp16 operands (clipx_gemm_f16_device, the LayerNorm-folded QKV / fc1):
    out = act(acc * rowscale + bias) against torch fp32 -- bf16 out, and fp16 out for epi 7 (round 4: the QKV projection) to fp16's
    tolerance --, bit-identical between the kernels."""
    from clip_retrieval_amd._lib import check

    g = torch.Generator(device="cuda").manual_seed(M + 3 * N)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    Ah = (torch.randn(M, K, generator=g, device="cuda") * 3.0).to(torch.float16)   # an un-normalised residual stream
    Ah[:, 5] += 40.0                                                                # with a 'massive activation' channel
    Wh = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.float16)
    Ah[:, 0] += (torch.arange(M, device="cuda") % 97).to(torch.float16) * 0.01
    bias = torch.randn(N, generator=g, device="cuda")
    x0 = (torch.randn(M, N, generator=g, device="cuda") * 4).to(torch.float16)
    rs = torch.rand(M, generator=g, device="cuda") * 0.3 + 0.05
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for variant in (3, 1, 6):
        os.environ["CLIPX_GEMM_VARIANT"] = str(variant)
        x = x0.clone()
        check(lib, lib.clipx_gemm_bf16_ex_device(0, _ptr(A), _ptr(W), _ptr(bias), _ptr(x), M, N, K, 6, None, None, C.c_void_p(st)), "clipx")
        # (3) the LayerNorm statistics of the rows it wrote, as the next LayerNorm-folded GEMM gets them
        if N % 256 == 0:
            r_pass = torch.empty(M, device="cuda")
            check(lib, lib.clipx_rowstats_device(0, _ptr(x), 1, _ptr(r_pass), M, N, C.c_float(1e-5), C.c_void_p(st)), "clipx")
            torch.cuda.synchronize()
            want_r = 1.0 / torch.sqrt(x.float().var(dim=1, unbiased=False) + 1e-5)
            assert torch.allclose(r_pass, want_r, rtol=1e-4, atol=0)
        ys = []
        for epi in (0, 1, 2, 7):  # 7 (round 4): epi 0 with an IEEE fp16 output -- the QKV projection
            y = torch.empty(M, N, device="cuda", dtype=torch.float16 if epi == 7 else torch.bfloat16)
            check(lib, lib.clipx_gemm_f16_device(0, _ptr(Ah), _ptr(Wh), _ptr(bias), _ptr(y), M, N, K, epi, _ptr(rs), C.c_void_p(st)), "clipx")
            ys.append(y)
        torch.cuda.synchronize()
        outs[variant] = [x] + ys
    os.environ.pop("CLIPX_GEMM_VARIANT")
    for other in (1, 6):
        for i, (a, b) in enumerate(zip(outs[3], outs[other])):
            bad = (a.view(torch.int32 if a.dtype == torch.float32 else torch.int16) != b.view(torch.int32 if b.dtype == torch.float32 else torch.int16)).nonzero()
            assert bad.numel() == 0, f"output {i}: variants 3 and {other} differ at {bad[:5].tolist()} ({bad.shape[0]} elements)"
    want = x0.float() + (A.float() @ W.float().T + bias)
    err = (outs[3][0].float() - want).abs()
    tol = 1e-3 + 1.2e-3 * want.abs()   # fp16: 2^-11 relative rounding + fp32 accumulation-order noise
    assert (err <= tol).all(), f"fp16 residual: max err {float(err.max()):.4g} at {(err > tol).nonzero()[:4].tolist()}"
    ref = (Ah.float() @ Wh.float().T) * rs[:, None] + bias
    for epi, y in zip((0, 1, 2, 7), outs[3][1:]):
        w = ref if epi in (0, 7) else (ref * torch.sigmoid(1.702 * ref) if epi == 1 else torch.nn.functional.gelu(ref))
        e = (y.float() - w).abs()
        t = (3e-4 + 6e-4 * w.abs()) if epi == 7 else (2e-3 + 4e-3 * w.abs())  # fp16: 2^-11 relative, bf16: 2^-8
        assert y.dtype == (torch.float16 if epi == 7 else torch.bfloat16)
        assert (e <= t).all(), f"f16 operands epi {epi}: max err {float(e.max()):.4g}"
