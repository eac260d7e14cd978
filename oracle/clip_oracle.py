"""CPU oracle for the encode half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (clip-retrieval_amd/) never does and fails loudly without its HIP library.

What it restates (reference file:line)
  * `ClipMapper.__call__`  clip_retrieval/clip_inference/mapper.py:49-78 -- encode, `x /= x.norm(dim=-1,
    keepdim=True)` (no epsilon), `.cpu().to(torch.float16).numpy()`: `mapper_semantics()` below.
  * `model.encode_image` / `model.encode_text` -- the arithmetic lives in un-vendored third-party packages
    (`all_clip>=1.3.0,<2` -> `clip-anytorch>=2.5.0,<3` / `open-clip-torch>=2.0.0,<3` / `transformers`;
    requirements.txt:2,21,25,28; call sites mapper.py:36-43,57,65).  Two independent statements are kept:
      - `HFClipOracle`: runs `transformers.CLIPModel` itself -- the module `all_clip` instantiates for the
        reference's `hf_clip:` model names (tests/test_clip_inference/test_mapper.py:13 exercises one) --
        with seeded random weights, fp32, CPU.  (transformers/models/clip/modeling_clip.py)
      - `functional_encode_image/_text`: a plain-torch restatement of the published OpenAI CLIP graph
        (conv1 patch embed, class token, positional embedding, ln_pre, pre-LN residual blocks with
        QuickGELU or erf-GELU, ln_post on token 0 / ln_final on the EOT token, bias-free projection),
        taking the flat weight blob the HIP library consumes (include/clipx.h).
    tests/test_oracle_clip.py checks the two against each other.

PARITY PIN: tests/golden/reference_mapper_*.npz hold fp16 embeddings produced by the reference's own
`ClipMapper.__call__` (mapper.py loaded by file path and run unmodified in the build container by
tests/golden/make_golden_mapper.py; the absent `all_clip` wheel is stubbed by its hf_clip wrapper around
`transformers.CLIPModel`, weights = this module's seeded random init); tests/test_oracle.py requires this oracle to
reproduce them bit for bit, tests/test_clip_gpu.py holds the HIP path to cosine >= 1 - 1e-3 against them.
STILL UNPINNED: real-checkpoint numbers -- the reference's own tests assert only shape and dtype at this boundary
(tests/test_clip_inference/test_mapper.py:37-38) and its fixture pairs test_tensors/*.pkl -> test_embeddings/*.pkl
need the OpenAI ViT-B/32 checkpoint, which is not available offline.
"""

import math
from dataclasses import dataclass, asdict

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class ClipArch:
    """Architecture numbers `all_clip.load_clip(name)` resolves a model name to."""

    image_size: int = 224
    patch_size: int = 14
    v_width: int = 1024
    v_layers: int = 24
    v_heads: int = 16
    v_mlp: int = 4096
    ctx_len: int = 77
    vocab: int = 49408
    t_width: int = 768
    t_layers: int = 12
    t_heads: int = 12
    t_mlp: int = 3072
    embed_dim: int = 768
    act: str = "quick_gelu"  # "quick_gelu" (OpenAI) | "gelu" (open_clip LAION)
    ln_eps: float = 1e-5

    @property
    def v_tokens(self):
        return (self.image_size // self.patch_size) ** 2 + 1


ARCHS = {
    "ViT-B/32": ClipArch(patch_size=32, v_width=768, v_layers=12, v_heads=12, v_mlp=3072, t_width=512, t_heads=8,
                         t_mlp=2048, embed_dim=512),
    "ViT-B/16": ClipArch(patch_size=16, v_width=768, v_layers=12, v_heads=12, v_mlp=3072, t_width=512, t_heads=8,
                         t_mlp=2048, embed_dim=512),
    "ViT-L/14": ClipArch(),
    # reduced-depth shapes for fast tests (same widths/heads as the real towers, fewer layers)
    "ViT-H/14": ClipArch(v_width=1280, v_layers=32, v_heads=16, v_mlp=5120, t_width=1024, t_layers=24, t_heads=16,
                         t_mlp=4096, embed_dim=1024, act="gelu"),
    "tiny-L/14": ClipArch(v_layers=2, t_layers=2),
    "tiny-H/14": ClipArch(v_width=1280, v_layers=2, v_heads=16, v_mlp=5120, t_width=1024, t_layers=2, t_heads=16,
                          t_mlp=4096, embed_dim=1024, act="gelu"),
    "tiny-B/32": ClipArch(patch_size=32, v_width=768, v_layers=2, v_heads=12, v_mlp=3072, t_width=512, t_heads=8,
                          t_mlp=2048, t_layers=2, embed_dim=512),
}


# ----------------------------------------------------------------------------------------------
# transformers.CLIPModel (the reference's `hf_clip:` backend) with seeded random weights
# ----------------------------------------------------------------------------------------------
class HFClipOracle:
    def __init__(self, arch: ClipArch, seed: int = 0, threads: int = 0):
        from transformers import CLIPConfig, CLIPModel

        if threads:
            torch.set_num_threads(threads)
        self.arch = arch
        cfg = CLIPConfig(
            text_config=dict(vocab_size=arch.vocab, hidden_size=arch.t_width, intermediate_size=arch.t_mlp,
                             num_hidden_layers=arch.t_layers, num_attention_heads=arch.t_heads,
                             max_position_embeddings=arch.ctx_len, hidden_act=arch.act, layer_norm_eps=arch.ln_eps,
                             projection_dim=arch.embed_dim, eos_token_id=2, bos_token_id=0, pad_token_id=1),
            vision_config=dict(hidden_size=arch.v_width, intermediate_size=arch.v_mlp, num_hidden_layers=arch.v_layers,
                               num_attention_heads=arch.v_heads, image_size=arch.image_size, patch_size=arch.patch_size,
                               hidden_act=arch.act, layer_norm_eps=arch.ln_eps, projection_dim=arch.embed_dim),
            projection_dim=arch.embed_dim,
        )
        cfg._attn_implementation = "eager"
        torch.manual_seed(seed)
        self.model = CLIPModel(cfg).eval().float()
        # CLIPModel's default init leaves LayerNorm at (1, 0) and biases at 0; a parity oracle must
        # exercise those paths, so perturb them deterministically.
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for name, p in self.model.named_parameters():
                if name.endswith("bias"):
                    p.add_(torch.randn(p.shape, generator=g) * 0.02)
                elif "layer_norm" in name or "layernorm" in name or "layrnorm" in name:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)

    @torch.no_grad()
    def make_trained_like(self, seed: int = 0, outliers=(300.0, -300.0), gain: float = 30.0) -> None:
        """Push the seeded random weights towards the statistics of TRAINED CLIP towers, which random init does not have and
        which stress a reduced-precision operand path: LayerNorm gains spread over two orders of magnitude with 1 % of them
        `gain` x larger, LayerNorm biases of order 1, 'massive activation' channels (a constant of `outliers[i]` in the residual
        stream from the first block on -- published CLIP ViT-L/14 checkpoints carry two channels in the hundreds, 500 - 1000 x
        the other channels; round 3 planted +-30 .. 40), sharper attention (q, k scaled up) and a negative
        fc1 bias.  No real checkpoint is available offline; this is the closest stand-in.  Call before export_blob()."""
        g = torch.Generator().manual_seed(1000 + seed)
        for name, p in self.model.named_parameters():
            if ("layer_norm" in name or "layernorm" in name or "layrnorm" in name) and name.endswith("weight"):
                p.mul_(torch.exp(torch.randn(p.shape, generator=g) * 0.5))
                idx = torch.randperm(p.numel(), generator=g)[: max(1, p.numel() // 100)]
                p.view(-1)[idx] *= gain
            elif ("layer_norm" in name or "layernorm" in name or "layrnorm" in name) and name.endswith("bias"):
                p.add_(torch.randn(p.shape, generator=g) * 0.5)
            elif name.endswith("q_proj.weight") or name.endswith("k_proj.weight"):
                p.mul_(2.5)
            elif name.endswith("mlp.fc1.bias"):
                p.sub_(0.5)
            elif name.endswith("encoder.layers.0.mlp.fc2.bias"):
                c = torch.randperm(p.numel(), generator=g)[: len(outliers)]
                for ci, v in zip(c, outliers):
                    p[ci] += float(v)

    @torch.no_grad()
    def encode_image(self, pixel_values: torch.Tensor) -> torch.Tensor:
        out = self.model.get_image_features(pixel_values=pixel_values.float())
        return out if isinstance(out, torch.Tensor) else out.pooler_output

    @torch.no_grad()
    def encode_text(self, ids: torch.Tensor) -> torch.Tensor:
        out = self.model.get_text_features(input_ids=ids.long())
        return out if isinstance(out, torch.Tensor) else out.pooler_output

    @torch.no_grad()
    def load_blob(self, blob) -> None:
        """Overwrite this model's parameters from a flat blob (include/clipx.h order): lets bench.py time and
        check the oracle on exactly the weights the HIP encoder was given."""
        W = unpack_blob(blob, self.arch)
        sd = self.model.state_dict()
        a = self.arch

        def put(name, t):
            sd[name].copy_(t.reshape(sd[name].shape))

        def tower(prefix, layers):
            for l, L in enumerate(layers):
                p = f"{prefix}.encoder.layers.{l}."
                w = L["out_w"].shape[0]
                put(p + "layer_norm1.weight", L["ln1_w"]); put(p + "layer_norm1.bias", L["ln1_b"])
                for i, x in enumerate("qkv"):
                    put(p + f"self_attn.{x}_proj.weight", L["qkv_w"][i * w:(i + 1) * w])
                    put(p + f"self_attn.{x}_proj.bias", L["qkv_b"][i * w:(i + 1) * w])
                put(p + "self_attn.out_proj.weight", L["out_w"]); put(p + "self_attn.out_proj.bias", L["out_b"])
                put(p + "layer_norm2.weight", L["ln2_w"]); put(p + "layer_norm2.bias", L["ln2_b"])
                put(p + "mlp.fc1.weight", L["fc1_w"]); put(p + "mlp.fc1.bias", L["fc1_b"])
                put(p + "mlp.fc2.weight", L["fc2_w"]); put(p + "mlp.fc2.bias", L["fc2_b"])

        v, t = "vision_model", "text_model"
        put(v + ".embeddings.patch_embedding.weight", W["conv"]); put(v + ".embeddings.class_embedding", W["cls"])
        put(v + ".embeddings.position_embedding.weight", W["vpos"])
        put(v + ".pre_layrnorm.weight", W["ln_pre_w"]); put(v + ".pre_layrnorm.bias", W["ln_pre_b"])
        tower(v, W["vlayers"])
        put(v + ".post_layernorm.weight", W["ln_post_w"]); put(v + ".post_layernorm.bias", W["ln_post_b"])
        put("visual_projection.weight", W["vproj"])
        put(t + ".embeddings.token_embedding.weight", W["tok"]); put(t + ".embeddings.position_embedding.weight", W["tpos"])
        tower(t, W["tlayers"])
        put(t + ".final_layer_norm.weight", W["ln_final_w"]); put(t + ".final_layer_norm.bias", W["ln_final_b"])
        put("text_projection.weight", W["tproj"])
        assert a.embed_dim == W["vproj"].shape[0]

    def export_blob(self) -> np.ndarray:
        """Flat f32 weight blob in the order include/clipx.h documents."""
        sd = {k: v.detach().float().cpu() for k, v in self.model.state_dict().items()}
        return blob_from_hf_state_dict(sd, self.arch)


def blob_from_hf_state_dict(sd, arch: ClipArch) -> np.ndarray:
    parts = []

    def add(t):
        parts.append(np.ascontiguousarray(t.numpy() if hasattr(t, "numpy") else t, dtype=np.float32).reshape(-1))

    def layers(prefix, n):
        for l in range(n):
            p = f"{prefix}.encoder.layers.{l}."
            add(sd[p + "layer_norm1.weight"]); add(sd[p + "layer_norm1.bias"])
            add(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0))
            add(torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]], 0))
            add(sd[p + "self_attn.out_proj.weight"]); add(sd[p + "self_attn.out_proj.bias"])
            add(sd[p + "layer_norm2.weight"]); add(sd[p + "layer_norm2.bias"])
            add(sd[p + "mlp.fc1.weight"]); add(sd[p + "mlp.fc1.bias"])
            add(sd[p + "mlp.fc2.weight"]); add(sd[p + "mlp.fc2.bias"])

    v = "vision_model"
    add(sd[v + ".embeddings.patch_embedding.weight"].reshape(arch.v_width, -1))
    add(sd[v + ".embeddings.class_embedding"])
    add(sd[v + ".embeddings.position_embedding.weight"])
    add(sd[v + ".pre_layrnorm.weight"]); add(sd[v + ".pre_layrnorm.bias"])
    layers(v, arch.v_layers)
    add(sd[v + ".post_layernorm.weight"]); add(sd[v + ".post_layernorm.bias"])
    add(sd["visual_projection.weight"])
    t = "text_model"
    add(sd[t + ".embeddings.token_embedding.weight"])
    add(sd[t + ".embeddings.position_embedding.weight"])
    layers(t, arch.t_layers)
    add(sd[t + ".final_layer_norm.weight"]); add(sd[t + ".final_layer_norm.bias"])
    add(sd["text_projection.weight"])
    return np.concatenate(parts)


# ----------------------------------------------------------------------------------------------
# plain-torch restatement of the OpenAI CLIP graph over the flat blob
# ----------------------------------------------------------------------------------------------
class _BlobReader:
    def __init__(self, blob):
        self.b = torch.from_numpy(np.asarray(blob, dtype=np.float32))
        self.o = 0

    def take(self, *shape):
        n = int(np.prod(shape))
        t = self.b[self.o:self.o + n].reshape(*shape)
        self.o += n
        return t


def _read_tower(r, width, mlp, n_layers):
    L = []
    for _ in range(n_layers):
        L.append(dict(ln1_w=r.take(width), ln1_b=r.take(width), qkv_w=r.take(3 * width, width), qkv_b=r.take(3 * width),
                      out_w=r.take(width, width), out_b=r.take(width), ln2_w=r.take(width), ln2_b=r.take(width),
                      fc1_w=r.take(mlp, width), fc1_b=r.take(mlp), fc2_w=r.take(width, mlp), fc2_b=r.take(width)))
    return L


def unpack_blob(blob, arch: ClipArch):
    r = _BlobReader(blob)
    P = arch.patch_size
    W = dict()
    W["conv"] = r.take(arch.v_width, 3 * P * P)
    W["cls"] = r.take(arch.v_width)
    W["vpos"] = r.take(arch.v_tokens, arch.v_width)
    W["ln_pre_w"] = r.take(arch.v_width); W["ln_pre_b"] = r.take(arch.v_width)
    W["vlayers"] = _read_tower(r, arch.v_width, arch.v_mlp, arch.v_layers)
    W["ln_post_w"] = r.take(arch.v_width); W["ln_post_b"] = r.take(arch.v_width)
    W["vproj"] = r.take(arch.embed_dim, arch.v_width)
    W["tok"] = r.take(arch.vocab, arch.t_width)
    W["tpos"] = r.take(arch.ctx_len, arch.t_width)
    W["tlayers"] = _read_tower(r, arch.t_width, arch.t_mlp, arch.t_layers)
    W["ln_final_w"] = r.take(arch.t_width); W["ln_final_b"] = r.take(arch.t_width)
    W["tproj"] = r.take(arch.embed_dim, arch.t_width)
    assert r.o == r.b.numel(), (r.o, r.b.numel())
    return W


def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    return torch.nn.functional.gelu(x)


def _blocks(x, layers, heads, eps, act, causal):
    B, T, w = x.shape
    dh = w // heads
    mask = None
    if causal:
        mask = torch.full((T, T), float("-inf")).triu_(1)
    for L in layers:
        h = torch.nn.functional.layer_norm(x, (w,), L["ln1_w"], L["ln1_b"], eps)
        qkv = h @ L["qkv_w"].T + L["qkv_b"]
        q, k, v = qkv.split(w, dim=-1)
        q = q.reshape(B, T, heads, dh).transpose(1, 2)
        k = k.reshape(B, T, heads, dh).transpose(1, 2)
        v = v.reshape(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        if mask is not None:
            s = s + mask
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, w)
        x = x + a @ L["out_w"].T + L["out_b"]
        h = torch.nn.functional.layer_norm(x, (w,), L["ln2_w"], L["ln2_b"], eps)
        x = x + _act(h @ L["fc1_w"].T + L["fc1_b"], act) @ L["fc2_w"].T + L["fc2_b"]
    return x


@torch.no_grad()
def functional_encode_image(W, arch: ClipArch, pixels: torch.Tensor) -> torch.Tensor:
    B = pixels.shape[0]
    P, g, w = arch.patch_size, arch.image_size // arch.patch_size, arch.v_width
    # conv1 with kernel = stride = P, no bias == matmul over unfolded patches (k = c*P*P + iy*P + ix)
    patches = pixels.float().reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    x = patches @ W["conv"].T
    x = torch.cat([W["cls"].expand(B, 1, w), x], dim=1) + W["vpos"]
    x = torch.nn.functional.layer_norm(x, (w,), W["ln_pre_w"], W["ln_pre_b"], arch.ln_eps)
    x = _blocks(x, W["vlayers"], arch.v_heads, arch.ln_eps, arch.act, causal=False)
    x = torch.nn.functional.layer_norm(x[:, 0], (w,), W["ln_post_w"], W["ln_post_b"], arch.ln_eps)
    return x @ W["vproj"].T


@torch.no_grad()
def functional_encode_text(W, arch: ClipArch, ids: torch.Tensor) -> torch.Tensor:
    ids = ids.long()
    w = arch.t_width
    x = W["tok"][ids] + W["tpos"]
    x = _blocks(x, W["tlayers"], arch.t_heads, arch.ln_eps, arch.act, causal=True)
    x = torch.nn.functional.layer_norm(x, (w,), W["ln_final_w"], W["ln_final_b"], arch.ln_eps)
    x = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]  # EOT = highest token id
    return x @ W["tproj"].T


def mapper_semantics(features: torch.Tensor):
    """mapper.py:58-59 / 66-67: divide by the L2 norm (no eps), cast fp16, numpy.  Also returns the
    fp32 normalised rows (what parity cosines are measured on)."""
    f = features.float()
    f = f / f.norm(dim=-1, keepdim=True)
    return f.to(torch.float16).numpy(), f.numpy()


# ----------------------------------------------------------------------------------------------
# the acceptance gate of the encode half (tests/, bench.py)
# ----------------------------------------------------------------------------------------------
NORTH_STAR_BAR = 1.0 - 1e-3   # BASELINE.json north_star: cosine to the reference CPU encoder within 1e-3
TIGHT_BAR = 1.0 - 1e-4        # what the kernels actually deliver is 1 - 4e-6 .. 1 - 3e-5; a gate at 1e-3 sits inside the
                              # inter-sample spread of some inputs, this one does not
CENTRED_BAR = 0.99


def _unit(a):
    a = np.asarray(a, dtype=np.float64)
    return a / np.maximum(np.linalg.norm(a, axis=-1, keepdims=True), 1e-300)


def parity_report(got, want_f32):
    """Per-row evidence that `got` [B, d] (any float dtype) is the oracle's `want_f32` [B, d] ROW FOR ROW:
      cos      raw cosine per row;
      centred  cosine after subtracting the oracle's batch-mean embedding from both (embeddings of one model share a large
               common component -- 0.7 .. 0.98 of the norm here -- which a raw cosine rewards whatever the row is); None for B = 1;
      nearest  for every row of `got` the index of the oracle row it is closest to (must be its own index);
      other    the largest cosine of a row of `got` to an oracle row that is NOT its own (how far a wrong row would be)."""
    g, w = _unit(got), _unit(want_f32)
    B = g.shape[0]
    c = g @ w.T
    rep = dict(cos=np.diag(c).copy(), nearest=c.argmax(axis=1), centred=None, other=None)
    if B > 1:
        mu = w.mean(axis=0, keepdims=True)
        rep["centred"] = (_unit(g - mu) * _unit(w - mu)).sum(-1)
        rep["other"] = (c - 2.0 * np.eye(B)).max(axis=1)
    return rep


def parity_gate(got, want_f32, what="", bar=TIGHT_BAR, centred_bar=CENTRED_BAR):
    """Raises AssertionError unless every row passes: finite, raw cosine >= bar, centred cosine >= centred_bar, nearest oracle
    row == own row.  A row-swapped, stale or input-independent output fails (tests/test_oracle.py::test_parity_gate_*)."""
    rep = parity_report(got, want_f32)
    assert np.isfinite(np.asarray(got, dtype=np.float64)).all(), f"{what}: non-finite values"
    assert rep["cos"].min() >= bar, f"{what}: cosine {rep['cos'].min():.7f} < {bar} (rows {np.nonzero(rep['cos'] < bar)[0][:8]})"
    B = len(rep["cos"])
    assert (rep["nearest"] == np.arange(B)).all(), f"{what}: rows {np.nonzero(rep['nearest'] != np.arange(B))[0][:8]} are closer to another row's oracle embedding"
    if rep["centred"] is not None:
        assert rep["centred"].min() >= centred_bar, f"{what}: centred cosine {rep['centred'].min():.5f} < {centred_bar}"
    return rep


# ----------------------------------------------------------------------------------------------
# synthetic inputs of SURVEY 8(d) config 2
# ----------------------------------------------------------------------------------------------
def synth_pixels_u8(B, size=224, seed=1):
    """Structured synthetic images (i.i.d. noise images all embed to the same point --
    inter-sample oracle cosine 0.998 -- so a cosine gate could not tell one row from another).  Sample b depends on (seed, b)
    only: a saturated background colour (a corner of the RGB cube, all eight distinct within each aligned group of eight
    samples, pulled up to 15 % towards a random colour) with a gradient, a low-frequency sinusoid, 2 - 5 filled rectangles /
    ellipses of random colours, mild per-sample noise.  Random-init CLIP towers are close to position-invariant colour
    statistics, so the dominant colour is what separates embeddings: mean inter-sample cosine ~0.72, < 0.93 within a group
    of eight (tests/test_oracle.py)."""
    out = np.empty((B, size, size, 3), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    u, v = xx / size, yy / size
    for b in range(B):
        r = np.random.default_rng([int(seed), b, 0x5EED])
        corner = int(np.random.default_rng([int(seed), b // 8, 0xC0]).permutation(8)[b % 8])
        c0 = np.asarray([corner & 1, (corner >> 1) & 1, (corner >> 2) & 1], dtype=np.float64) * 255.0
        c0 = c0 + r.uniform(0, 0.15) * (r.uniform(0, 255, 3) - c0)
        c1 = c0 + r.uniform(0.1, 0.3) * (r.uniform(0, 255, 3) - c0)
        ang = r.uniform(0, 2 * np.pi)
        t = (np.cos(ang) * (u - 0.5) + np.sin(ang) * (v - 0.5)) * 1.2 + 0.5
        img = c0 + np.clip(t, 0, 1)[..., None] * (c1 - c0)
        fx, fy, ph = r.uniform(0.5, 6), r.uniform(0.5, 6), r.uniform(0, 2 * np.pi)
        img = img + (r.uniform(5, 30) * np.sin(2 * np.pi * (fx * u + fy * v) + ph))[..., None] * r.uniform(-1, 1, 3)
        for _ in range(int(r.integers(2, 6))):
            cx, cy = r.uniform(0, 1, 2)
            hw, hh = r.uniform(0.04, 0.2, 2)
            col = r.uniform(0, 255, 3)
            if r.random() < 0.5:
                m = (np.abs(u - cx) < hw) & (np.abs(v - cy) < hh)
            else:
                m = ((u - cx) / hw) ** 2 + ((v - cy) / hh) ** 2 < 1.0
            img[m] = col
        img = img + r.normal(0, r.uniform(1, 10), (size, size, 3))
        out[b] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def normalise_u8_nhwc(u8: np.ndarray) -> np.ndarray:
    """u8 NHWC -> the reference's `image_tensor`: f32 NCHW, /255, CLIP mean/std."""
    x = u8.astype(np.float32) / np.float32(255.0)  # torchvision: ToTensor, then Normalize -- two f32 divisions
    x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def synth_tokens(B: int, ctx_len: int = 77, vocab: int = 49408, seed: int = 2) -> np.ndarray:
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(B, ctx_len, dtype=torch.int32)
    lens = torch.randint(4, ctx_len - 1, (B,), generator=g)
    body = torch.randint(1, vocab - 3, (B, ctx_len), generator=g, dtype=torch.int32)
    for b in range(B):
        L = int(lens[b])
        ids[b, 0] = vocab - 2           # SOT 49406
        ids[b, 1:L] = body[b, 1:L]
        ids[b, L] = vocab - 1           # EOT 49407 (the maximum id -> argmax pooling)
    return ids.numpy()


def arch_to_desc_dict(arch: ClipArch) -> dict:
    d = asdict(arch)
    d["act"] = 0 if arch.act == "quick_gelu" else 1
    return d


FLOPS = {  # algorithmic GFLOP per sample: 2*MACs, no padding
    "ViT-L/14": dict(image=162.03, text=13.30),
    "ViT-B/32": dict(image=8.82, text=5.96),
}


def tower_gflop(arch: ClipArch):
    def tower(T, w, mlp, layers):
        per_layer = 2 * T * (3 * w * w + w * w + 2 * w * mlp) + 4 * T * T * w
        return layers * per_layer
    g2 = (arch.image_size // arch.patch_size) ** 2
    img = tower(arch.v_tokens, arch.v_width, arch.v_mlp, arch.v_layers) + 2 * g2 * 3 * arch.patch_size ** 2 * arch.v_width \
        + 2 * arch.v_width * arch.embed_dim
    txt = tower(arch.ctx_len, arch.t_width, arch.t_mlp, arch.t_layers) + 2 * arch.t_width * arch.embed_dim
    return img / 1e9, txt / 1e9
