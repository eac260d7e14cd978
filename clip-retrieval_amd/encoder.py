"""Host wrapper of the CLIP encoder C ABI (include/clipx.h): model descriptions, weight-blob assembly
from the checkpoint formats `all_clip.load_clip` understands, and a `load_clip`-shaped factory.

Reference seams mirrored here
  * `load_clip(clip_model, use_jit, warmup_batch_size, clip_cache_path[, device]) -> (model, preprocess,
    tokenizer)` as called at clip_retrieval/clip_inference/mapper.py:36-41, worker.py:52-57 and
    clip_back.py:865-868; `model.encode_image(x)` / `model.encode_text(tok)` as used at mapper.py:57,65
    and clip_back.py:230,244.
All arithmetic happens in lib/libclipx.so on the GPU; this file only moves bytes and reshapes weights.
"""

import ctypes as C
import os
import threading
from dataclasses import dataclass

import numpy as np

from ._lib import ClipxModelDesc, HipLibraryError, check, load_library

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PIX_F32_NCHW, PIX_U8_NHWC = 0, 1


@dataclass(frozen=True)
class ClipArch:
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
    act: str = "quick_gelu"
    ln_eps: float = 1e-5

    @property
    def v_tokens(self):
        return (self.image_size // self.patch_size) ** 2 + 1

    def to_desc(self):
        d = ClipxModelDesc()
        for f in ("image_size", "patch_size", "v_width", "v_layers", "v_heads", "v_mlp", "ctx_len", "vocab", "t_width",
                  "t_layers", "t_heads", "t_mlp", "embed_dim"):
            setattr(d, f, int(getattr(self, f)))
        d.act = 0 if self.act == "quick_gelu" else 1
        d.ln_eps = float(self.ln_eps)
        d.pix_mean = (C.c_float * 3)(*CLIP_MEAN)
        d.pix_std = (C.c_float * 3)(*CLIP_STD)
        return d


# the model names the reference passes as `clip_model` (main.py:88 default "ViT-B/32"; docs use L/14)
ARCHS = {
    "ViT-B/32": ClipArch(patch_size=32, v_width=768, v_layers=12, v_heads=12, v_mlp=3072, t_width=512, t_heads=8,
                         t_mlp=2048, embed_dim=512),
    "ViT-B/16": ClipArch(patch_size=16, v_width=768, v_layers=12, v_heads=12, v_mlp=3072, t_width=512, t_heads=8,
                         t_mlp=2048, embed_dim=512),
    "ViT-L/14": ClipArch(),
    "open_clip:ViT-B-32": ClipArch(patch_size=32, v_width=768, v_layers=12, v_heads=12, v_mlp=3072, t_width=512,
                                   t_heads=8, t_mlp=2048, embed_dim=512, act="gelu"),
    "open_clip:ViT-L-14": ClipArch(act="gelu"),
    # LAION ViT-H/14 (the 1024-d model of docs/laion5B_h14_back.md; BASELINE config 5): image heads are 80 wide
    "open_clip:ViT-H-14": ClipArch(v_width=1280, v_layers=32, v_heads=16, v_mlp=5120, t_width=1024, t_layers=24,
                                   t_heads=16, t_mlp=4096, embed_dim=1024, act="gelu"),
}


def blob_floats(arch: ClipArch) -> int:
    lib = load_library()
    d = arch.to_desc()
    return int(lib.clipx_blob_floats(C.byref(d)))


# ----------------------------------------------------------------------------------------------
# weight blobs
# ----------------------------------------------------------------------------------------------
def _cat(parts):
    return np.concatenate([np.ascontiguousarray(p, dtype=np.float32).reshape(-1) for p in parts])


def _np(t):
    return t.detach().float().cpu().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32)


def blob_from_openai_state_dict(sd, arch: ClipArch) -> np.ndarray:
    """OpenAI `clip` / open_clip state-dict names (visual.conv1.weight, transformer.resblocks.N.attn.in_proj_weight,
    ...).  `visual.proj` and `text_projection` are stored [width, embed] there and are transposed to the
    library's [embed, width]."""
    parts = []

    def tower(prefix, n):
        for l in range(n):
            p = f"{prefix}.resblocks.{l}."
            for k in ("ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                      "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                      "mlp.c_proj.weight", "mlp.c_proj.bias"):
                parts.append(_np(sd[p + k]))

    parts.append(_np(sd["visual.conv1.weight"]).reshape(arch.v_width, -1))
    parts.append(_np(sd["visual.class_embedding"]))
    parts.append(_np(sd["visual.positional_embedding"]))
    parts += [_np(sd["visual.ln_pre.weight"]), _np(sd["visual.ln_pre.bias"])]
    tower("visual.transformer", arch.v_layers)
    parts += [_np(sd["visual.ln_post.weight"]), _np(sd["visual.ln_post.bias"])]
    parts.append(_np(sd["visual.proj"]).T)
    parts.append(_np(sd["token_embedding.weight"]))
    parts.append(_np(sd["positional_embedding"]))
    tower("transformer", arch.t_layers)
    parts += [_np(sd["ln_final.weight"]), _np(sd["ln_final.bias"])]
    parts.append(_np(sd["text_projection"]).T)
    return _cat(parts)


def blob_from_hf_state_dict(sd, arch: ClipArch) -> np.ndarray:
    """HF `transformers.CLIPModel` names (the reference's `hf_clip:` models); q/k/v are stacked into the
    in_proj layout."""
    parts = []

    def tower(prefix, n):
        for l in range(n):
            p = f"{prefix}.encoder.layers.{l}."
            parts.extend([_np(sd[p + "layer_norm1.weight"]), _np(sd[p + "layer_norm1.bias"])])
            parts.append(np.concatenate([_np(sd[p + f"self_attn.{x}_proj.weight"]) for x in "qkv"], 0))
            parts.append(np.concatenate([_np(sd[p + f"self_attn.{x}_proj.bias"]) for x in "qkv"], 0))
            parts.extend([_np(sd[p + "self_attn.out_proj.weight"]), _np(sd[p + "self_attn.out_proj.bias"]),
                          _np(sd[p + "layer_norm2.weight"]), _np(sd[p + "layer_norm2.bias"]),
                          _np(sd[p + "mlp.fc1.weight"]), _np(sd[p + "mlp.fc1.bias"]),
                          _np(sd[p + "mlp.fc2.weight"]), _np(sd[p + "mlp.fc2.bias"])])

    v, t = "vision_model", "text_model"
    parts.append(_np(sd[v + ".embeddings.patch_embedding.weight"]).reshape(arch.v_width, -1))
    parts.append(_np(sd[v + ".embeddings.class_embedding"]))
    parts.append(_np(sd[v + ".embeddings.position_embedding.weight"]))
    parts += [_np(sd[v + ".pre_layrnorm.weight"]), _np(sd[v + ".pre_layrnorm.bias"])]
    tower(v, arch.v_layers)
    parts += [_np(sd[v + ".post_layernorm.weight"]), _np(sd[v + ".post_layernorm.bias"])]
    parts.append(_np(sd["visual_projection.weight"]))
    parts.append(_np(sd[t + ".embeddings.token_embedding.weight"]))
    parts.append(_np(sd[t + ".embeddings.position_embedding.weight"]))
    tower(t, arch.t_layers)
    parts += [_np(sd[t + ".final_layer_norm.weight"]), _np(sd[t + ".final_layer_norm.bias"])]
    parts.append(_np(sd["text_projection.weight"]))
    return _cat(parts)


def random_blob(arch: ClipArch, seed: int = 0) -> np.ndarray:
    """Random-init weights of the right shapes (benchmarks: no checkpoints offline).  Scales follow the
    published CLIP initialisation (std width^-0.5 attention/proj, (2 width)^-0.5 fc, 0.02 embeddings).
    Filled in place, segment by segment, straight into one preallocated blob."""
    import torch  # pylint: disable=import-outside-toplevel

    segs = []  # (count, std, offset_value)

    def tower(w, mlp, n):
        attn_std, proj_std, fc_std = w ** -0.5, (w ** -0.5) * ((2 * n) ** -0.5), (2 * w) ** -0.5
        for _ in range(n):
            segs.extend([(w, 0.05, 1.0), (w, 0.02, 0.0), (3 * w * w, attn_std, 0.0), (3 * w, 0.02, 0.0),
                         (w * w, proj_std, 0.0), (w, 0.02, 0.0), (w, 0.05, 1.0), (w, 0.02, 0.0),
                         (mlp * w, fc_std, 0.0), (mlp, 0.02, 0.0), (w * mlp, proj_std, 0.0), (w, 0.02, 0.0)])

    w, tw, P, E = arch.v_width, arch.t_width, arch.patch_size, arch.embed_dim
    segs.extend([(w * 3 * P * P, 0.02, 0.0), (w, w ** -0.5, 0.0), (arch.v_tokens * w, 0.02, 0.0), (w, 0.05, 1.0), (w, 0.02, 0.0)])
    tower(w, arch.v_mlp, arch.v_layers)
    segs.extend([(w, 0.05, 1.0), (w, 0.02, 0.0), (E * w, w ** -0.5, 0.0)])
    segs.extend([(arch.vocab * tw, 0.02, 0.0), (arch.ctx_len * tw, 0.01, 0.0)])
    tower(tw, arch.t_mlp, arch.t_layers)
    segs.extend([(tw, 0.05, 1.0), (tw, 0.02, 0.0), (E * tw, tw ** -0.5, 0.0)])
    total = sum(c for c, _, _ in segs)
    g = torch.Generator().manual_seed(seed)
    blob = torch.empty(total, dtype=torch.float32)
    blob.normal_(0.0, 1.0, generator=g)
    o = 0
    for c, std, off in segs:
        v = blob[o:o + c]
        v.mul_(std)
        if off:
            v.add_(off)
        o += c
    return blob.numpy()


def blob_from_checkpoint(path: str, arch: ClipArch) -> np.ndarray:
    """A local checkpoint file: raw f32 blob (.bin/.npy), HF (.safetensors / state dict) or OpenAI/open_clip
    state dict (.pt).  TorchScript archives (OpenAI's published files) are read through torch.jit.load."""
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32).reshape(-1)
    if path.endswith(".bin") and os.path.getsize(path) == 4 * blob_floats(arch):
        return np.fromfile(path, dtype=np.float32)
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file  # pylint: disable=import-outside-toplevel

        sd = load_file(path)
    else:
        import torch  # pylint: disable=import-outside-toplevel

        try:
            sd = torch.jit.load(path, map_location="cpu").state_dict()
        except RuntimeError:
            sd = torch.load(path, map_location="cpu", weights_only=True)
            sd = sd.get("state_dict", sd)
    sd = _strip_prefixes(dict(sd))
    if any(k.startswith("vision_model.") for k in sd):
        return blob_from_hf_state_dict(sd, arch)
    return blob_from_openai_state_dict(sd, arch)


# ----------------------------------------------------------------------------------------------
# the encoder object
# ----------------------------------------------------------------------------------------------
class ClipEncoder:
    """One CLIP model resident on one GPU.  `encode_*` return fresh fp16 numpy arrays (unit L2 norm)."""

    def __init__(self, arch: ClipArch, blob: np.ndarray, device: int = 0):
        self._lib = load_library()
        self.arch = arch
        self.device = int(device)
        blob = np.ascontiguousarray(blob, dtype=np.float32).reshape(-1)
        desc = arch.to_desc()
        want = int(self._lib.clipx_blob_floats(C.byref(desc)))
        if blob.size != want:
            raise ValueError(f"weight blob has {blob.size} floats, model needs {want}")
        h = C.c_void_p()
        check(self._lib, self._lib.clipx_create(C.byref(desc), blob.ctypes.data, blob.size, self.device, C.byref(h)), "clipx")
        self._h = h
        self.embed_dim = arch.embed_dim
        self.max_batch = int(self._lib.clipx_max_batch(self._h))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.clipx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    # ---- host buffers (the ClipMapper path)
    def _image_array(self, pixels):
        a = pixels.numpy() if hasattr(pixels, "numpy") else np.asarray(pixels)
        S = self.arch.image_size
        if a.dtype == np.uint8:
            if a.ndim != 4 or a.shape[1:] != (S, S, 3):
                raise ValueError(f"uint8 images must be [B,{S},{S},3] (NHWC), got {a.shape}")
            fmt = PIX_U8_NHWC
        else:
            if a.ndim != 4 or a.shape[1:] != (3, S, S):
                raise ValueError(f"float images must be [B,3,{S},{S}] (NCHW), got {a.shape}")
            a = a.astype(np.float32, copy=False)
            fmt = PIX_F32_NCHW
        return np.ascontiguousarray(a), fmt

    def _token_array(self, ids):
        a = ids.numpy() if hasattr(ids, "numpy") else np.asarray(ids)
        if a.ndim != 2 or a.shape[1] != self.arch.ctx_len:
            raise ValueError(f"token ids must be [B,{self.arch.ctx_len}], got {a.shape}")
        if a.size and (a.min() < 0 or a.max() >= self.arch.vocab):
            # the reference's nn.Embedding raises on an out-of-range id; the kernel would clamp it silently
            raise IndexError(f"token id out of range [0, {self.arch.vocab}): min {a.min()}, max {a.max()}")
        return np.ascontiguousarray(a.astype(np.int32, copy=False))

    def encode_image(self, pixels, f32=False) -> np.ndarray:
        """fp16 [B, E] unit-norm rows (mapper.py:57-59); f32=True: the fp32 rows before the fp16 rounding."""
        a, fmt = self._image_array(pixels)
        out = np.empty((a.shape[0], self.embed_dim), dtype=np.float32 if f32 else np.float16)
        fn = self._lib.clipx_encode_image_f32 if f32 else self._lib.clipx_encode_image
        check(self._lib, fn(self._h, a.ctypes.data, a.shape[0], fmt, out.ctypes.data), "clipx")
        return out

    def encode_text(self, ids, f32=False) -> np.ndarray:
        a = self._token_array(ids)
        out = np.empty((a.shape[0], self.embed_dim), dtype=np.float32 if f32 else np.float16)
        fn = self._lib.clipx_encode_text_f32 if f32 else self._lib.clipx_encode_text
        check(self._lib, fn(self._h, a.ctypes.data, a.shape[0], out.ctypes.data), "clipx")
        return out

    # ---- decoded sources of any size: resize + centre crop on the GPU (csrc/preprocess.hip), then the image tower
    def resize_crop_device(self, raw, stream=None):
        """`raw` = {"pixels": uint8 1-D torch tensor (page-locked or not) with the packed RGB sources, "offsets": int64 [B],
        "hw": int32 [B, 2]} (reader._collate) -> uint8 [B, S, S, 3] torch tensor on this encoder's GPU, bit-identical to
        Pillow's bicubic resize + centre crop (the reference's image_transform, reader.py:83,87)."""
        import torch  # pylint: disable=import-outside-toplevel

        dev = torch.device("cuda", self.device)
        B, S = int(raw["hw"].shape[0]), self.arch.image_size
        src = raw["pixels"].to(dev, non_blocking=True)
        offsets = np.ascontiguousarray(raw["offsets"], dtype=np.int64)
        hw = np.ascontiguousarray(raw["hw"], dtype=np.int32)
        out = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        check(self._lib, self._lib.clipx_resize_crop_u8_device(self.device, C.c_void_p(src.data_ptr()), offsets.ctypes.data, hw.ctypes.data,
                                                               B, S, C.c_void_p(out.data_ptr()), C.c_void_p(st) if st else None), "clipx")
        out._clipx_src = src  # the kernel reads `src` asynchronously: keep it alive as long as the result
        return out

    def encode_image_raw(self, raw) -> np.ndarray:
        """fp16 [B, E] unit-norm rows from decoded sources (see resize_crop_device): upload once, resize / crop / normalise and
        encode on the GPU.  Everything is ordered on one side stream of this encoder (a NULL stream handle would make the
        library fall back to its own stream and lose the ordering against the resize kernel)."""
        import torch  # pylint: disable=import-outside-toplevel

        dev = torch.device("cuda", self.device)
        if getattr(self, "_raw_stream", None) is None:
            self._raw_stream = torch.cuda.Stream(dev)
        s = self._raw_stream
        with torch.cuda.stream(s):
            u8 = self.resize_crop_device(raw, s.cuda_stream)
            out = torch.empty((u8.shape[0], self.embed_dim), dtype=torch.float16, device=dev)
            self.encode_image_device(u8.data_ptr(), u8.shape[0], PIX_U8_NHWC, out.data_ptr(), None, s.cuda_stream)
            host = torch.empty(out.shape, dtype=torch.float16, pin_memory=True)
            host.copy_(out, non_blocking=True)
        # synchronises the stream AND raises ResidualStreamOverflow if a row of this call left the fp16 range (include/clipx.h: "reported,
        # never hidden" -- the *_device entry points only set the flag; ADVICE r4: a plain synchronize here wrote finite-looking wrong rows)
        self.check_range(s.cuda_stream)
        return host.numpy()

    # ---- asynchronous tickets (clipx_encode_*_async / clipx_wait): submit now, collect later
    def submit_image(self, pixels):
        a, fmt = self._image_array(pixels)
        out = np.empty((a.shape[0], self.embed_dim), dtype=np.float16)
        t = C.c_void_p()
        check(self._lib, self._lib.clipx_encode_image_async(self._h, a.ctypes.data, a.shape[0], fmt, out.ctypes.data, C.byref(t)), "clipx")
        return {"ticket": t, "out": out, "keep": (a, pixels)}  # inputs stay alive until the upload is done

    def submit_text(self, ids):
        a = self._token_array(ids)
        out = np.empty((a.shape[0], self.embed_dim), dtype=np.float16)
        t = C.c_void_p()
        check(self._lib, self._lib.clipx_encode_text_async(self._h, a.ctypes.data, a.shape[0], out.ctypes.data, C.byref(t)), "clipx")
        return {"ticket": t, "out": out, "keep": (a, ids)}

    def collect(self, handle) -> np.ndarray:
        t, handle["ticket"] = handle["ticket"], None
        if t is None:
            raise HipLibraryError("ticket was already collected")
        check(self._lib, self._lib.clipx_wait(t), "clipx")
        handle["keep"] = None
        return handle["out"]

    # ---- device buffers (benchmark; callers that stage their own input)
    def encode_image_device(self, pix_ptr, B, fmt, out_f16_ptr, out_f32_ptr=None, stream=None):
        check(self._lib, self._lib.clipx_encode_image_device(
            self._h, C.c_void_p(int(pix_ptr)), int(B), int(fmt), C.c_void_p(int(out_f16_ptr)),
            C.c_void_p(int(out_f32_ptr)) if out_f32_ptr else None, C.c_void_p(int(stream)) if stream else None), "clipx")

    def encode_text_device(self, ids_ptr, B, out_f16_ptr, out_f32_ptr=None, stream=None, ids_host=None):
        """ids_host: the same ids as a C-contiguous int32 numpy array [B, ctx_len] when the caller has them on the host (the
        reference's batch does): the ragged text tower then needs no read-back and the call stays asynchronous."""
        o32 = C.c_void_p(int(out_f32_ptr)) if out_f32_ptr else None
        st = C.c_void_p(int(stream)) if stream else None
        if ids_host is not None:
            ids_host = np.ascontiguousarray(ids_host, dtype=np.int32)
            if ids_host.shape != (int(B), self.arch.ctx_len):
                raise ValueError(f"ids_host must be [{B}, {self.arch.ctx_len}]")
            check(self._lib, self._lib.clipx_encode_text_device_ids(self._h, C.c_void_p(int(ids_ptr)), ids_host.ctypes.data, int(B),
                                                                    C.c_void_p(int(out_f16_ptr)), o32, st), "clipx")
            return
        check(self._lib, self._lib.clipx_encode_text_device(self._h, C.c_void_p(int(ids_ptr)), int(B), C.c_void_p(int(out_f16_ptr)), o32, st), "clipx")

    def check_range(self, stream=None):
        """After *_device calls: synchronise `stream` and raise ResidualStreamOverflow if the fp16 residual stream overflowed in any
        of them since the last check (include/clipx.h: clipx_range_check).  The host-buffer calls and tickets check by themselves."""
        check(self._lib, self._lib.clipx_range_check(self._h, C.c_void_p(int(stream)) if stream else None), "clipx")

    OPT_RAGGED_TEXT, OPT_POOL_LAST_BLOCK = 1, 2

    def set_option(self, option, value):
        check(self._lib, self._lib.clipx_set_option(self._h, int(option), int(value)), "clipx")

    def get_option(self, option):
        return int(self._lib.clipx_get_option(self._h, int(option)))

    def graphs_cached(self):
        """Small-batch launch sequences captured as hipGraphs so far (include/clipx.h: clipx_graphs_cached)."""
        return int(self._lib.clipx_graphs_cached(self._h))

    def profile(self, on):
        """on: False/0 off, True/1 every kind, or a mask with bit (kind + 1): 2 gemm, 4 attention, 8 layernorm, 16 other."""
        check(self._lib, self._lib.clipx_profile_enable(self._h, int(on)), "clipx")

    def profile_get(self, kind):
        n, ms, fl = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        check(self._lib, self._lib.clipx_profile_get(self._h, int(kind), C.byref(n), C.byref(ms), C.byref(fl)), "clipx")
        return int(n.value), float(ms.value), float(fl.value)


class _TorchModelFacade:
    """The `model` object of `load_clip`: torch tensors in, torch float32 features out (already unit
    norm, so the caller's `/= norm` of mapper.py:58 / clip_back.py:231 is a no-op up to rounding)."""

    def __init__(self, enc: ClipEncoder):
        self._enc = enc

    def encode_image(self, x):
        import torch  # pylint: disable=import-outside-toplevel

        return torch.from_numpy(self._enc.encode_image(x.detach().cpu(), f32=True))

    def encode_text(self, tok):
        import torch  # pylint: disable=import-outside-toplevel

        return torch.from_numpy(self._enc.encode_text(tok.detach().cpu(), f32=True))


_registry = {}
_registry_lock = threading.Lock()


def register_encoder(name: str, encoder: ClipEncoder):
    """Make an already-built encoder available as clip_model=`registered:<name>` (tests inject oracle weights)."""
    with _registry_lock:
        _registry[("registered:" + name, encoder.device)] = encoder


# `clip_model` strings of the reference (all_clip: "ViT-L/14", "open_clip:ViT-H-14/laion2b_s32b_b79k", "hf_clip:openai/clip-vit-large-patch14", ...)
_HF_NAMES = {"openai/clip-vit-base-patch32": "ViT-B/32", "openai/clip-vit-base-patch16": "ViT-B/16",
             "openai/clip-vit-large-patch14": "ViT-L/14"}


def resolve_arch_name(clip_model: str) -> str:
    """Map a reference model string onto a key of ARCHS (raises ValueError for unknown architectures)."""
    if clip_model in ARCHS:
        return clip_model
    if clip_model.startswith("hf_clip:"):
        name = _HF_NAMES.get(clip_model[len("hf_clip:"):])
        if name:
            return name
    if clip_model.startswith("open_clip:"):
        arch = clip_model[len("open_clip:"):].split("/")[0]  # "ViT-H-14/laion2b_s32b_b79k" -> "ViT-H-14"
        pretrained = clip_model.split("/", 1)[1] if "/" in clip_model else ""
        if pretrained == "openai" and "/".join(arch.rsplit("-", 1)) in ARCHS:
            return "/".join(arch.rsplit("-", 1))  # open_clip:ViT-L-14/openai = the OpenAI (quick_gelu) weights
        if "open_clip:" + arch in ARCHS:
            return "open_clip:" + arch
    raise ValueError(f"unknown clip_model {clip_model!r}; known architectures: {sorted(ARCHS)} "
                     "(also hf_clip:openai/clip-vit-*, open_clip:<arch>/<pretrained>)")


def _strip_prefixes(sd):
    """Checkpoints saved from DataParallel / Lightning wrappers: drop a common 'module.' / 'model.' prefix."""
    for pre in ("module.", "model."):
        if sd and all(k.startswith(pre) for k in sd):
            sd = {k[len(pre):]: v for k, v in sd.items()}
    return sd


def get_encoder(clip_model: str, clip_cache_path=None, device: int = 0) -> ClipEncoder:
    """Resolve a `clip_model` string to a resident encoder; cached per (model, device) because the
    reference constructs a ClipMapper per output partition (runner.py:31).

      registered:<name>        an encoder passed to register_encoder()
      random:<arch>[:seed]     random-init weights (benchmarks)
      <arch>                   checkpoint `<clip_cache_path>/<arch with / -> ->.{pt,safetensors,bin,npy}`; also the
                               reference's `hf_clip:openai/clip-vit-*` and `open_clip:<arch>/<pretrained>` spellings
    """
    key = (clip_model, int(device))
    with _registry_lock:
        if key in _registry:
            return _registry[key]
    if clip_model.startswith("registered:"):
        raise KeyError(f"no encoder registered as {clip_model!r} on device {device}")
    if clip_model.startswith("random:"):
        parts = clip_model.split(":")
        seed = int(parts[-1]) if parts[-1].isdigit() else 0
        name = ":".join(parts[1:-1] if parts[-1].isdigit() else parts[1:])
        arch = ARCHS[name]
        enc = ClipEncoder(arch, random_blob(arch, seed), device)
    else:
        name = resolve_arch_name(clip_model)
        arch = ARCHS[name]
        stem = name.replace("open_clip:", "").replace("/", "-")
        cands = []
        if clip_cache_path:
            if os.path.isfile(clip_cache_path):
                cands.append(clip_cache_path)
            cands += [os.path.join(clip_cache_path, stem + ext) for ext in (".pt", ".safetensors", ".bin", ".npy")]
        path = next((c for c in cands if os.path.isfile(c)), None)
        if path is None:
            raise FileNotFoundError(
                f"no checkpoint for {clip_model!r} (looked for {cands or 'nothing: clip_cache_path is unset'}); "
                "this environment has no network, pass clip_cache_path=<local checkpoint>")
        enc = ClipEncoder(arch, blob_from_checkpoint(path, arch), device)
    with _registry_lock:
        _registry[key] = enc
    return enc


def load_clip(clip_model="ViT-B/32", use_jit=True, warmup_batch_size=1, clip_cache_path=None, device=None, bpe_path=None):  # pylint: disable=unused-argument
    """`all_clip.load_clip`-shaped factory -> (model, preprocess, tokenizer) (mapper.py:36-41, worker.py:52-57,
    clip_back.py:865-868).

    model       `.encode_image(x)` / `.encode_text(tok)`: torch tensors in, torch fp32 unit-norm features out (the service
                feeds fp32 features to the index, clip_back.py:231-232)
    preprocess  PIL image -> f32 [3, S, S]: CLIP's transform restated (reader.clip_preprocess)
    tokenizer   CLIP's BPE (tokenizer.SimpleTokenizer); needs the merges file (bpe_path / CLIP_BPE_PATH / next to the
                checkpoint).  When the file is absent the returned tokenizer raises that FileNotFoundError on its first
                call, so image-only pipelines still run and text pipelines fail loudly where the tokens are needed.
    """
    from .reader import ClipTransform  # pylint: disable=import-outside-toplevel
    from .tokenizer import MissingTokenizer, SimpleTokenizer  # pylint: disable=import-outside-toplevel

    dev = 0
    if isinstance(device, int):
        dev = device
    elif isinstance(device, str) and ":" in device:
        dev = int(device.split(":")[1])
    enc = get_encoder(clip_model, clip_cache_path, dev)
    if warmup_batch_size:
        S = enc.arch.image_size
        enc.encode_image(np.zeros((warmup_batch_size, 3, S, S), dtype=np.float32))
        ids = np.zeros((warmup_batch_size, enc.arch.ctx_len), dtype=np.int32)
        ids[:, 0], ids[:, 1] = enc.arch.vocab - 2, enc.arch.vocab - 1
        enc.encode_text(ids)
    size = enc.arch.image_size

    preprocess = ClipTransform(size)  # a picklable callable (the readers send it to their decode processes)

    try:
        tokenizer = SimpleTokenizer(bpe_path=bpe_path, clip_cache_path=clip_cache_path, context_length=enc.arch.ctx_len)
    except FileNotFoundError as e:
        tokenizer = MissingTokenizer(e)
    return _TorchModelFacade(enc), preprocess, tokenizer
