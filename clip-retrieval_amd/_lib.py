"""ctypes binding of lib/libclipx.so -- the C ABI of include/clipx.h and include/knnx.h.

This is the same stub INTEGRATION.md shows a clip-retrieval maintainer.  There is no fallback: if the
shared library cannot be loaded, or a symbol is missing, importing callers get `HipLibraryError`.
"""

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libclipx.so")
_lock = threading.Lock()
_lib = None


class HipLibraryError(RuntimeError):
    """The HIP extension is missing/unloadable or a call into it failed."""


class ResidualStreamOverflow(HipLibraryError):
    """CLIPX_E_RANGE (include/clipx.h): the fp16 residual stream overflowed for these weights / inputs; the embeddings of the
    call are invalid and were not returned."""


def library_path():
    return _LIB_PATH


class ClipxModelDesc(C.Structure):
    _fields_ = [
        ("image_size", C.c_int), ("patch_size", C.c_int), ("v_width", C.c_int), ("v_layers", C.c_int),
        ("v_heads", C.c_int), ("v_mlp", C.c_int), ("ctx_len", C.c_int), ("vocab", C.c_int),
        ("t_width", C.c_int), ("t_layers", C.c_int), ("t_heads", C.c_int), ("t_mlp", C.c_int),
        ("embed_dim", C.c_int), ("act", C.c_int), ("ln_eps", C.c_float),
        ("pix_mean", C.c_float * 3), ("pix_std", C.c_float * 3),
    ]


# name -> (restype, argtypes): every symbol include/*.h declares
_P = C.c_void_p
SIGNATURES = {
    # include/knnx.h
    "knnx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "knnx_destroy": (None, [_P]),
    "knnx_reserve": (C.c_int, [_P, C.c_int64]),
    "knnx_add_f16": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_add_f32": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_attach_device_f16": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_set_id_base": (C.c_int, [_P, C.c_int64]),
    "knnx_reset": (C.c_int, [_P]),
    "knnx_ntotal": (C.c_int64, [_P]),
    "knnx_dim": (C.c_int, [_P]),
    "knnx_search": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "knnx_search_device": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "knnx_reconstruct": (C.c_int, [_P, _P, C.c_int64, _P]),
    "knnx_range_search": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "knnx_range_search_once": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P, _P, C.c_int64]),
    "knnx_ivf_set_lists": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "knnx_ivf_set_nprobe": (C.c_int, [_P, C.c_int]),
    "knnx_ivf_nlist": (C.c_int, [_P]),
    "knnx_ivf_nprobe": (C.c_int, [_P]),
    "knnx_ivf_last_scan_tiles": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "knnx_ivf_last_scan_union_tiles": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "knnx_ivfb_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "knnx_ivfb_destroy": (None, [_P]),
    "knnx_ivfb_set_centroids": (C.c_int, [_P, _P]),
    "knnx_ivfb_get_centroids": (C.c_int, [_P, _P]),
    "knnx_ivfb_set_sample": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_ivfb_assign_sample": (C.c_int, [_P, _P]),
    "knnx_ivfb_update": (C.c_int, [_P, _P, _P]),
    "knnx_ivfb_assign": (C.c_int, [_P, _P, C.c_int64, _P]),
    "knnx_ivf_begin": (C.c_int, [_P, C.c_int, _P, _P]),
    "knnx_ivf_add_assigned": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "knnx_ivf_end": (C.c_int, [_P]),
    "knnx_ivfb_set_sample_device": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_ivfb_seed_from_sample": (C.c_int, [_P, _P, _P, C.c_int64]),
    "knnx_ivfb_lloyd": (C.c_int, [_P, _P]),
    "knnx_ivfb_assign_device": (C.c_int, [_P, _P, C.c_int64, _P]),
    "knnx_ivfb_list_sizes": (C.c_int, [_P, _P, C.c_int]),
    "knnx_ivf_add_assigned_device": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "knnx_synth_rows_device": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int64, _P]),
    "knnx_mlp_create": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "knnx_mlp_forward": (C.c_int, [_P, _P, C.c_int, _P]),
    "knnx_mlp_destroy": (C.c_int, [_P]),
    "knnx_merge_topk_device": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "knnx_shards_create": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "knnx_shards_adopt": (C.c_int, [C.c_int, _P, _P, _P, C.POINTER(_P)]),
    "knnx_shards_destroy": (None, [_P]),
    "knnx_shards_reserve": (C.c_int, [_P, C.c_int64]),
    "knnx_shards_add_f16": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_shards_add_f32": (C.c_int, [_P, _P, C.c_int64]),
    "knnx_shards_synth_fill": (C.c_int, [_P, C.c_int64, C.c_uint64]),
    "knnx_shards_ntotal": (C.c_int64, [_P]),
    "knnx_shards_count": (C.c_int, [_P]),
    "knnx_shards_exchange": (C.c_int, [_P]),
    "knnx_shards_get": (_P, [_P, C.c_int]),
    "knnx_shards_search": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "knnx_shards_reconstruct": (C.c_int, [_P, _P, C.c_int64, _P]),
    "knnx_shards_range_search": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "knnx_set_coalesce": (C.c_int, [_P, C.c_int]),
    "knnx_coalesce_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knnx_i8_served": (C.c_int64, [_P]),
    "knnx_i8_rows": (C.c_int64, [_P]),
    "knnx_i8_planes": (C.c_int, [_P]),
    "knnx_i8_dominant": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "knnx_search_dedup": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_float, _P, C.c_int, C.POINTER(C.c_int)]),
    "knnx_get_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knnx_profile_enable": (C.c_int, [_P, C.c_int]),
    "knnx_profile_get": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "knnx_synth_fill": (C.c_int, [_P, C.c_int64, C.c_uint64]),
    "knnx_last_error": (C.c_char_p, []),
    # include/clipx.h
    "clipx_blob_floats": (C.c_size_t, [C.POINTER(ClipxModelDesc)]),
    "clipx_create": (C.c_int, [C.POINTER(ClipxModelDesc), _P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "clipx_destroy": (None, [_P]),
    "clipx_encode_image": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "clipx_encode_text": (C.c_int, [_P, _P, C.c_int, _P]),
    "clipx_encode_image_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "clipx_encode_text_f32": (C.c_int, [_P, _P, C.c_int, _P]),
    "clipx_encode_image_async": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "clipx_encode_text_async": (C.c_int, [_P, _P, C.c_int, _P, C.POINTER(_P)]),
    "clipx_wait": (C.c_int, [_P]),
    "clipx_encode_image_device": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "clipx_encode_text_device": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "clipx_encode_text_device_ids": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    "clipx_resize_crop_u8_device": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "clipx_range_check": (C.c_int, [_P, _P]),
    "clipx_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "clipx_get_option": (C.c_int, [_P, C.c_int]),
    "clipx_max_batch": (C.c_int, [_P]),
    "clipx_graphs_cached": (C.c_int, [_P]),
    "clipx_embed_dim": (C.c_int, [_P]),
    "clipx_gemm_bf16_device": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "clipx_gemm_bf16_ex_device": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "clipx_gemm_f16_device": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "clipx_gemm_f16_ln_device": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P]),
    "clipx_attention_device": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "clipx_attention_dh_device": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "clipx_layernorm_device": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "clipx_rowstats_device": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_float, _P]),
    "clipx_profile_enable": (C.c_int, [_P, C.c_int]),
    "clipx_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "clipx_last_error": (C.c_char_p, []),
}


def load_library():
    """dlopen lib/libclipx.so and type every entry point.  Raises HipLibraryError, never falls back."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise HipLibraryError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
        try:
            lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise HipLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{_LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(lib, rc, which):
    """Turn a negative return code into an exception carrying the library's thread-local message."""
    if rc != 0:
        fn = lib.knnx_last_error if which == "knnx" else lib.clipx_last_error
        msg = fn()
        if which == "clipx" and rc == -6:
            raise ResidualStreamOverflow(f"clipx call failed (code {rc}): {msg.decode() if msg else '?'}")
        raise HipLibraryError(f"{which} call failed (code {rc}): {msg.decode() if msg else '?'}")
