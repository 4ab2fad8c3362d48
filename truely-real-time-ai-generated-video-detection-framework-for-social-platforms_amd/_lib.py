"""ctypes binding of libtruely_hip.so (include/truely_hip.h).  No CPU fallback: if the HIP
library is missing or fails to load, importing the engine raises -- the product path never
routes through the oracle or through eager PyTorch ops."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtruely_hip.so")


class TrlConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("min_face_size", C.c_int), ("thr0", C.c_float), ("thr1", C.c_float),
                ("thr2", C.c_float), ("factor", C.c_double), ("cap_level", C.c_int), ("cap_frame", C.c_int),
                ("max_faces", C.c_int), ("pnet_mode", C.c_int), ("embed_mode", C.c_int), ("embed_precision", C.c_int)]


class TrlError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libtruely_hip status {status}: {msg}")
        self.status = status


_lib = None
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

_SIGNATURES = {
    "trl_abi_version": (C.c_int, []),
    "trl_last_error": (C.c_char_p, []),
    "trl_default_config": (C.c_int, [C.POINTER(TrlConfig)]),
    "trl_create": (C.c_int, [C.POINTER(TrlConfig), C.POINTER(_vp)]),
    "trl_destroy": (C.c_int, [_vp]),
    "trl_load_weights": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "trl_mtcnn_detect": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "trl_mtcnn_detect_landmarks": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "trl_facenet_embed": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "trl_detect_embed": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trl_detect_crop": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trl_detect_embed_begin": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trl_detect_crop_begin": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trl_detect_embed_end": (C.c_int, [_vp]),
    "trl_facenet_embed_masked": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "trl_drift_score": (C.c_int, [_vp, _vp, _vp, _i, C.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "trl_ingest_nv12": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i), _vp]),
    "trl_ingest_i420": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i), _vp]),
    "trl_drift_update": (C.c_int, [_vp, _vp, _vp, _vp, _i, C.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "trl_debug_stage_boxes": (C.c_int, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "trl_debug_level_counts": (C.c_int, [_vp, _i, _vp, _vp, C.POINTER(_i)]),
    "trl_debug_level_cands": (C.c_int, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "trl_debug_level_keep": (C.c_int, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "trl_debug_batch_capacity": (C.c_int, [_vp, _f, _f, C.POINTER(_i)]),
    "trl_debug_nms_tiers": (C.c_int, [_vp, _i, _i]),
    "trl_debug_option": (C.c_int, [_vp, C.c_char_p, _i]),
    "trl_debug_list_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "trl_debug_poison": (C.c_int, [_vp, _i]),
    "trl_debug_pyramid_level": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, C.POINTER(_i), C.POINTER(_i), _vp]),
    "trl_debug_pnet_level": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, C.POINTER(_i), C.POINTER(_i), _vp]),
    "trl_debug_rnet": (C.c_int, [_vp, _vp, _i, _vp, _vp]),
    "trl_debug_onet": (C.c_int, [_vp, _vp, _i, _vp, _vp]),
    "trl_debug_front_net": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "trl_debug_crop_resize": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "trl_debug_crop_aligned": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "trl_debug_timings": (C.c_int, [_vp, C.POINTER(_f)]),
    "trl_debug_stage_totals": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "trl_debug_pnet_run": (C.c_int, [_vp, _i]),
    "trl_debug_pnet_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "trl_debug_pnet_span": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
}
EXPORTS = tuple(_SIGNATURES)


def load(path: str | None = None):
    """Load the shared library (after torch, so both share torch's HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("TRUELY_HIP_LIB") or LIB_PATH      # (the override is a tuning aid: A/B of two builds on one box)
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C {_HERE}/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; our .so then binds to the same runtime)
    except Exception:  # pragma: no cover - symbol-table checks still work without torch
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.trl_abi_version() != 7:
        raise ImportError("libtruely_hip ABI mismatch")
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise TrlError(status, load().trl_last_error().decode(errors="replace"))
