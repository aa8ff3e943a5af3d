"""ctypes binding of libclearcam_hip.so (include/clearcam_hip.h).  Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclearcam_hip.so")

# every symbol include/clearcam_hip.h declares
SYMBOLS = [
    "cc_last_error", "cc_version", "cc_device_count",
    "cc_yolo_create", "cc_yolo_load", "cc_yolo_calibrate", "cc_yolo_calibration_info", "cc_gptq_round_f16", "cc_yolo_finalize", "cc_yolo_detect", "cc_yolo_set_in_flight", "cc_yolo_submit", "cc_yolo_wait",
    "cc_yolo_get_tensor", "cc_yolo_nonfinite",
    "cc_yolo_last_gpu_ms", "cc_yolo_profile", "cc_yolo_profile_graph", "cc_yolo_destroy", "cc_conv2d_nhwc", "cc_conv_bench", "cc_dev_set", "cc_round_weights", "cc_attn_bench",
    "cc_clip_create", "cc_clip_load", "cc_clip_finalize", "cc_clip_encode_image", "cc_clip_encode_text",
    "cc_clip_set_in_flight", "cc_clip_submit_image", "cc_clip_wait",
    "cc_clip_last_gpu_ms", "cc_clip_destroy", "cc_crop_preprocess", "cc_cv_resize_linear_u8", "cc_cv_warp_affine_u8",
    "cc_blaze_create", "cc_blaze_load", "cc_blaze_finalize", "cc_blaze_detect", "cc_blaze_destroy",
    "cc_face_create", "cc_face_load", "cc_face_finalize", "cc_face_embed", "cc_face_destroy",
    "cc_ocsort_create", "cc_ocsort_update", "cc_ocsort_update_many", "cc_ocsort_num_tracks", "cc_ocsort_destroy",
    "cc_index_create", "cc_index_create_ex", "cc_index_add", "cc_index_add_grouped", "cc_index_size", "cc_index_info", "cc_index_scores",
    "cc_index_search", "cc_index_search_groups", "cc_index_destroy",
]


class ClipConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("image_size", "patch", "v_width", "v_layers", "v_heads", "v_mlp",
                                       "t_ctx", "t_vocab", "t_width", "t_layers", "t_heads", "t_mlp", "embed")]


class CCError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library (no CPU fallback: a missing .so is an error, not a slow path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CCError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        # PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64.  Whichever copy is mapped first serves the whole
        # process (same SONAME), and torch fails to find the GPU on the system copy: let torch map its runtime first.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.cc_last_error.restype = C.c_char_p
    vp, ip, fp, i64p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int64)
    sig = {
        "cc_version": [],
        "cc_device_count": [ip],
        "cc_yolo_create": [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int],
        "cc_yolo_load": [vp, C.c_char_p, vp, i64p, C.c_int],
        "cc_yolo_finalize": [vp],
        "cc_yolo_calibrate": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "cc_yolo_calibration_info": [vp, ip, ip],
        "cc_gptq_round_f16": [vp, C.c_int64, C.c_int64, vp, C.c_double, vp],
        "cc_yolo_detect": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp],
        "cc_yolo_set_in_flight": [vp, C.c_int],
        "cc_yolo_submit": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.POINTER(C.c_longlong)],
        "cc_yolo_wait": [vp, C.c_longlong, vp],
        "cc_yolo_get_tensor": [vp, C.c_char_p, vp, i64p, ip],
        "cc_yolo_nonfinite": [vp, ip],
        "cc_yolo_last_gpu_ms": [vp, fp],
        "cc_yolo_profile": [vp, C.c_int, fp, C.POINTER(C.c_double), ip],
        "cc_yolo_profile_graph": [vp, C.c_int, C.c_int, fp],
        "cc_conv2d_nhwc": [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_int, vp, C.c_int, vp],
        "cc_conv_bench": [C.c_int] * 11 + [fp],
        "cc_dev_set": [C.c_char_p, C.c_int],
        "cc_round_weights": [C.c_int, vp, C.c_int64, C.c_int64, C.c_int64, vp],
        "cc_attn_bench": [C.c_int] * 7 + [fp],
        "cc_clip_create": [C.POINTER(vp), C.POINTER(ClipConfig), C.c_int, C.c_int],
        "cc_clip_load": [vp, C.c_char_p, vp, i64p, C.c_int],
        "cc_clip_finalize": [vp],
        "cc_clip_encode_image": [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp],
        "cc_clip_encode_text": [vp, vp, C.c_int, vp, C.c_int, vp],
        "cc_clip_set_in_flight": [vp, C.c_int],
        "cc_clip_submit_image": [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.POINTER(C.c_longlong)],
        "cc_clip_wait": [vp, C.c_longlong, vp],
        "cc_clip_last_gpu_ms": [vp, fp],
        "cc_crop_preprocess": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp],
        "cc_cv_resize_linear_u8": [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int],
        "cc_cv_warp_affine_u8": [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int],
        "cc_blaze_create": [C.POINTER(vp), C.c_int, C.c_int],
        "cc_blaze_load": [vp, C.c_char_p, vp, i64p, C.c_int],
        "cc_blaze_finalize": [vp],
        "cc_blaze_detect": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp],
        "cc_face_create": [C.POINTER(vp), C.c_int, C.c_int],
        "cc_face_load": [vp, C.c_char_p, vp, i64p, C.c_int],
        "cc_face_finalize": [vp],
        "cc_face_embed": [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp],
        "cc_ocsort_create": [C.POINTER(vp), C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int],
        "cc_ocsort_update": [vp, vp, C.c_int, C.c_double, vp, C.c_int, ip],
        "cc_ocsort_update_many": [vp, C.c_int, vp, C.c_int, C.c_double, vp, C.c_int, vp, C.c_int],
        "cc_ocsort_num_tracks": [vp, ip],
        "cc_index_create": [C.POINTER(vp), C.c_int, C.c_int64, C.c_int],
        "cc_index_create_ex": [C.POINTER(vp), C.c_int, C.c_int64, C.c_int, C.c_int],
        "cc_index_add": [vp, vp, C.c_int64, C.c_int],
        "cc_index_add_grouped": [vp, vp, C.c_int64, C.c_int, vp],
        "cc_index_info": [vp, i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "cc_index_search_groups": [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp],
        "cc_index_size": [vp, i64p],
        "cc_index_scores": [vp, vp, C.c_int, vp, C.c_int, vp],
        "cc_index_search": [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("cc_yolo_destroy", "cc_clip_destroy", "cc_index_destroy", "cc_ocsort_destroy", "cc_face_destroy", "cc_blaze_destroy"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = None
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise CCError(f"libclearcam_hip error {rc}: {lib().cc_last_error().decode(errors='replace')}")


def ptr(x):
    """Raw address of a numpy array / torch tensor / int."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)
