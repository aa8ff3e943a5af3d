"""Host-side mirror of ``models/adaface.py``: ``ADAFACE`` with the reference's call surface, compute in libclearcam_hip.

    adaface = ADAFACE(weights="weights/adaface_ir50_ms1mv2.safetensors")        # reference: download in the constructor (:77)
    emb = adaface(Tensor(face_img)).numpy()                                       # (1,512) float32, clearcam.py:674,1236

`face_img` is the 112x112x3 aligned face the reference passes (uint8 or float; it applies [:,:,::-1] itself, and so does the
kernel).  `embed_batch` takes (B,112,112,3).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .helpers import Tensor, as_numpy
from .weights import load_safetensors
from .yolov9 import DTYPES


class ADAFACE:
    def __init__(self, state_dict: Optional[Dict[str, np.ndarray]] = None, weights: Optional[str] = None,
                 dtype: str = "bf16", device: int = 0):
        if state_dict is None:
            path = weights or os.path.join(os.environ.get("CLEARCAM_WEIGHTS_DIR", "weights"), "adaface_ir50_ms1mv2.safetensors")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: pass state_dict= or weights= (the reference downloads "
                                        "adaface_ir50_ms1mv2.safetensors from HuggingFace; there is no network here)")
            state_dict = load_safetensors(path)
        self.dtype, self.device = dtype, device
        L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(L.cc_face_create(C.byref(self._h), DTYPES[dtype], device))
        for name, arr in state_dict.items():
            if name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(L.cc_face_load(self._h, name.encode(), _lib.ptr(a), shp, a.ndim))
        _lib.check(L.cc_face_finalize(self._h))

    def embed_batch(self, faces) -> np.ndarray:
        """(B,112,112,3) uint8 / float32 -> (B,512) float32, unit rows."""
        f = as_numpy(faces)
        if f.dtype != np.uint8:
            f = f.astype(np.float32, copy=False)
        f = np.ascontiguousarray(f)
        if f.ndim != 4 or f.shape[1:] != (112, 112, 3):
            raise ValueError(f"faces must be (B,112,112,3), got {f.shape}")
        out = np.empty((f.shape[0], 512), np.float32)
        _lib.check(_lib.lib().cc_face_embed(self._h, _lib.ptr(f), f.shape[0], int(f.dtype == np.float32), 0, _lib.ptr(out), 0, None))
        return out

    def __call__(self, x) -> Tensor:
        f = as_numpy(x)
        if f.shape != (112, 112, 3):
            raise ValueError(f"face must be (112,112,3), got {f.shape}")
        return Tensor(self.embed_batch(f[None]))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().cc_face_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
