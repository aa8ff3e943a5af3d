"""Host-side mirror of ``models/blazeface.py``: ``BlazeFace`` with the reference's call surface, compute in libclearcam_hip.

    blazeface = BlazeFace(weights="models/blazeface.safetensors")               # the reference loads this file (:135)
    detections = blazeface(Tensor(img)).numpy()                                   # (896,17), models/objects.py:254-255
    detections = detections[detections[:, 0] != 0]                                # the caller's own filter

Rows are [ymin, xmin, ymax, xmax, 6 x (kx, ky), score] in source pixels, score-descending, rows that fail the 0.85 score
test or the overlap rule are zeroed before the back-map - exactly what the reference returns.  There is no CPU path.
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


class BlazeFace:
    def __init__(self, state_dict: Optional[Dict[str, np.ndarray]] = None, weights: Optional[str] = None,
                 dtype: str = "bf16", device: int = 0):
        if state_dict is None:
            path = weights or os.path.join(os.environ.get("CLEARCAM_WEIGHTS_DIR", "weights"), "blazeface.safetensors")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: pass state_dict= or weights= (models/blazeface.safetensors of the reference)")
            state_dict = load_safetensors(path)
        self.dtype, self.device = dtype, device
        L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(L.cc_blaze_create(C.byref(self._h), DTYPES[dtype], device))
        for name, arr in state_dict.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(L.cc_blaze_load(self._h, name.encode(), _lib.ptr(a), shp, a.ndim))
        _lib.check(L.cc_blaze_finalize(self._h))

    def __call__(self, img) -> Tensor:
        f = as_numpy(img)
        if f.ndim != 3 or f.shape[2] != 3:
            raise ValueError(f"image must be (H,W,3), got {f.shape}")
        if f.dtype != np.uint8:
            f = f.astype(np.float32, copy=False)
        f = np.ascontiguousarray(f)
        out = np.empty((896, 17), np.float32)
        _lib.check(_lib.lib().cc_blaze_detect(self._h, _lib.ptr(f), f.shape[0], f.shape[1], int(f.dtype == np.float32), 0, _lib.ptr(out), 0, None))
        return Tensor(out)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().cc_blaze_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
