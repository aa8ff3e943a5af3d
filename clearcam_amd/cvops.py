"""The OpenCV image calls of the face path on the GPU: `resize_linear` = cv2.resize(img, (w, h)) (default INTER_LINEAR) and
`warp_affine` = cv2.warpAffine(img, M, (w, h)) (defaults INTER_LINEAR / BORDER_CONSTANT 0) for uint8 HWC 3-channel images,
plus the two trivial host helpers `copy_make_border` and `rotation_matrix_2d` (cv2.getRotationMatrix2D).  They exist so that
`ObjectFinder.img_to_face` (models/objects.py:243-354) runs without OpenCV; libclearcam_hip does the pixel work."""
from __future__ import annotations

import math

import numpy as np

from . import _lib


def _u8(img) -> np.ndarray:
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.size == 0:
        raise ValueError(f"expected a non-empty (H,W,3) uint8 image, got {a.dtype} {a.shape}")
    return a


def resize_linear(img, dsize, device: int = 0) -> np.ndarray:
    a = _u8(img)
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw <= 0 or dh <= 0:
        raise ValueError("empty destination size")
    out = np.empty((dh, dw, 3), np.uint8)
    _lib.check(_lib.lib().cc_cv_resize_linear_u8(_lib.ptr(a), a.shape[0], a.shape[1], _lib.ptr(out), dh, dw, device))
    return out


def warp_affine(img, M, dsize, device: int = 0) -> np.ndarray:
    a = _u8(img)
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw <= 0 or dh <= 0:
        raise ValueError("empty destination size")
    m = np.ascontiguousarray(np.asarray(M, np.float64).reshape(2, 3))
    out = np.empty((dh, dw, 3), np.uint8)
    _lib.check(_lib.lib().cc_cv_warp_affine_u8(_lib.ptr(a), a.shape[0], a.shape[1], _lib.ptr(m), _lib.ptr(out), dh, dw, device))
    return out


def copy_make_border(img, top: int, bottom: int, left: int, right: int, value: int = 0) -> np.ndarray:
    a = _u8(img)
    out = np.full((a.shape[0] + top + bottom, a.shape[1] + left + right, 3), value, np.uint8)
    out[top:top + a.shape[0], left:left + a.shape[1]] = a
    return out


def rotation_matrix_2d(center, angle_deg: float, scale: float = 1.0) -> np.ndarray:
    a = math.cos(angle_deg * math.pi / 180.0) * scale
    b = math.sin(angle_deg * math.pi / 180.0) * scale
    cx, cy = float(center[0]), float(center[1])
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], np.float64)
