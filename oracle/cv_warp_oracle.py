"""CPU oracle for the OpenCV calls of `ObjectFinder.img_to_face` (models/objects.py:243-354): cv2.resize (INTER_LINEAR),
cv2.copyMakeBorder (BORDER_CONSTANT), cv2.getRotationMatrix2D, cv2.warpAffine (INTER_LINEAR, BORDER_CONSTANT 0), for 8-bit
3-channel images, plus the alignment arithmetic of that function.

TEST INFRASTRUCTURE ONLY (imported by tests/; nothing under clearcam_amd/ imports it).
PARITY UNPINNED: opencv-python-headless==4.10.0.84 (requirements.txt:3) is neither vendored nor installed here.  Restated from
OpenCV 4.10's portable code paths (modules/imgproc/src/resize.cpp, imgwarp.cpp):
  resize, INTER_LINEAR, 8U   fx = (float)((dx+0.5)*scale-0.5), clamped at the borders; weights (1-fx, fx) as shorts * 2048;
                             horizontal pass exact in int32; vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2;
                             an exact 2x2 decimation is switched to INTER_AREA: (a+b+c+d+2) >> 2
  getRotationMatrix2D        alpha = cos, beta = sin (degrees); [[a, b, (1-a)cx - b cy], [-b, a, b cx + (1-a) cy]] in double
  warpAffine, INTER_LINEAR   matrix inverted in double; source coordinates in 10-bit fixed point (round-half-even), +16, >> 5:
                             5-bit sub-pixel positions; weights 32*(32-fy)*(32-fx) ... (sum 2^15); (sum + 2^14) >> 15;
                             outside pixels read the constant border value 0
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import numpy as np


def _rint(x):                                                 # cvRound / saturate_cast<int>(double): round half to even
    return np.rint(x).astype(np.int64)


def resize_linear_u8(img: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (dw, dh)) with the default INTER_LINEAR, (H,W,C) uint8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (W, H):
        return img.copy()
    inv_x, inv_y = dw / W, dh / H
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if abs(scale_x - 2) < np.finfo(np.float64).eps and abs(scale_y - 2) < np.finfo(np.float64).eps and W % 2 == 0 and H % 2 == 0:
        s = img.astype(np.int32)                              # INTER_LINEAR with an exact 2x2 decimation -> INTER_AREA fast path
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def axis(n_src, n_dst, scale):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0; s[hi] = n_src - 1
        a0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
        a1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1), a0, a1

    x0, x1, ax0, ax1 = axis(W, dw, scale_x)
    y0, y1, ay0, ay1 = axis(H, dh, scale_y)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]          # (H, dw, C), values * 2048
    s0, s1 = hor[y0], hor[y1]
    out = (((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def copy_make_border(img: np.ndarray, top: int, bottom: int, left: int, right: int, value=0) -> np.ndarray:
    H, W, C = img.shape
    out = np.full((H + top + bottom, W + left + right, C), value, img.dtype)
    out[top:top + H, left:left + W] = img
    return out


def get_rotation_matrix_2d(center, angle_deg: float, scale: float) -> np.ndarray:
    a = math.cos(angle_deg * math.pi / 180.0) * scale
    b = math.sin(angle_deg * math.pi / 180.0) * scale
    cx, cy = float(center[0]), float(center[1])
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], np.float64)


def warp_affine_u8(img: np.ndarray, M: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.warpAffine(img, M, (dw, dh)) with the defaults INTER_LINEAR, BORDER_CONSTANT 0; (H,W,C) uint8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    m = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11; m[0, 1] *= -D; m[1, 0] *= -D; m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs, ys = np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64)
    adelta, bdelta = _rint(m[0, 0] * xs * 1024), _rint(m[1, 0] * xs * 1024)
    X0 = _rint((m[0, 1] * ys + m[0, 2]) * 1024) + 16
    Y0 = _rint((m[1, 1] * ys + m[1, 2]) * 1024) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5                   # 5 fractional bits
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [32 * (32 - fy) * (32 - fx), 32 * (32 - fy) * fx, 32 * fy * (32 - fx), 32 * fy * fx]
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    acc = (tap(sy, sx) * w[0][..., None] + tap(sy, sx + 1) * w[1][..., None] +
           tap(sy + 1, sx) * w[2][..., None] + tap(sy + 1, sx + 1) * w[3][..., None])
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def img_to_face(orig: np.ndarray, blazeface: Callable[[np.ndarray], np.ndarray]) -> Optional[np.ndarray]:
    """The alignment of models/objects.py:243-354: letterbox to 640, detect, take the first surviving face, rotate the 2x
    face box so the eyes are level, scale/translate the eyes to (38,51) / (73,51) in a 112x112 image, swap RGB<->BGR."""
    full = orig
    h, w = orig.shape[:2]
    scale = 640 / max(h, w)
    small = resize_linear_u8(orig, (int(w * scale), int(h * scale)))
    dw, dh = 640 - small.shape[1], 640 - small.shape[0]
    top, left = dh // 2, dw // 2
    det = np.asarray(blazeface(copy_make_border(small, top, dh - top, left, dw - left)))
    det = det[det[:, 0] != 0]
    if det.shape[0] == 0:
        return None
    y1, x1, y2, x2 = (float(v) for v in det[0][:4])
    eye_l = np.array([det[0][4], det[0][5]])
    eye_r = np.array([det[0][6], det[0][7]])
    x1, x2, y1, y2 = (x1 - left) / scale, (x2 - left) / scale, (y1 - top) / scale, (y2 - top) / scale
    eye_l = (eye_l - np.array([left, top])) / scale
    eye_r = (eye_r - np.array([left, top])) / scale
    if (x2 - x1) < 50:
        return None
    tgt_l, tgt_r = np.array([38, 51]), np.array([73, 51])
    centre = (eye_l + eye_r) / 2
    tgt_dist = np.linalg.norm(tgt_r - tgt_l)
    angle = np.degrees(np.arctan2(eye_r[1] - eye_l[1], eye_r[0] - eye_l[0]))
    size = max(x2 - x1, y2 - y1) * 2.0
    H, W = full.shape[:2]
    cx1, cy1 = max(0, int(centre[0] - size / 2)), max(0, int(centre[1] - size / 2))
    cx2, cy2 = min(W, int(centre[0] + size / 2)), min(H, int(centre[1] + size / 2))
    if cx2 <= cx1 or cy2 <= cy1:
        return None
    crop = full[cy1:cy2, cx1:cx2]
    ch, cw = crop.shape[:2]
    el, er = eye_l - np.array([cx1, cy1]), eye_r - np.array([cx1, cy1])
    R = get_rotation_matrix_2d((cw / 2, ch / 2), float(angle), 1.0)
    ca, sa = abs(R[0, 0]), abs(R[0, 1])
    nw, nh = int(ch * sa + cw * ca), int(ch * ca + cw * sa)
    R[0, 2] += nw / 2 - cw / 2
    R[1, 2] += nh / 2 - ch / 2
    rotated = warp_affine_u8(crop, R, (nw, nh))
    el_r, er_r = R[:, :2] @ el + R[:, 2], R[:, :2] @ er + R[:, 2]
    s = tgt_dist / np.linalg.norm(er_r - el_r)
    T = np.array([[s, 0, tgt_l[0] - el_r[0] * s], [0, s, tgt_l[1] - el_r[1] * s]], np.float32)
    face = warp_affine_u8(rotated, T, (112, 112))
    return np.ascontiguousarray(face[:, :, ::-1])               # cv2.COLOR_RGB2BGR
