"""CPU oracle for `ObjectFinder.preprocess` (models/objects.py:237-242): cv2.resize(img,(224,224),INTER_CUBIC) -> f32/255
-> (x-0.5)/0.5 -> HWC->CHW.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(clearcam_amd/) never imports it.

PARITY UNPINNED: OpenCV (opencv-python-headless==4.10.0.84, requirements.txt:3) is a third-party dependency that is
neither vendored in /root/reference nor installed in this image, and the reference keeps no resized-image fixture.
This file restates the published algorithm of OpenCV 4.10 `cv::resize` for CV_8UC3 / INTER_CUBIC — the portable
scalar path of modules/imgproc/src/resize.cpp:
  * resizeGeneric setup loop: scale = 1/(dsize/ssize) in double; fx = (float)((d+0.5)*scale-0.5); s = cvFloor(fx);
    fx -= s; interpolateCubic(fx) with A=-0.75 in float; coefficients -> short via saturate_cast<short>(c*2048)
    (INTER_RESIZE_COEF_BITS = 11, cvRound = round-half-to-even);
  * HResizeCubic<uchar,int,short>: 4 taps at s-1..s+2, indices clamped to the row (border replicate), int32 sums;
  * VResizeCubic + FixedPtCast<int,uchar,22>: 4 rows at s-1..s+2 clamped, (sum + 2^21) >> 22, saturate to uint8.
Binary wheels may dispatch the vertical pass to a float SIMD variant (VResizeCubicVec_32s8u) or to IPP; those can
differ from this path by 1 LSB on rounding ties.  The HIP kernel (csrc/crop.hip) is held bit-exact to THIS file.
"""
from __future__ import annotations

import numpy as np


def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """interpolateCubic(float x, float* coeffs) evaluated in float32 in OpenCV's operation order -> (n,4) int16."""
    f = np.float32
    x = x.astype(np.float32)
    A = f(-0.75)
    x1 = x + f(1)
    c0 = ((A * x1 - f(5) * A) * x1 + f(8) * A) * x1 - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    ix = f(1) - x
    c2 = ((A + f(2)) * ix - (A + f(3))) * ix * ix + f(1)
    c3 = f(1) - c0 - c1 - c2
    c = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32) * f(2048)
    return np.clip(np.rint(c), -32768, 32767).astype(np.int16)       # np.rint = round half to even = cvRound


def _axis(src: int, dst: int):
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    fx = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int32)
    fx = fx - s.astype(np.float32)
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, src - 1)   # (dst,4) clamped tap positions
    return idx, _cubic_coeffs(fx)


def resize_cubic_u8(img: np.ndarray, size: int = 224) -> np.ndarray:
    """cv2.resize(img, (size,size), interpolation=cv2.INTER_CUBIC) for an (H,W,C) uint8 image."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    xi, xa = _axis(W, size)
    yi, ya = _axis(H, size)
    src = img.astype(np.int64)
    # horizontal pass: (H, size, C) int sums, no rounding
    hor = (src[:, xi, :] * xa.astype(np.int64)[None, :, :, None]).sum(axis=2)
    # vertical pass
    ver = (hor[yi, :, :] * ya.astype(np.int64)[:, :, None, None]).sum(axis=1)
    out = (ver + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess(img: np.ndarray, size: int = 224) -> np.ndarray:
    """ObjectFinder.preprocess: (H,W,3) uint8 -> (3,size,size) float32."""
    r = resize_cubic_u8(img, size).astype(np.float32) / 255.0
    r = (r - 0.5) / 0.5
    return np.transpose(r.astype(np.float32), (2, 0, 1))
