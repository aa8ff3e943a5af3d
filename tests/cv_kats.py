"""Known-answer cases for the three OpenCV 4.x uint8 operations on the crop / face paths, worked out BY HAND from the
constants of OpenCV's implementation, independent of oracle/cv_*_oracle.py (which the GPU kernels are otherwise compared
with).  cv2 itself is not installable here; these are the corner cases its fixed-point code is known for.

Sources of the constants (modules/imgproc/src/resize.cpp, imgwarp.cpp of OpenCV 4.10):
  * cv2.resize INTER_CUBIC, 8-bit: Keys kernel with A = -0.75; coefficients in 11 bits (INTER_RESIZE_COEF_BITS, scale 2048,
    saturate_cast<short>, no sum correction); source position fx = (dx + 0.5) * scale - 0.5, replicate border
    (out-of-range taps clip to the edge pixel); horizontal pass keeps 32-bit sums, vertical pass ends with
    (sum + 2^21) >> 22 and saturates to [0, 255].
    Coefficients at the dyadic fractions an exact 2x upscale produces (every one a multiple of 1/2048, so no rounding):
        f = 0.25:  -216, 1800,  536,  -72        f = 0.75:  -72,  536, 1800, -216        (f = 0: 0, 2048, 0, 0)
    e.g. f = 0.25: w0 = ((A(f+1) - 5A)(f+1) + 8A)(f+1) - 4A = -0.10546875, w1 = ((A+2)f - (A+3))f^2 + 1 = 0.87890625,
    w2 = ((A+2)(1-f) - (A+3))(1-f)^2 + 1 = 0.26171875, w3 = 1 - w0 - w1 - w2 = -0.03515625.
  * cv2.resize INTER_LINEAR, 8-bit: coefficients in 11 bits; the vertical pass is
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 on the 32-bit horizontal sums S; an exact 2x decimation in
    both axes takes the INTER_AREA path instead: (a + b + c + d + 2) >> 2.
  * cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT 0: dst(x, y) = src(M^-1 (x, y)); coordinates in 10 fractional bits
    (AB_BITS) plus a rounding term of 16, cut to 5 bits of sub-pixel position (INTER_BITS); bilinear weights in 15 bits
    (INTER_REMAP_COEF_BITS), result (sum + 2^14) >> 15.
"""
import numpy as np

C25 = (-216, 1800, 536, -72)        # cubic taps at fraction 0.25 (taps at source offsets -1, 0, +1, +2)
C75 = (-72, 536, 1800, -216)


def _rgb(a):
    return np.repeat(np.asarray(a, np.uint8)[:, :, None], 3, 2)


def cubic_2x_impulse():
    """4x4 image, one pixel = 200 at (row 1, col 2); cv2.resize to 8x8 with INTER_CUBIC.  fx = (dx + 0.5)/2 - 0.5: even dx ->
    floor = dx/2 - 1, fraction 0.75; odd dx -> floor = (dx-1)/2, fraction 0.25; taps sit at floor-1 .. floor+2, clipped to 0..3.
    Worked by hand, the coefficient that multiplies source column 2 / source row 1 at the eight destination positions:"""
    wc = [0, -72, -216, 536, 1800, 1800, 536, -216]       # dx=7: floor 3, taps at 2,3,4,5 -> tap 0 (-216) is column 2; 4 and 5 clip to column 3
    wr = [-216, 536, 1800, 1800, 536, -216, -72, 0]
    assert _axis_weights(2) == wc and _axis_weights(1) == wr
    src = np.zeros((4, 4), np.uint8)
    src[1, 2] = 200
    exp = np.array([[min(255, max(0, (r * c * 200 + (1 << 21)) >> 22)) for c in wc] for r in wr], np.uint8)
    assert exp[2, 4] == 154 and exp[2, 3] == 46 and exp[0, 4] == 0 and exp[1, 2] == 0    # 1800*1800*200 / 2^22 = 154.5 -> 154 (+2^21 then floor); negatives clamp to 0
    return _rgb(src), (8, 8), _rgb(exp)


def _axis_weights(hot, n=4):
    """Weight that multiplies source index `hot` for each of the 2n destination positions of an exact 2x cubic upscale
    (replicate border: a tap whose index clips onto `hot` adds its coefficient)."""
    out = []
    for d in range(2 * n):
        fl, co = (d // 2 - 1, C75) if d % 2 == 0 else ((d - 1) // 2, C25)
        out.append(sum(c for t, c in enumerate(co) if min(max(fl - 1 + t, 0), n - 1) == hot))
    return out


def cubic_2x_corner():
    """Impulse in the corner pixel (0, 0) = 255: every tap that clips onto index 0 adds up (replicate border), and the overshoot of
    the cubic kernel saturates at 255."""
    src = np.zeros((4, 4), np.uint8)
    src[0, 0] = 255
    w = _axis_weights(0)                                    # [-72+536+1800, -216+1800, -72+536, -216, -72, 0, 0, 0] = [2264, 1584, 464, -216, -72, 0, 0, 0]
    assert w == [2264, 1584, 464, -216, -72, 0, 0, 0]
    exp = np.array([[min(255, max(0, (a * b * 255 + (1 << 21)) >> 22)) for b in w] for a in w], np.uint8)
    assert exp[0, 0] == 255 and exp[0, 3] == 0 and exp[1, 1] == 153        # 2264*2264*255 >> 22 = 311 -> 255; negative -> 0; 1584^2*255 = 152.5.. -> 153
    return _rgb(src), (8, 8), _rgb(exp)


def linear_cases():
    """(src, dsize (w, h), expected) for cv2.resize INTER_LINEAR."""
    cases = []
    # exact 2x decimation in both axes -> INTER_AREA: (a+b+c+d+2) >> 2
    src = np.array([[1, 2, 250, 255], [3, 5, 251, 254], [0, 0, 9, 9], [0, 1, 9, 10]], np.uint8)
    cases.append((_rgb(src), (2, 2), _rgb([[(1 + 2 + 3 + 5 + 2) >> 2, (250 + 255 + 251 + 254 + 2) >> 2], [(0 + 0 + 0 + 1 + 2) >> 2, (9 + 9 + 9 + 10 + 2) >> 2]])))
    # 1x2 -> 1x4 (2x upscale of a row): fx = (dx+.5)/2 - .5 = -.25, .25, .75, 1.25 -> clamped ends copy the edge pixel; inner
    # positions mix with 11-bit coefficients (1536, 512) / (512, 1536).  One source row: both vertical rows are the same row S,
    # coefficients (2048, 0):  ((2048 * (S >> 4)) >> 16 + 0 + 2) >> 2 with S = 1536*a + 512*b.
    a, b = 10, 201
    s1, s2 = 1536 * a + 512 * b, 512 * a + 1536 * b
    v = lambda S: (((2048 * (S >> 4)) >> 16) + 2) >> 2          # noqa: E731
    cases.append((_rgb([[a, b]]), (4, 1), _rgb([[a, v(s1), v(s2), b]])))
    assert (v(s1), v(s2)) == (58, 153)                             # exact values 57.75 / 153.25
    return cases


def warp_cases():
    """(src, M, dsize (w, h), expected) for cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0)."""
    row = np.array([[10, 20, 40, 80]], np.uint8)
    half = lambda p, q: (16384 * p + 16384 * q + (1 << 14)) >> 15   # noqa: E731  sub-pixel 16/32 -> weights (0.5, 0.5) in 15 bits
    cases = []
    # shift right by 0.5: dst(x) = src(x - 0.5) = mean of src(x-1), src(x); left of the image is the constant border 0
    exp = [[half(0, 10), half(10, 20), half(20, 40), half(40, 80)]]
    assert exp == [[5, 15, 30, 60]]
    cases.append((_rgb(row), np.array([[1, 0, 0.5], [0, 1, 0]], np.float64), (4, 1), _rgb(exp)))
    # shift by 0.49: the inverse map gives x - 0.49 -> fixed point round(-0.49 * 1024) = -502, + 16, >> 5 lands on the same 16/32
    # sub-pixel position as 0.5: identical output (the 5-bit sub-pixel grid)
    cases.append((_rgb(row), np.array([[1, 0, 0.49], [0, 1, 0]], np.float64), (4, 1), _rgb(exp)))
    # integer shift by one pixel down and right on a 2x2 image: exact copy, border 0
    img = np.array([[1, 2], [3, 4]], np.uint8)
    cases.append((_rgb(img), np.array([[1, 0, 1], [0, 1, 1]], np.float64), (3, 3), _rgb([[0, 0, 0], [0, 1, 2], [0, 3, 4]])))
    # quarter-pixel: weights (24576, 8192): dst(1) = src(0.75) = 0.25*10 + 0.75*20 -> (8192*10 + 24576*20 + 16384) >> 15 = 17
    q = lambda p, r: (8192 * p + 24576 * r + (1 << 14)) >> 15     # noqa: E731
    cases.append((_rgb(row), np.array([[1, 0, 0.25], [0, 1, 0]], np.float64), (4, 1), _rgb([[q(0, 10), q(10, 20), q(20, 40), q(40, 80)]])))
    return cases
