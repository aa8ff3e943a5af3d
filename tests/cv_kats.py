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
from fractions import Fraction

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


def cubic_coeffs_exact(f: Fraction):
    """The Keys kernel (A = -3/4) at fraction f in EXACT rational arithmetic, times 2048, rounded half to even - independent of
    oracle/cv_resize_oracle.py, which evaluates OpenCV's float32 expression.  The two can only differ where c * 2048 lies within
    float32 round-off (~1e-4) of a .5 tie; the fractions used below are nowhere near one."""
    A = Fraction(-3, 4)
    x1, ix = f + 1, 1 - f
    c0 = ((A * x1 - 5 * A) * x1 + 8 * A) * x1 - 4 * A
    c1 = ((A + 2) * f - (A + 3)) * f * f + 1
    c2 = ((A + 2) * ix - (A + 3)) * ix * ix + 1
    c3 = 1 - c0 - c1 - c2
    out = []
    for c in (c0, c1, c2, c3):
        v = c * 2048
        fl = v.numerator // v.denominator
        r = v - fl
        out.append(fl + (1 if r > Fraction(1, 2) or (r == Fraction(1, 2) and fl % 2 == 1) else 0))
    return tuple(out)


# Hand-worked (decimal long-hand, f = 0.1): x1 = 1.1 -> ((-0.825 + 3.75) * 1.1 - 6) * 1.1 + 3 = -0.06075 -> * 2048 = -124.416 -> -124;
# (1.25 * 0.1 - 2.25) * 0.01 + 1 = 0.97875 -> 2004.48 -> 2004;  (1.25 * 0.9 - 2.25) * 0.81 + 1 = 0.08875 -> 181.76 -> 182;
# 1 + 0.06075 - 0.97875 - 0.08875 = -0.00675 -> -13.824 -> -14.   f = 0.3: -0.11025 -> -225.792 -> -226; 0.83125 -> 1702.4 -> 1702;
# 0.32625 -> 668.16 -> 668; -0.04725 -> -96.768 -> -97: the four sum to 2047 - OpenCV does not renormalise ("no sum correction").
# f = 0.5: -0.09375, 0.59375, 0.59375, -0.09375 -> -192, 1216, 1216, -192 exactly.
C10, C30, C50 = (-124, 2004, 182, -14), (-226, 1702, 668, -97), (-192, 1216, 1216, -192)
C70, C90 = C30[::-1], C10[::-1]


def cubic_4_to_5_impulse():
    """A NON-dyadic scale: 4x4 -> 5x5, scale 0.8.  Source position of destination d: (d + 0.5) * 0.8 - 0.5 = -0.1, 0.7, 1.5, 2.3, 3.1
    -> floor -1, 0, 1, 2, 3 with fractions 0.9, 0.7, 0.5, 0.3, 0.1 (none near a float32 boundary); taps at floor-1 .. floor+2 clip to
    0..3.  Impulse 250 at (row 1, column 2): out[dy][dx] = (WY[dy] * WX[dx] * 250 + 2^21) >> 22, saturated."""
    assert [cubic_coeffs_exact(Fraction(k, 10)) for k in (1, 3, 5, 7, 9)] == [C10, C30, C50, C70, C90]
    floors, coefs = [-1, 0, 1, 2, 3], [C90, C70, C50, C30, C10]

    def weights(hot):
        return [sum(c for t, c in enumerate(co) if min(max(fl - 1 + t, 0), 3) == hot) for fl, co in zip(floors, coefs)]
    wx, wy = weights(2), weights(1)
    # column 2 is tap +2 for d=0 (floor -1: taps -2,-1,0,1 -> clipped 0,0,0,1: column 2 not reached -> 0), tap 3 for d=1 (floor 0: taps -1..2),
    # tap 2 for d=2, tap 1 for d=3, tap 0 for d=4 (floor 3: taps 2,3,4,5 -> 2,3,3,3)
    assert wx == [0, C70[3], C50[2], C30[1], C10[0]] == [0, -226, 1216, 1702, -124]
    assert wy == [C90[3], C70[2], C50[1], C30[0], 0] == [-124, 1702, 1216, -226, 0]
    src = np.zeros((4, 4), np.uint8)
    src[1, 2] = 250
    exp = np.array([[min(255, max(0, (r * c * 250 + (1 << 21)) >> 22)) for c in wx] for r in wy], np.uint8)
    assert exp[1, 3] == 173 and exp[2, 2] == 88 and exp[1, 2] == 123 and exp[0, 3] == 0      # 1702*1702*250/2^22 = 172.66 -> 173; 1216^2*250/2^22 = 88.13; 1702*1216*250 = 123.36
    return _rgb(src), (5, 5), _rgb(exp)


def cubic_4_to_5_flat_rows():
    """Rows 10, 60, 200, 255 (constant along x) resized 4x4 -> 5x5: horizontally every destination sums the four coefficients of its
    fraction over ONE value - 2048 for fractions 0.9 / 0.5 / 0.1 but 2047 for 0.7 and 0.3 (no renormalisation) - vertically the taps
    mix the four rows with replicate clipping.  Expected = (SY[dy] * SX[dx] + 2^21) >> 22 with SY = sum_t cy[t] * row[clip(tap)]."""
    rows = [10, 60, 200, 255]
    floors, coefs = [-1, 0, 1, 2, 3], [C90, C70, C50, C30, C10]
    sx = [sum(co) for co in coefs]
    assert sx == [2048, 2047, 2048, 2047, 2048]
    sy = [sum(c * rows[min(max(fl - 1 + t, 0), 3)] for t, c in enumerate(co)) for fl, co in zip(floors, coefs)]
    assert sy[0] == (-14 + 182 + 2004) * 10 + (-124) * 60 == 14280                # taps -2,-1,0 clip onto row 0, tap +1 is row 1
    exp = np.array([[min(255, max(0, (a * b + (1 << 21)) >> 22)) for b in sx] for a in sy], np.uint8)
    assert exp[0, 0] == 7 and exp[4, 4] == 255 and exp[2, 0] == 130              # 14280/2048 = 6.97 -> 7;  row 2: (-192*10 + 1216*60 + 1216*200 - 192*255)/2048 = 129.5 -> 130
    return _rgb(np.repeat(np.asarray(rows, np.uint8)[:, None], 4, 1)), (5, 5), _rgb(exp)


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
    # 2x decimation in x but 4x in y: NOT the area path (OpenCV switches only when both integer scales are 2).  Bilinear with
    # fy = (0 + 0.5) * 4 - 0.5 = 1.5 reads rows 1 and 2 only (an area average would pull in the 255s of rows 0 and 3: 140);
    # fx = 0.5: columns 0 and 1 with coefficients (1024, 1024).  S_row = 1024 * (a + b); ((1024 * (S1 >> 4)) >> 16) + ((1024 * (S2 >> 4)) >> 16) + 2) >> 2.
    src = np.array([[255, 255], [10, 20], [30, 40], [255, 255]], np.uint8)
    S1, S2 = 1024 * 30, 1024 * 70
    r = (((1024 * (S1 >> 4)) >> 16) + ((1024 * (S2 >> 4)) >> 16) + 2) >> 2
    assert r == 25
    cases.append((_rgb(src), (1, 1), _rgb([[r]])))
    # 2x in x, 1x in y (3 rows): bilinear per row with (1024, 1024), the vertical pass sees coefficients (2048, 0)
    src = np.array([[10, 21, 200, 255], [0, 1, 2, 3], [7, 7, 9, 8]], np.uint8)
    hv = lambda a, b: (((2048 * ((1024 * (a + b)) >> 4)) >> 16) + 2) >> 2       # noqa: E731
    exp = [[hv(10, 21), hv(200, 255)], [hv(0, 1), hv(2, 3)], [hv(7, 7), hv(9, 8)]]
    assert exp == [[16, 228], [1, 3], [7, 9]]                                   # 15.5 -> 16, 227.5 -> 228, 0.5 -> 1 (the +2 >> 2 rounds half up), 2.5 -> 3, 8.5 -> 9
    cases.append((_rgb(src), (2, 3), _rgb(exp)))
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
    # a ROTATION by 90 degrees (off-diagonal matrix, inverted inside warpAffine): M maps (xs, ys) -> (ys, 2 - xs), so
    # dst[row yd][col xd] = src[row xd][col 2 - yd]; every coordinate is an integer, the bilinear weights are (32768, 0, 0, 0): exact copy
    img = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.uint8)
    cases.append((_rgb(img), np.array([[0, 1, 0], [-1, 0, 2]], np.float64), (3, 3), _rgb([[3, 6, 9], [2, 5, 8], [1, 4, 7]])))
    # a SHEAR: xd = xs + 0.5 ys.  Inverse xs = xd - 0.5 yd: row 0 copies, row 1 is the half-pixel blend (fixed point: (1024 x - 512 + 16) >> 5
    # = 32 x - 16 -> pixel x - 1, sub-pixel 16/32), row 2 is shifted by one whole pixel; left of the image reads the border 0
    img = np.array([[10, 20, 40]] * 3, np.uint8)
    cases.append((_rgb(img), np.array([[1, 0.5, 0], [0, 1, 0]], np.float64), (3, 3), _rgb([[10, 20, 40], [half(0, 10), half(10, 20), half(20, 40)], [0, 10, 20]])))
    return cases


def tinygrad_interpolate_u8_2x2_to_3x3():
    """tinygrad `Tensor.interpolate(size, mode="linear", align_corners=False)` on uint8 (utils/helpers.py:127-131 -> the detector's
    letterbox), worked by hand from SURVEY.md Appendix B-1, independently of oracle/yolov9_oracle.py and of tools/refshim:
      index(i) = clip(scale * (i + 0.5) - 0.5, 0, n_in - 1), scale = 2/3:  i=0: -0.1667 -> 0 (low 0, high 0, frac 0);
                 i=1: 0.6667 * 1.5 - 0.5 = 0.5 (low 0, high 1, frac 0.5);  i=2: 1.1667 -> clipped to 1 (frac 0)
      lerp(a, b, frac) for uint8: w = int(frac * 128 + 0.5) = 64;  d = int8(b - a);  a + (uint16(int16(d * w + 64)) >> 7)  (mod 256)
    W axis first, then H.  Source [[10, 250], [100, 20]]:
      row 0: lerp(10, 250): b - a = 240 -> int8 -16 (THE WRAP) -> -16 * 64 + 64 = -960 -> uint16 64576 >> 7 = 504 -> (10 + 504) mod 256 = 2   (not 130)
      row 1: lerp(100, 20): d = -80 -> -5056 -> 60480 >> 7 = 472 -> 572 mod 256 = 60
      middle output row = lerp(row 0, row 1): (10, 100): d = 90 -> 5824 >> 7 = 45 -> 55;  (2, 60): d = 58 -> 3776 >> 7 = 29 -> 31;
                                              (250, 20): b - a = -230 = 26 mod 256 -> int8 26 -> 1728 >> 7 = 13 -> 263 mod 256 = 7"""
    src = np.array([[10, 250], [100, 20]], np.uint8)
    exp = np.array([[10, 2, 250], [55, 31, 7], [100, 60, 20]], np.uint8)
    return src, (3, 3), exp
