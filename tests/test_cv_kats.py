"""The OpenCV restatements (oracle/cv_resize_oracle.py, oracle/cv_warp_oracle.py) against known answers worked out by hand
from OpenCV's published fixed-point constants (tests/cv_kats.py) - the pin that does not come from the restatements
themselves.  The GPU kernels are held to the same cases in tests/test_gpu_cvops.py."""
import numpy as np

import cv_kats as K
from oracle import cv_resize_oracle as ro
from oracle import cv_warp_oracle as wo


def test_cubic_resize_known_answers():
    for case in (K.cubic_2x_impulse, K.cubic_2x_corner, K.cubic_4_to_5_impulse, K.cubic_4_to_5_flat_rows):
        src, size, exp = case()
        assert np.array_equal(ro.resize_cubic_u8(src, size[0]), exp), case.__name__
    # dyadic fractions: the four taps sum to exactly 2048, so a flat image stays flat under an exact 2x upscale
    assert (ro.resize_cubic_u8(np.full((4, 4, 3), 77, np.uint8), 8) == 77).all()


def test_linear_resize_known_answers():
    for src, dsize, exp in K.linear_cases():
        assert np.array_equal(wo.resize_linear_u8(src, dsize), exp), dsize


def test_warp_affine_known_answers():
    for src, M, dsize, exp in K.warp_cases():
        assert np.array_equal(wo.warp_affine_u8(src, M, dsize), exp), M.tolist()


def test_tinygrad_uint8_interpolate_hand_worked_2x2_to_3x3():
    """The detector's letterbox resize (tinygrad's uint8 `interpolate` with its 7-bit fixed-point lerp and int8 wrap) against a case
    worked by hand from SURVEY.md Appendix B - a pin for the restatement in oracle/yolov9_oracle.py that is neither the
    restatement itself nor the PyTorch stand-in the reference-run fixtures were generated over."""
    from oracle import yolov9_oracle as yo
    src, (w, h), exp = K.tinygrad_interpolate_u8_2x2_to_3x3()
    got = yo.resize_bilinear(np.repeat(src[:, :, None], 3, 2), h, w)
    assert np.array_equal(got, np.repeat(exp[:, :, None], 3, 2)), got[..., 0]
