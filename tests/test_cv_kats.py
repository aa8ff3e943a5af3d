"""The OpenCV restatements (oracle/cv_resize_oracle.py, oracle/cv_warp_oracle.py) against known answers worked out by hand
from OpenCV's published fixed-point constants (tests/cv_kats.py) - the pin that does not come from the restatements
themselves.  The GPU kernels are held to the same cases in tests/test_gpu_cvops.py."""
import numpy as np

import cv_kats as K
from oracle import cv_resize_oracle as ro
from oracle import cv_warp_oracle as wo


def test_cubic_resize_known_answers():
    for case in (K.cubic_2x_impulse, K.cubic_2x_corner):
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
