"""GPU image ops of the face path (clearcam_amd/cvops.py, csrc/crop.hip) bit for bit against the OpenCV restatement
(oracle/cv_warp_oracle.py), and `ObjectFinder.img_to_face` end to end against the oracle's version of the same steps."""
import numpy as np
import pytest

from oracle import cv_warp_oracle as cvo

pytestmark = pytest.mark.gpu


def test_resize_linear_bit_exact():
    from clearcam_amd import cvops
    rng = np.random.default_rng(0)
    for (h, w), (dw, dh) in [((90, 140), (300, 200)), ((90, 140), (70, 45)), ((480, 640), (640, 480)), ((7, 5), (64, 64)),
                             ((301, 203), (100, 77)), ((1080, 810), (640, 480)), ((64, 64), (64, 64)), ((50, 60), (30, 25)), ((3, 2), (1, 1))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(cvops.resize_linear(img, (dw, dh)), cvo.resize_linear_u8(img, (dw, dh))), ((h, w), (dw, dh))
    with pytest.raises(ValueError):
        cvops.resize_linear(np.zeros((4, 4), np.uint8), (2, 2))


def test_warp_affine_bit_exact():
    from clearcam_amd import cvops
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    mats = [np.array([[1, 0, 0], [0, 1, 0]], float), np.array([[1, 0, 5.5], [0, 1, -3.25]], float),
            cvo.get_rotation_matrix_2d((65.5, 48.5), 30.0, 1.0), cvo.get_rotation_matrix_2d((10, 90), -123.4, 0.7),
            np.array([[2.3, 0, -20], [0, 2.3, -31]], np.float32), np.array([[0.31, 0.1, 4], [-0.2, 0.4, 9]], float),
            np.array([[1, 0, 1e6], [0, 1, 0]], float)]                 # everything outside: all border
    for M in mats:
        for size in [(131, 97), (112, 112), (40, 200)]:
            assert np.array_equal(cvops.warp_affine(img, M, size), cvo.warp_affine_u8(img, M, size)), (M, size)
    assert np.array_equal(cvops.warp_affine(img, mats[0], (131, 97)), img)
    assert np.array_equal(cvops.rotation_matrix_2d((3, 4), 33.0, 1.5), cvo.get_rotation_matrix_2d((3, 4), 33.0, 1.5))
    assert np.array_equal(cvops.copy_make_border(img, 1, 2, 3, 4), cvo.copy_make_border(img, 1, 2, 3, 4))


def test_img_to_face_matches_oracle_steps():
    from clearcam_amd.objects import ObjectFinder
    from clearcam_amd.weights import synthetic_adaface_state_dict, synthetic_blazeface_state_dict
    f = ObjectFinder()
    f.init_face(blazeface_kwargs=dict(state_dict=synthetic_blazeface_state_dict(555), dtype="f32"),
                adaface_kwargs=dict(state_dict=synthetic_adaface_state_dict(777), dtype="f32"))
    rng = np.random.default_rng(2)
    done = 0
    for shape in [(720, 960, 3), (1280, 900, 3), (500, 500, 3)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        got = f.img_to_face(img)
        ref = cvo.img_to_face(img, lambda x: f.blazeface(x).numpy())     # same detector, oracle pixel ops and arithmetic
        assert (got is None) == (ref is None)
        if got is not None:
            assert got.shape == (112, 112, 3) and np.array_equal(got, ref)
            emb = f.adaface(got).numpy()
            assert emb.shape == (1, 512) and abs(np.linalg.norm(emb) - 1) < 1e-5
            assert np.array_equal(f.preprocess_face(got), got)
            done += 1
    assert done >= 1
    f.turn_off_face()
    assert f.blazeface is None and f.adaface is None


def test_face_path_against_committed_golden():
    """The HIP face path against tests/golden/face_path.npz (oracle outputs committed by tools/make_golden.py)."""
    import os
    from clearcam_amd import cvops
    from clearcam_amd.adaface import ADAFACE
    from clearcam_amd.blazeface import BlazeFace
    from clearcam_amd.objects import preprocess_crops
    from clearcam_amd.weights import synthetic_adaface_state_dict, synthetic_blazeface_state_dict
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "face_path.npz"))
    assert np.abs(ADAFACE(state_dict=synthetic_adaface_state_dict(777), dtype="f32")(g["face"]).numpy() - g["adaface"]).max() < 3e-5
    b = BlazeFace(state_dict=synthetic_blazeface_state_dict(555), dtype="f32")(g["img"]).numpy()
    assert np.array_equal(b[:, 16] != 0, g["blazeface"][:, 16] != 0) and np.abs(b - g["blazeface"]).max() < 2e-2
    assert np.array_equal(cvops.resize_linear(g["crop"], (200, 90)), g["crop_linear_200x90"])
    assert np.array_equal(cvops.warp_affine(g["crop"], g["warp_M"], (150, 80)), g["crop_warp_150x80"])
    cubic = preprocess_crops([g["crop"]]).cpu().numpy()[0]
    ref = np.transpose((g["crop_cubic_224"].astype(np.float32) / 255.0 - 0.5) / 0.5, (2, 0, 1))
    assert np.array_equal(cubic, ref.astype(np.float32))


def test_kernels_reproduce_the_hand_computed_opencv_cases():
    """The GPU kernels against tests/cv_kats.py: expected bytes derived by hand from OpenCV's fixed-point constants
    (11-bit cubic / linear coefficients, (sum + 2^21) >> 22, replicate border, the 2x-decimation area rule, warpAffine's
    5-bit sub-pixel grid and 15-bit weights) - independent of oracle/cv_*_oracle.py."""
    import cv_kats as K
    from clearcam_amd import cvops
    from clearcam_amd.objects import preprocess_crops
    for case in (K.cubic_2x_impulse, K.cubic_2x_corner, K.cubic_4_to_5_impulse, K.cubic_4_to_5_flat_rows):
        src, size, exp = case()
        got = preprocess_crops([src], size[0]).cpu().numpy()[0]                     # (3,8,8) float32 = (v/255 - 0.5)/0.5
        want = ((exp.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
        assert np.array_equal(got, want), case.__name__
    for src, dsize, exp in K.linear_cases():
        assert np.array_equal(cvops.resize_linear(src, dsize), exp), dsize
    for src, M, dsize, exp in K.warp_cases():
        assert np.array_equal(cvops.warp_affine(src, M, dsize), exp), M.tolist()
