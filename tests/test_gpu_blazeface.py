"""BlazeFace (clearcam_amd/blazeface.py -> csrc/blaze.hip) against the CPU oracle (oracle/blazeface_oracle.py) on seeded
weights with the reference's parameter names: asymmetric pads, densified depthwise convs, shortcut + ReLU epilogue,
anchor decode, stable score ordering, the overlap rule and the back-map."""
import numpy as np
import pytest

from clearcam_amd.weights import synthetic_blazeface_state_dict
from oracle.blazeface_oracle import BlazeFaceOracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blaze():
    sd = synthetic_blazeface_state_dict(555)
    return sd, BlazeFaceOracle(sd)


def _alive(d):
    return d[d[:, 16] != 0]


@pytest.mark.parametrize("shape", [(640, 640), (360, 480), (256, 256), (300, 200)])
def test_blazeface_f32_matches_oracle(blaze, shape):
    from clearcam_amd.blazeface import BlazeFace
    sd, o = blaze
    img = np.random.default_rng(shape[0] + shape[1]).integers(0, 256, (*shape, 3), dtype=np.uint8)
    ref = o(img)
    got = BlazeFace(state_dict=sd, dtype="f32")(img).numpy()
    assert got.shape == (896, 17) and got.dtype == np.float32
    ra, ga = _alive(ref), _alive(got)
    assert len(ra) > 10 and len(ga) == len(ra)                    # same rows survive the score test and the overlap rule
    scale = min(256 / shape[1], 256 / shape[0])
    assert np.abs(ra[:, :16] - ga[:, :16]).max() <= 2e-3 / scale   # source pixels; 1e-5 of the 256-pixel network frame
    assert np.abs(ra[:, 16] - ga[:, 16]).max() <= 1e-5 * 256 / scale   # the score column is x 256 / scale too (as the reference writes it)
    # suppressed rows are zero before the back-map: the same constants in the same positions
    assert np.allclose(ref[ref[:, 16] == 0], got[got[:, 16] == 0], atol=1e-6)
    assert np.array_equal(ref[:, 16] != 0, got[:, 16] != 0)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_blazeface_16bit_modes(blaze, dtype):
    from clearcam_amd.blazeface import BlazeFace
    sd, o = blaze
    img = np.random.default_rng(9).integers(0, 256, (640, 640, 3), dtype=np.uint8)
    ref, got = _alive(o(img)), _alive(BlazeFace(state_dict=sd, dtype=dtype)(img).numpy())
    assert np.isfinite(got).all() and abs(len(got) - len(ref)) <= max(3, len(ref) // 4)
    # most surviving reference boxes have a counterpart within a few pixels
    hit = sum(1 for r in ref if len(got) and np.abs(got[:, :4] - r[:4]).max(1).min() < (8 if dtype == "f16" else 30))
    assert hit >= 0.6 * len(ref)


def test_blazeface_float_input_and_errors(blaze):
    from clearcam_amd.blazeface import BlazeFace
    sd, o = blaze
    m = BlazeFace(state_dict=sd, dtype="f32")
    img = np.random.default_rng(3).integers(0, 256, (320, 320, 3), dtype=np.uint8).astype(np.float32)
    ref, got = o(img), m(img).numpy()                                  # float frames take the float resize path
    assert np.array_equal(ref[:, 16] != 0, got[:, 16] != 0) and np.abs(_alive(ref) - _alive(got)).max() < 5e-3
    with pytest.raises(ValueError):
        m(np.zeros((10, 10), np.uint8))
    m.close()
