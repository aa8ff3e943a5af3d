"""AdaFace IR-50 (clearcam_amd/adaface.py -> csrc/face.hip) against the CPU oracle (oracle/adaface_oracle.py) on seeded
weights with the reference's parameter names: BatchNorm folding, PReLU epilogue, strided shortcuts, the permuted linear."""
import numpy as np
import pytest

from clearcam_amd.weights import synthetic_adaface_state_dict
from oracle.adaface_oracle import AdaFaceOracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def face():
    sd = synthetic_adaface_state_dict(777)
    return sd, AdaFaceOracle(sd)


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("f16", 2e-3), ("bf16", 2e-2)])
def test_adaface_matches_oracle(face, dtype, tol):
    from clearcam_amd.adaface import ADAFACE
    sd, o = face
    rng = np.random.default_rng(4)
    faces = rng.integers(0, 256, (3, 112, 112, 3), dtype=np.uint8)
    ref = np.concatenate([o(f) for f in faces])
    m = ADAFACE(state_dict=sd, dtype=dtype)
    got = m.embed_batch(faces)
    assert got.shape == (3, 512) and np.allclose(np.linalg.norm(got, axis=1), 1, atol=1e-5)
    cos = (ref * got).sum(1)
    assert (cos >= 1 - tol).all(), cos
    if dtype == "f32":
        assert np.abs(ref - got).max() < 2e-5
    one = m(faces[1]).numpy()                                             # the reference's single-image surface, batch invariance
    assert one.shape == (1, 512) and np.abs(one[0] - got[1]).max() < (1e-6 if dtype == "f32" else 5e-3)
    f32in = m.embed_batch(faces.astype(np.float32))                      # float input path (Tensor(im).cast(float32))
    assert np.abs(f32in - got).max() < (1e-6 if dtype == "f32" else 5e-3)
    with pytest.raises(ValueError):
        m(np.zeros((100, 100, 3), np.uint8))
    m.close()


def test_adaface_missing_parameter_is_an_error(face):
    from clearcam_amd._lib import CCError
    from clearcam_amd.adaface import ADAFACE
    sd, _ = face
    bad = {k: v for k, v in sd.items() if k != "body.list.7.shortcut_layer0.weight"}
    with pytest.raises(CCError):
        ADAFACE(state_dict=bad)
