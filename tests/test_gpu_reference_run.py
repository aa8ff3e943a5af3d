"""The HIP path (f32 parity mode, through the C ABI) against outputs of the REFERENCE'S OWN model code.

Fixtures: tests/golden/refrun_*.npz, written by tools/make_reference_run_golden.py (the reference's detection/yolov9.py,
models/objects.py OpenCLIP, models/adaface.py and models/blazeface.py executed unchanged over a PyTorch-CPU stand-in for
tinygrad, seeded synthetic checkpoints).  Unlike the other GPU tests this file does not call the oracle at all: it is the
direct HIP-vs-reference comparison, on the inputs and weights the reference run used."""
import glob
import os

import numpy as np
import pytest

from clearcam_amd.arch import CLIP_L14
from clearcam_amd.weights import (synthetic_adaface_state_dict, synthetic_blazeface_state_dict, synthetic_clip_state_dict,
                                  synthetic_yolov9_state_dict)
from oracle.yolov9_oracle import match_detections          # the matching metric only

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
YOLO_FILES = sorted(glob.glob(os.path.join(GOLD, "refrun_yolo_*.npz")))


def _yolo_weights(g):
    from clearcam_amd.weights import shift_class_bias
    sd = synthetic_yolov9_state_dict(str(g["size"]), int(g["weights_seed"]))
    return shift_class_bias(sd, float(g["class_bias_shift"])) if "class_bias_shift" in g and float(g["class_bias_shift"]) else sd


def frame_of(seed, shape):
    return np.random.default_rng(int(seed)).integers(0, 256, tuple(int(s) for s in shape), dtype=np.uint8)


@pytest.mark.parametrize("path", YOLO_FILES, ids=[os.path.basename(f)[12:-4] for f in YOLO_FILES])
def test_yolo_hip_equals_reference_run(path):
    """Boxes within 1e-3 of the image size, scores within 1e-3, every reference detection matched one to one (class, IoU >= 0.9);
    a detection within float32 noise of the 0.25 / 0.45 thresholds may flip, hence 99 % and not 100 %."""
    from clearcam_amd.helpers import Tensor, jit_infer
    from clearcam_amd.yolov9 import YOLOv9
    g = np.load(path)
    size, res, ref = str(g["size"]), int(g["res"]), g["det"]
    frame = frame_of(g["seed"], g["shape"])
    if "float_frame" in g and bool(g["float_frame"]):
        frame = frame.astype(np.float32)                                 # test/run_mot.py:33-34: Tensor(frame).cast(float32)
    m = YOLOv9(size, res, state_dict=_yolo_weights(g), dtype="f32", device=0)
    got = jit_infer(m, Tensor(frame), {}).numpy()                        # the reference's call sequence (clearcam.py:583)
    assert got.shape == (300, 6) and got.dtype == np.float32
    n_ref, n_got, n_match, box_err, sc_err = match_detections(ref, got, 0.9)
    assert n_ref >= 9 and n_match >= 0.99 * max(n_ref, n_got) - 1, (n_ref, n_got, n_match)
    assert box_err <= 1e-3 * max(frame.shape[:2]) and sc_err <= 1e-3, (box_err, sc_err)
    # rows sit in score order; two detections whose scores differ by float32 noise may swap slots, nothing more
    assert (ref[:, 5] != got[:, 5]).mean() <= 0.05
    m.close()


def test_clip_hip_equals_reference_run():
    from clearcam_amd.objects import OpenCLIP
    g = np.load(os.path.join(GOLD, "refrun_clip_l14.npz"))
    m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_L14, int(g["weights_seed"])), arch=CLIP_L14, dtype="f32")
    x = np.random.default_rng(int(g["image_seed"])).standard_normal((2, 3, 224, 224)).astype(np.float32)
    img = m.precompute_embedding(x).numpy()
    txt = m.encode_tokens(g["tokens"])
    assert np.abs(img - g["image_emb"]).max() <= 2e-5 and ((img * g["image_emb"]).sum(1) >= 1 - 1e-6).all()
    assert np.abs(txt - g["text_emb"]).max() <= 2e-5 and ((txt * g["text_emb"]).sum(1) >= 1 - 1e-6).all()
    # the quantity the reference's own test pins (test/test_clip.py:9-12): the text-image cosine
    assert np.abs(txt @ img.T - g["text_emb"] @ g["image_emb"].T).max() <= 1e-5


def test_adaface_hip_equals_reference_run():
    from clearcam_amd.adaface import ADAFACE
    g = np.load(os.path.join(GOLD, "refrun_adaface.npz"))
    m = ADAFACE(state_dict=synthetic_adaface_state_dict(int(g["weights_seed"])), dtype="f32")
    got = np.concatenate([m(frame_of(s, (112, 112, 3))).numpy() for s in g["face_seeds"]])
    assert got.shape == g["emb"].shape and np.abs(got - g["emb"]).max() <= 2e-5
    m.close()


def test_blazeface_hip_equals_reference_run():
    from clearcam_amd.blazeface import BlazeFace
    g = np.load(os.path.join(GOLD, "refrun_blazeface.npz"))
    m = BlazeFace(state_dict=synthetic_blazeface_state_dict(int(g["weights_seed"])), dtype="f32")
    for name in ("wide", "tall", "square"):
        ref, shape = g[f"{name}_det"], g[f"{name}_shape"]
        got = m(frame_of(g[f"{name}_seed"], shape)).numpy()
        scale = min(256 / shape[1], 256 / shape[0])
        assert np.array_equal(ref[:, 16] != 0, got[:, 16] != 0)           # score test + overlap rule keep the same rows
        alive = ref[:, 16] != 0
        assert np.abs(ref[alive, :16] - got[alive, :16]).max() <= 2e-3 / scale
        assert np.abs(ref[alive, 16] - got[alive, 16]).max() <= 1e-5 * 256 / scale
        assert np.allclose(ref[~alive], got[~alive], atol=1e-6)
    m.close()


def test_search_hip_equals_reference_run():
    """ObjectFinder.search over the device index == the reference's ObjectFinder.search on the same store (paths, order, scores)."""
    import json
    from clearcam_amd.objects import ObjectFinder
    g = np.load(os.path.join(GOLD, "refrun_search.npz"))
    f = ObjectFinder()
    f.image_embeddings = {str(p): e[None] for p, e in zip(g["paths"], g["embs"])}
    for kw, ref in zip(json.loads(str(g["cases"])), json.loads(str(g["results"]))):
        got = f.search(text_embedding=g["query"], **kw)
        assert [p for p, _ in got] == [p for p, _ in ref], kw
        assert np.allclose([s for _, s in got], [s for _, s in ref], atol=1e-6)
