"""The CPU oracle against outputs of the REFERENCE'S OWN model code (tests/golden/refrun_*.npz).

The fixtures come from tools/make_reference_run_golden.py: the reference's detection/yolov9.py, models/objects.py (OpenCLIP),
models/adaface.py and models/blazeface.py imported unchanged and executed over a PyTorch-CPU stand-in for tinygrad
(tools/refshim), with this repo's seeded synthetic checkpoints loaded through the reference's own strict load_state_dict.
This is what pins the oracle's layer wiring, concat orders, decode, NMS / overlap rules and box scaling to the reference's code.
Tolerances are float32 reassociation only (the oracle and the reference run call different torch kernels for some layers; the
seeded detector amplifies 1e-8 to about 0.04 px over its depth): rows must agree one for one, in order."""
import glob
import os

import numpy as np
import pytest

from clearcam_amd.arch import CLIP_L14
from clearcam_amd.weights import (synthetic_adaface_state_dict, synthetic_blazeface_state_dict, synthetic_clip_state_dict,
                                  synthetic_yolov9_state_dict)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
YOLO_FILES = sorted(glob.glob(os.path.join(GOLD, "refrun_yolo_*.npz")))


def _yolo_weights(g):
    from clearcam_amd.weights import shift_class_bias
    sd = synthetic_yolov9_state_dict(str(g["size"]), int(g["weights_seed"]))
    return shift_class_bias(sd, float(g["class_bias_shift"])) if "class_bias_shift" in g and float(g["class_bias_shift"]) else sd


def frame_of(seed, shape):
    return np.random.default_rng(int(seed)).integers(0, 256, tuple(int(s) for s in shape), dtype=np.uint8)


def test_fixture_set_is_complete():
    sizes = {str(np.load(f)["size"]) for f in YOLO_FILES}
    assert sizes == {"t", "s", "m", "c", "e"}
    for f in ("refrun_clip_l14.npz", "refrun_adaface.npz", "refrun_blazeface.npz"):
        assert os.path.exists(os.path.join(GOLD, f))


@pytest.mark.parametrize("path", YOLO_FILES, ids=[os.path.basename(f)[12:-4] for f in YOLO_FILES])
def test_yolo_oracle_equals_reference_run(path):
    from oracle.yolov9_oracle import YOLOv9Oracle
    g = np.load(path)
    size, res, ref = str(g["size"]), int(g["res"]), g["det"]
    frame = frame_of(g["seed"], g["shape"])
    if "float_frame" in g and bool(g["float_frame"]):
        frame = frame.astype(np.float32)                                    # the MOT call path: float letterbox
    got = YOLOv9Oracle(size, res, _yolo_weights(g))(frame)
    assert got.shape == ref.shape == (300, 6)
    assert (ref[:, 4] > 0).sum() >= 9                                       # the fixture exercises top-k / NMS with real rows
    assert np.array_equal(ref[:, 4] > 0, got[:, 4] > 0)                     # the same rows survive NMS, in the same slots
    assert np.array_equal(ref[:, 5], got[:, 5])                             # classes
    # source-frame pixels: float32 reassociation amplified by the seeded net, times 1/gain of the box back-map
    assert np.abs(ref[:, :4] - got[:, :4]).max() <= max(0.1, 2e-4 * max(frame.shape[:2]))     # measured 0.04 px at 640, 0.22 px at 1920
    assert np.abs(ref[:, 4] - got[:, 4]).max() <= 2e-4                      # measured <= 6e-5


def test_clip_oracle_equals_reference_run():
    from oracle.clip_oracle import OpenCLIPOracle
    g = np.load(os.path.join(GOLD, "refrun_clip_l14.npz"))
    o = OpenCLIPOracle(synthetic_clip_state_dict(CLIP_L14, int(g["weights_seed"])), CLIP_L14)
    x = np.random.default_rng(int(g["image_seed"])).standard_normal((2, 3, 224, 224)).astype(np.float32)
    assert np.abs(o.precompute_embedding(x) - g["image_emb"]).max() <= 2e-6     # measured 8e-8
    assert np.abs(o.encode_tokens(g["tokens"]) - g["text_emb"]).max() <= 2e-6   # measured 1.2e-7


def test_clip_tokens_of_the_reference_run():
    """The token rows in the fixture were produced by the reference's tokenizer inside the run; ours must give the same ids."""
    from clearcam_amd.clip_tokenizer import SimpleTokenizer, find_vocab
    from oracle.clip_oracle import pad_tokens
    try:
        find_vocab()
    except FileNotFoundError:
        pytest.skip("vocab file not on this box")
    g = np.load(os.path.join(GOLD, "refrun_clip_l14.npz"))
    tok = SimpleTokenizer()
    for q, row in zip(g["queries"], g["tokens"]):
        assert np.array_equal(pad_tokens(tok.encode(str(q)))[0], row)


def test_adaface_oracle_equals_reference_run():
    from oracle.adaface_oracle import AdaFaceOracle
    g = np.load(os.path.join(GOLD, "refrun_adaface.npz"))
    o = AdaFaceOracle(synthetic_adaface_state_dict(int(g["weights_seed"])))
    got = np.concatenate([o(frame_of(s, (112, 112, 3))) for s in g["face_seeds"]])
    assert np.abs(got - g["emb"]).max() <= 5e-6                                 # measured 2e-7


def test_blazeface_oracle_equals_reference_run():
    from oracle.blazeface_oracle import BlazeFaceOracle
    g = np.load(os.path.join(GOLD, "refrun_blazeface.npz"))
    o = BlazeFaceOracle(synthetic_blazeface_state_dict(int(g["weights_seed"])))
    for name in ("wide", "tall", "square"):
        ref = g[f"{name}_det"]
        got = o(frame_of(g[f"{name}_seed"], g[f"{name}_shape"]))
        assert (ref[:, 16] != 0).sum() >= 30
        assert np.array_equal(ref[:, 16] != 0, got[:, 16] != 0)                 # score test + overlap rule keep the same rows
        assert np.abs(ref - got).max() <= 1e-3                                  # source pixels; measured 0.0


def _search_fixture():
    import json
    g = np.load(os.path.join(GOLD, "refrun_search.npz"))
    store = {str(p): e[None] for p, e in zip(g["paths"], g["embs"])}
    return store, g["query"], json.loads(str(g["cases"])), json.loads(str(g["results"]))


def test_search_restatement_equals_reference_run():
    """The reference's ObjectFinder._load_all_embeddings + search (models/objects.py:356-422, with clearcam.py's own
    event_img_info) run on pickles in its own format; the restated loop must return the same paths in the same order."""
    from oracle.clip_oracle import search_reference
    store, q, cases, results = _search_fixture()
    assert [len(r) for r in results] == [10, 3, 10, 10, 8, 0]
    for kw, ref in zip(cases, results):
        got = search_reference(store, q, **kw)
        assert [p for p, _ in got] == [p for p, _ in ref], kw
        assert np.allclose([s for _, s in got], [s for _, s in ref], atol=1e-6)


def test_stand_in_interpolate_equals_oracle_letterbox_resize():
    """Two restatements of tinygrad's uint8 / float `interpolate` (the stand-in the reference run used and the oracle's
    numpy tables) must agree bit for bit on shapes the fixtures do not cover - a guard against either one drifting."""
    import sys
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "refshim")
    sys.path.insert(0, shim)
    try:
        for k in [k for k in sys.modules if k == "tinygrad" or k.startswith("tinygrad.")]:
            del sys.modules[k]
        from tinygrad import Tensor
        from oracle.yolov9_oracle import resize_bilinear
        rng = np.random.default_rng(77)
        for (h, w), (nh, nw) in [((37, 53), (64, 96)), ((200, 120), (75, 45)), ((1080, 1920), (360, 640)), ((90, 160), (91, 161)), ((5, 7), (5, 7))]:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            for src in (img, img.astype(np.float32)):
                got = Tensor(src).permute(2, 0, 1).interpolate(size=(nh, nw), mode="linear", align_corners=False).permute(1, 2, 0).numpy()
                assert np.array_equal(got, resize_bilinear(src, nh, nw)), (h, w, nh, nw, src.dtype)
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k == "tinygrad" or k.startswith("tinygrad.")]:
            del sys.modules[k]
