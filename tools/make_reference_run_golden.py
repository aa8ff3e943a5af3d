"""Run the REFERENCE'S OWN model code here and store what it returns: tests/golden/refrun_*.npz.

The reference's detection/yolov9.py, models/objects.py (OpenCLIP), models/adaface.py and models/blazeface.py are imported
unchanged from /root/reference.  Their only missing dependency, tinygrad, is replaced by `tools/refshim/tinygrad` (a
PyTorch-CPU implementation of the Tensor / nn calls those files make; see its docstring for what that does and does not
pin) and `import cv2` by a placeholder that nothing here touches.  Weights are this repo's seeded synthetic checkpoints,
loaded through the reference's own constructors (`load_state_dict(self, safe_load(fetch(...)))`, strict), which also checks
that their key names and shapes are exactly the reference's module tree.

Inputs are regenerated from the seeds stored in each file; tests/test_reference_run.py compares the CPU oracle with these
outputs, tests/test_gpu_reference_run.py the HIP path.  Needs /root/reference, so it runs in the build container only:

    python tools/make_reference_run_golden.py [yolo] [clip] [adaface] [blazeface] [search]
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CLEARCAM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path[:0] = [os.path.join(ROOT, "tools", "refshim"), REF, ROOT]
os.chdir(REF)                                                       # the reference opens a few files relative to its root

# models/objects.py does `from clearcam import event_img_info`.  Importing the NVR server module itself is out of the question
# (ffmpeg, sockets, argparse at import time), so that one function is lifted out of clearcam.py's syntax tree and executed
# as it stands; nothing of it is stored in this repo.
import ast  # noqa: E402
_cl = types.ModuleType("clearcam")
_tree = ast.parse(open(os.path.join(REF, "clearcam.py")).read())
_fn = [n for n in _tree.body if isinstance(n, ast.FunctionDef) and n.name == "event_img_info"]
assert len(_fn) == 1
exec(compile(ast.Module(body=_fn, type_ignores=[]), os.path.join(REF, "clearcam.py"), "exec"), _cl.__dict__)
sys.modules["clearcam"] = _cl

from tinygrad import Tensor  # noqa: E402  (the stand-in)
from tinygrad.nn import state  # noqa: E402
from clearcam_amd.arch import CLIP_L14  # noqa: E402
from clearcam_amd.weights import (shift_class_bias, synthetic_adaface_state_dict, synthetic_blazeface_state_dict,  # noqa: E402
                                  synthetic_clip_state_dict, synthetic_yolov9_state_dict)

torch.set_grad_enabled(False)

# (name, size, res, frame seed, frame shape): every detector size; square, letterboxed-wide and letterboxed-tall frames
# a trailing "f32" = the MOT call path, `model(Tensor(frame).cast(float32))` (test/run_mot.py:33-34): float letterbox
YOLO_CASES = [("t_320", "t", 320, 7, (320, 320, 3)), ("t_640_from_540x960", "t", 640, 8, (540, 960, 3)),
              ("t_640_from_540x960_f32", "t", 640, 13, (540, 960, 3)),
              ("t_960_from_1080x1920", "t", 960, 14, (1080, 1920, 3)),        # the application's default: YOLOv9("t", 960) on a 1080p camera
              ("s_320_from_400x300", "s", 320, 9, (400, 300, 3)), ("m_320", "m", 320, 10, (320, 320, 3)),
              ("c_640", "c", 640, 11, (640, 640, 3)), ("e_640_from_360x640", "e", 640, 12, (360, 640, 3))]
CLIP_QUERIES = ["a white van parked on the street", "person walking a dog at night"]


def frame_of(seed, shape):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def run_yolo():
    from detection.yolov9 import YOLOv9
    from tinygrad.dtype import dtypes
    from utils.helpers import jit_infer
    for name, size, res, seed, shape in YOLO_CASES:
        # float-interpolated noise is too smooth for the seeded head to fire (the uint8 path's detections ride on the int8 wrap
        # of tinygrad's fixed-point lerp), so the float case raises every class bias by 3
        shift = 3.0 if name.endswith("_f32") else 0.0
        sd = dict(shift_class_bias(synthetic_yolov9_state_dict(size, 1234), shift))
        head = 42 if size == "e" else 22
        # buffers a real checkpoint carries and the constructor's strict load asks for; __call__ recomputes both (:209)
        sd[f"model.list.{head}.anchors"] = np.zeros((2, 22680), np.float32)
        sd[f"model.list.{head}.strides"] = np.zeros((1, 22680), np.float32)
        state.inject(f"yolov9-{size}.safetensors", sd)
        t = time.time()
        model = YOLOv9(size, res)
        assert state.LOADED[-1][1] == [], f"checkpoint keys the reference model does not have: {state.LOADED[-1][1][:5]}"
        frame = frame_of(seed, shape)
        if name.endswith("_f32"):
            det = model(Tensor(frame).cast(dtypes.float32)).numpy()
        else:
            det = jit_infer(model, Tensor(frame), {}).numpy()           # the production call sequence (clearcam.py:582-583)
        np.savez_compressed(os.path.join(OUT, f"refrun_yolo_{name}.npz"), size=size, res=res, seed=seed, shape=np.array(shape),
                            weights_seed=1234, float_frame=name.endswith("_f32"), class_bias_shift=shift, det=det.astype(np.float32))
        print(f"yolo {name}: {(det[:, 4] > 0).sum()} detections, {time.time() - t:.1f}s")


def run_clip():
    sd = dict(synthetic_clip_state_dict(CLIP_L14, 4321))
    sd["attn_mask"] = np.triu(np.full((77, 77), -np.inf, np.float32), 1)     # open_clip's buffer, part of the real checkpoint
    state.inject("CLIP-ViT-L-14-laion2B-s32B-b82K.safetensors", sd)
    from models.objects import OpenCLIP
    model = OpenCLIP()
    assert state.LOADED[-1][1] == []
    x = np.random.default_rng(5).standard_normal((2, 3, 224, 224)).astype(np.float32)
    img = model.precompute_embedding(Tensor(x)).numpy()
    tokens, text = [], []
    for q in CLIP_QUERIES:
        ids = [49406] + model.tokenizer.encode(q) + [49407]
        tokens.append(ids + [0] * (77 - len(ids)))
        text.append(model._encode_text(q, realize=True))
    np.savez_compressed(os.path.join(OUT, "refrun_clip_l14.npz"), weights_seed=4321, image_seed=5, image_emb=img.astype(np.float32),
                        queries=np.array(CLIP_QUERIES), tokens=np.array(tokens, np.int32), text_emb=np.stack(text).astype(np.float32))
    print("clip: image", img.shape, "text", np.stack(text).shape, "cos(img, text)", float(img[0] @ text[0]))


def run_adaface():
    state.inject("adaface_ir50_ms1mv2.safetensors", synthetic_adaface_state_dict(777))
    from models.adaface import ADAFACE
    model = ADAFACE()
    assert state.LOADED[-1][1] == []
    faces = np.stack([frame_of(21, (112, 112, 3)), frame_of(22, (112, 112, 3))])
    emb = np.concatenate([model(Tensor(f)).numpy() for f in faces])
    np.savez_compressed(os.path.join(OUT, "refrun_adaface.npz"), weights_seed=777, face_seeds=np.array([21, 22]), emb=emb.astype(np.float32))
    print("adaface:", emb.shape, "norms", np.linalg.norm(emb, axis=1))


def run_blazeface():
    state.inject("blazeface.safetensors", synthetic_blazeface_state_dict(555))
    from models.blazeface import BlazeFace
    model = BlazeFace()
    assert state.LOADED[-1][1] == []
    out = {}
    for name, seed, shape in (("wide", 21, (360, 480, 3)), ("tall", 23, (500, 300, 3)), ("square", 24, (256, 256, 3))):
        det = model(Tensor(frame_of(seed, shape))).numpy()
        out[f"{name}_seed"], out[f"{name}_shape"], out[f"{name}_det"] = seed, np.array(shape), det.astype(np.float32)
        print(f"blazeface {name}: {(det[:, 16] != 0).sum()} rows kept")
    np.savez_compressed(os.path.join(OUT, "refrun_blazeface.npz"), weights_seed=555, **out)


SEARCH_CASES = [{}, {"top_k": 3}, {"cam_name": "back"}, {"timestamp": "2026-01-02"}, {"cam_name": "front", "timestamp": "2026-01-01", "top_k": 50},
                {"cam_name": "nowhere"}]


def search_store(seed=31, dim=768):
    """{crop path: (1,dim) unit vector} the way clearcam.py:1282-1287 keys it, plus the query.  Track ids repeat across crops
    (best-per-id rule), some crops have no id, one file is not a .jpg, one day folder is the 'video' folder."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal(dim).astype(np.float32)
    q /= np.linalg.norm(q)
    store = {}
    n = 0
    for cam in ("front", "back", "yard"):
        for day in ("2026-01-01", "2026-01-02", "video"):
            for j in range(5):
                n += 1
                oid = (n * 7) % 11                                    # ids collide across cameras and days on purpose
                name = f"{1700000000 + n}.0_{oid}_{n % 3}.jpg" if j != 3 else f"snapshot{n}.jpg"
                if j == 4 and cam == "yard":
                    name = f"notes{n}.txt"
                v = q * rng.uniform(0.0, 0.6) + rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
                store[f"data/cameras/{cam}/objects/{day}/{name}"] = (v / np.linalg.norm(v)).astype(np.float32)[None]
    return store, q


def run_search():
    import json
    import pickle
    import tempfile
    from models.objects import ObjectFinder
    store, q = search_store()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        folders = {}
        for path, e in store.items():
            folders.setdefault(os.path.dirname(path), {})[path] = e
        for folder, emb in folders.items():                           # one embeddings.pkl per camera/day folder (clearcam.py:1286)
            os.makedirs(folder)
            with open(os.path.join(folder, "embeddings.pkl"), "wb") as f:
                pickle.dump({"embeddings": emb}, f)
        finder = ObjectFinder("data/cameras")
        finder._load_all_embeddings()
        assert len(finder.image_embeddings) == len(store)
        results = [finder.search(text_embedding=q, **kw) for kw in SEARCH_CASES]
        os.chdir(REF)
    paths = sorted(store)
    np.savez_compressed(os.path.join(OUT, "refrun_search.npz"), paths=np.array(paths), embs=np.concatenate([store[p] for p in paths]), query=q,
                        cases=json.dumps(SEARCH_CASES), results=json.dumps([[[p, float(s)] for p, s in r] for r in results]))
    print("search:", [len(r) for r in results], "top hit", results[0][0])


if __name__ == "__main__":
    runs = {"yolo": run_yolo, "clip": run_clip, "adaface": run_adaface, "blazeface": run_blazeface, "search": run_search}
    for w in sys.argv[1:] or list(runs):
        runs[w]()
