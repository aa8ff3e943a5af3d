"""Restate the reference's own golden data for the CLIP path as a small JSON fixture.

Sources (read here, in the container where /root/reference exists; the GPU box has no reference tree):
  * test/clip_images/embeddings.pkl — two (1,768) f32 image embeddings (numpy-only pickle)
  * test/test_clip.py:12 — cos(text "ferrari f40", image f40.jpg) == 0.330654 (rtol/atol 1e-6)
  * clearcam.py:689 / :48 — search display cut-off 0.21, default text-alert threshold 0.28
Run from the repo root:  python tools/make_reference_pins.py
"""
import hashlib
import json
import os
import pickle

import numpy as np

REF = "/root/reference"
if __name__ == "__main__":
    with open(os.path.join(REF, "test/clip_images/embeddings.pkl"), "rb") as f:
        d = pickle.load(f)
    emb = {os.path.basename(k): np.asarray(v, np.float32) for k, v in (d.get("embeddings", d)).items()}
    out = {"provenance": "roryclear/clearcam test/clip_images/embeddings.pkl, test/test_clip.py:12, clearcam.py:48,689",
           "text_image_cosine_ferrari_f40": 0.330654, "cosine_tolerance": 1e-6,
           "search_display_threshold": 0.21, "default_alert_threshold": 0.28, "embeddings": {}}
    for name, v in emb.items():
        out["embeddings"][name] = {"shape": list(v.shape), "norm": float(np.linalg.norm(v)), "first5": [float(x) for x in v.reshape(-1)[:5]],
                                   "sha1_12": hashlib.sha1(v.tobytes()).hexdigest()[:12], "values": [float(x) for x in v.reshape(-1)]}
    names = sorted(emb)
    out["cos_f40_micra"] = float((emb[names[0]].reshape(-1) * emb[names[1]].reshape(-1)).sum())
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_pins.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, names, out["cos_f40_micra"])
