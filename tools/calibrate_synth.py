"""One-off tool: LSUV-style per-conv scale table for the seeded synthetic YOLOv9 weights.

A 144-conv SiLU stack without normalisation either explodes or dies under any fixed init gain,
so ``clearcam_amd.weights.synthetic_yolov9_state_dict`` multiplies each seeded N(0,1/fan_in)
weight by a committed per-conv scale.  This script derives those scales by running the CPU
oracle on a seeded noise batch and normalising every conv's pre-activation std in execution
order.  Output: clearcam_amd/assets/synth_scales.json (committed; the generator itself needs
neither torch nor the oracle).  Run from the repo root:  python tools/calibrate_synth.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearcam_amd import weights as W  # noqa: E402
from oracle.yolov9_oracle import YOLOv9Oracle  # noqa: E402


def calibrate(size: str, seed: int = 1234, res: int = 640):
    sd = W.synthetic_yolov9_state_dict(size, seed, scales={})
    o = YOLOv9Oracle(size, res, sd)
    scales = {}
    orig = o._conv2d

    def hooked(x, name, stride=1, groups=1):
        y = orig(x, name, stride, groups)
        if name not in scales:
            target = 1.0
            if ".m.list." in name and name.endswith("cv2.conv"):
                target = 0.6                      # residual branch
            is_cls = (".cv3.list" in name) and name.endswith(".list.2") and ("model.list.22." in name or "model.list.42." in name)
            if "cv2.list" in name and name.endswith(".list.2") and ("model.list.22." in name or "model.list.42." in name):
                target = 2.0                      # DFL logits
            b = o.sd[name + ".bias"]
            z = y - b.view(1, -1, 1, 1)
            s = float(target / z.std())
            if is_cls:
                # activations of a random unnormalised net are heavy-tailed: place the 1-3e-4
                # quantile of the class logits at the 0.25-score line (logit -1.1 given bias -5)
                s = float(3.9 / torch.quantile(z.flatten()[::7], 1.0 - 3e-4))
            scales[name] = s
            o.sd[name + ".weight"] *= s
            y = orig(x, name, stride, groups)
        return y

    o._conv2d = hooked
    fr = np.random.default_rng(7).integers(0, 256, (4, res, res, 3), dtype=np.uint8)
    with torch.no_grad():
        o.head_raw(o.features(o.network_input(fr)))
    return scales


if __name__ == "__main__":
    out = {s: calibrate(s) for s in "tsmce"}
    path = os.path.join(os.path.dirname(W.__file__), "assets", "synth_scales.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})
