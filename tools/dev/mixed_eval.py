"""Development probe (CPU): which parts of the detector must leave 16-bit storage for the speed modes to meet the f32 gate's
own yardstick (matched boxes within 1e-3 * max(H, W) px, >= 99 % one-to-one matches)?

    python tools/dev/mixed_eval.py [frames] [exact|raw]

Emulates storage rounding with oracle/lowprec_oracle.py and lifts chosen convs back to f32 (weights and outputs):
  full        every conv in the storage type (the r02 speed modes)
  box         DDetect's box branch cv2[i] (detection/yolov9.py:202-207) in f32: its 3x3 entry conv reads the 16-bit P3..P5
  head        both DDetect branches in f32
`raw` uses the conditioned checkpoint WITHOUT the rounding of its weights to bf16/f16-exact values (weights.py::_storage_exact).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import weights as W  # noqa: E402
from oracle.lowprec_oracle import LowPrecOracle, rel_rms  # noqa: E402
from oracle.yolov9_oracle import YOLOv9Oracle, match_detections  # noqa: E402


class MixedOracle(LowPrecOracle):
    def __init__(self, size, res, sd, dtype, f32_parts=()):
        YOLOv9Oracle.__init__(self, size, res, sd)
        self.t = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
        self.f32_parts = tuple(f32_parts)
        for k in list(self.sd):
            if k.endswith(".weight") and self.sd[k].ndim == 4 and "dfl" not in k and not self._is_f32(k):
                self.sd[k] = self.q(self.sd[k])

    def _is_f32(self, name):
        return any(p in name for p in self.f32_parts)

    def conv(self, x, name, stride=1, groups=1):
        y = YOLOv9Oracle.conv(self, x, name, stride, groups)
        return y if self._is_f32(name) else self.q(y)


def evaluate(sd, frames, variants):
    o = YOLOv9Oracle("c", 640, sd)
    with torch.no_grad():
        ref = o.detect_batch(frames)
        feats = [f.clone() for f in o.features(o.network_input(frames))]
    for label, dt, parts in variants:
        lo = MixedOracle("c", 640, sd, dt, parts)
        with torch.no_grad():
            got = lo.detect_batch(frames)
            rr = [rel_rms(a, b) for a, b in zip(lo.features(lo.network_input(frames)), feats)]
        tot, be, se, errs = [0, 0, 0], 0.0, 0.0, []
        for b in range(len(frames)):
            a, c, k, e, s = match_detections(ref[b], got[b], 0.9)
            tot[0] += a; tot[1] += c; tot[2] += k; be = max(be, e); se = max(se, s); errs.append(e)
        print(f"{label:22s} matched {tot[2]}/{max(tot[0], tot[1])} = {tot[2] / max(tot[0], tot[1], 1):.4f}  max box err {be:.3f} px  "
              f"score err {se:.2e}  feat rel {rr[0]:.2e}/{rr[1]:.2e}/{rr[2]:.2e}  per-frame box err {np.round(errs, 2).tolist()}", flush=True)


if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    kind = sys.argv[2] if len(sys.argv) > 2 else "exact"
    if kind == "raw":
        W._storage_exact = lambda w: np.asarray(w, np.float32)
    sd = W.conditioned_yolov9_state_dict("c", 1234)
    frames = np.random.default_rng(1).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    t0 = time.time()
    evaluate(sd, frames, [("bf16 full", "bf16", ()), ("bf16 box-f32", "bf16", ("model.list.22.cv2.",)), ("bf16 head-f32", "bf16", ("model.list.22.cv2.", "model.list.22.cv3.")),
                          ("f16 full", "f16", ()), ("f16 box-f32", "f16", ("model.list.22.cv2.",))])
    print("seconds", round(time.time() - t0, 1))
