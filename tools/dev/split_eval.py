"""Development probe (CPU): does f16 storage with SPLIT weights (W = W_hi + W_lo, two f16 planes, power-of-two layer scale) meet the
f32 gate's yardstick on the conditioned checkpoint with UN-ROUNDED weights?

    python tools/dev/split_eval.py [frames] [controlled]

Emulates the storage roundings with oracle/lowprec_oracle.py; the weights are replaced by the 22-bit values the two planes hold.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import weights as W  # noqa: E402
from oracle.lowprec_oracle import LowPrecOracle  # noqa: E402
import oracle.yolov9_oracle as yo  # noqa: E402


def split_f16(w: torch.Tensor) -> torch.Tensor:
    """The value W_hi + W_lo of the two f16 planes (per-tensor power-of-two scale so that the low plane stays a normal number)."""
    m = float(w.abs().max())
    s = 2.0 ** np.floor(np.log2(32768.0 / m)) if m > 0 else 1.0
    ws = w * s
    hi = ws.to(torch.float16).to(torch.float32)
    lo = (ws - hi).to(torch.float16).to(torch.float32)
    return (hi + lo) / s


class SplitOracle(LowPrecOracle):
    def __init__(self, size, res, sd, mode):
        yo.YOLOv9Oracle.__init__(self, size, res, sd)
        self.t = torch.float16
        for k in list(self.sd):
            if k.endswith(".weight") and self.sd[k].ndim == 4 and "dfl" not in k:
                self.sd[k] = split_f16(self.sd[k]) if mode == "split" else self.q(self.sd[k])


def run(o, frames, chunk=4):
    det, dec = [], []
    with torch.no_grad():
        for i in range(0, len(frames), chunk):
            x = o.network_input(frames[i:i + chunk])
            y = o.decode(o.head_raw(o.features(x)))
            dec.append(yo.decoded_rows(y))
            det.append(o.scale_boxes(tuple(x.shape[2:]), o.postprocess(y), frames.shape[1:3]).numpy())
    return np.concatenate(det), np.concatenate(dec)


if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sd = W.conditioned_yolov9_state_dict("c", 1234, exact=False)
    frames = np.random.default_rng(1).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    t0 = time.time()
    ref, dec_ref = run(yo.YOLOv9Oracle("c", 640, sd), frames)
    for mode in ("split", "plain"):
        got, dec = run(SplitOracle("c", 640, sd, mode), frames)
        s = yo.parity_summary(ref, got, 0.64, dec_ref, dec, score_margin=2e-3)
        print(mode, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()}, flush=True)
    print("seconds", round(time.time() - t0, 1))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "controlled":
    # plain f16 storage with the library's controlled weight rounding (yolo.hip round_controlled = LowPrecOracle(feedback=True))
    got, dec = run(LowPrecOracle("c", 640, sd, "f16", feedback=True), frames)
    s = yo.parity_summary(ref, got, 0.64, dec_ref, dec, score_margin=2e-3)
    print("controlled", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()}, flush=True)
