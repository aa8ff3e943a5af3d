"""Development probe: how well conditioned is a synthetic YOLOv9 checkpoint for end-to-end 16-bit tests?

    python tools/dev/cond_eval.py [c] [chaotic|conditioned] [frames]

Prints (a) the f32 perturbation gain (relative RMS of p3/p4/p5 after adding white noise of relative RMS 1e-3 to the
network input) and (b) what the 16-bit storage emulation (oracle/lowprec_oracle.py) does to features and detections.
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


def evaluate(size, sd, frames, res=640, dtypes=("bf16", "f16"), verbose=True):
    out = {}
    o = YOLOv9Oracle(size, res, sd)
    with torch.no_grad():
        x = o.network_input(frames)
        feats = [f.clone() for f in o.features(x)]
        ref = o.detect_batch(frames)
        g = torch.Generator().manual_seed(5)
        noise = torch.randn(x.shape, generator=g)
        xp = x + noise * (1e-3 * float(x.pow(2).mean().sqrt()))
        fp = o.features(xp)
        out["gain"] = [rel_rms(a, b) / 1e-3 for a, b in zip(fp, feats)]
    nref = [(r[:, 4] > 0).sum() for r in ref]
    out["n_ref"] = [int(n) for n in nref]
    near = np.concatenate([r[:, 4][(r[:, 4] > 0)] for r in ref])
    out["frac_scores_within_1e-2_of_thr"] = float((near < 0.26).mean()) if len(near) else 0.0
    for dt in dtypes:
        lo = LowPrecOracle(size, res, sd, dt)
        with torch.no_grad():
            lf = lo.features(lo.network_input(frames))
            got = lo.detect_batch(frames)
        rr = [rel_rms(a, b) for a, b in zip(lf, feats)]
        m9, m5, be, se = [], [], 0.0, 0.0
        for b in range(len(frames)):
            n_ref, n_got, n_match, box_err, sc_err = match_detections(ref[b], got[b], 0.9)
            m9.append(n_match / max(n_ref, n_got, 1)); be = max(be, box_err); se = max(se, sc_err)
            m5.append(match_detections(ref[b], got[b], 0.5)[2] / max(n_ref, n_got, 1))
        out[dt] = {"feat_rel": rr, "match_iou90": m9, "match_iou50": m5, "box_err": be, "score_err": se}
    if verbose:
        for k, v in out.items():
            print(k, v, flush=True)
    return out


if __name__ == "__main__":
    size = sys.argv[1] if len(sys.argv) > 1 else "c"
    kind = sys.argv[2] if len(sys.argv) > 2 else "chaotic"
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    sd = W.synthetic_yolov9_state_dict(size, 1234) if kind == "chaotic" else W.conditioned_yolov9_state_dict(size, 1234)
    frames = np.random.default_rng(1).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    t0 = time.time()
    evaluate(size, sd, frames)
    print("seconds", round(time.time() - t0, 1))
