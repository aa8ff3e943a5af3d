# dev tool (GPU): which natural-statistics frames carry the f16 tail on the "nat" checkpoint, and by how much per mode
# (reference = the library's f32 mode).  Prints frame indices so that a CPU emulation (oracle/lowprec_oracle.py) can replay one.
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.streams import natural_frames
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fr = natural_frames(nf, 640, 640, seed=2234)
sd = conditioned_yolov9_state_dict("c", 1234, exact=False, stress="nat")
def run(dt):
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt); dec = []
    for i in range(0, nf, 64):
        m.detect_batch(fr[i:i + 64]); dec.append(m.get_tensor("decoded"))
    feats = {n: m.get_tensor(n) for n in ("p3", "p4", "p5")}   # last chunk only
    m.close(); return np.concatenate(dec), feats
ref, fref = run("f32")
for dt in ("f16s", "f16h", "f16"):
    got, fg = run(dt)
    both = (ref[..., 4] > 0) & (got[..., 4] > 0)
    ae = np.where(both, np.abs(ref[..., :4] - got[..., :4]).max(-1), 0.0)
    bad = (ae > 0.64).sum(1)
    idx = np.nonzero(bad)[0]
    print(dt, "bad frames", [(int(i), int(bad[i]), round(float(ae[i].max()), 1)) for i in idx])
    print(dt, "feature rel rms (last chunk)", {n: round(float(np.sqrt(((fg[n] - fref[n]) ** 2).mean() / (fref[n] ** 2).mean())), 5) for n in fg},
          "abs max", {n: round(float(np.abs(fref[n]).max()), 1) for n in fg})
    # per-frame feature error of P5 in the last chunk: is the bad frame bad already in the features?
    e = np.sqrt(((fg["p5"] - fref["p5"]) ** 2).mean((1, 2, 3)) / (fref["p5"] ** 2).mean((1, 2, 3)))
    print(dt, "p5 rel rms per frame of the last chunk: max", round(float(e.max()), 5), "median", round(float(np.median(e)), 5), "argmax", int(e.argmax()) + nf - len(e))
