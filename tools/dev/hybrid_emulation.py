# dev tool (CPU): which convs need the low weight plane, in the storage-rounding emulation (oracle/lowprec_oracle.py) - the CPU-side companion of hybrid_eval.py.  env SEED (checkpoint), NF (frames)
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location("se", os.path.join(os.path.dirname(os.path.abspath(__file__)), "split_eval.py")); se = importlib.util.module_from_spec(spec); spec.loader.exec_module(se)
from clearcam_amd import weights as W
import oracle.yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle
SEED = int(os.environ.get("SEED", "1234")); NF = int(os.environ.get("NF", "96"))
sd = W.conditioned_yolov9_state_dict("c", SEED, exact=False)
frames = np.random.default_rng(SEED + 1).integers(0, 256, (NF, 640, 640, 3), dtype=np.uint8)
ref, dec_ref = se.run(yo.YOLOv9Oracle("c", 640, sd), frames)
emu = LowPrecOracle.__new__(LowPrecOracle); emu.t = torch.float16
o0 = yo.YOLOv9Oracle("c", 640, sd)
ctrl, spl = {}, {}
for k, v in o0.sd.items():
    if k.endswith(".weight") and v.ndim == 4 and "dfl" not in k:
        ctrl[k] = emu.q_feedback(v); spl[k] = se.split_f16(v)
blk = lambda n: int(n.split(".")[2])
class O(LowPrecOracle):
    def __init__(self, pick):
        yo.YOLOv9Oracle.__init__(self, "c", 640, sd); self.t = torch.float16; self.n = 0
        for k in list(self.sd):
            if k in ctrl:
                if pick(k, self.sd[k]): self.sd[k] = spl[k]; self.n += 1
                else: self.sd[k] = ctrl[k]
k1 = lambda w: w.shape[2] == 1
inner = lambda k: ".list.0.cv" in k and (".cv2.list.0." in k or ".cv3.list.0." in k)     # RepNCSP's own cv1 / cv2 / cv3
subsets = {
 "f16h final: stem + backbone 1x1": lambda k, w: blk(k) == 0 or (k1(w) and blk(k) <= 9),
 "... minus RepNCSP-internal 1x1 of blocks 2, 4": lambda k, w: blk(k) == 0 or (k1(w) and blk(k) <= 9 and not (inner(k) and blk(k) in (2, 4))),
 "... minus every RepNCSP-internal 1x1": lambda k, w: blk(k) == 0 or (k1(w) and blk(k) <= 9 and not inner(k)),
 "stem + 1x1 of blocks <= 6": lambda k, w: blk(k) == 0 or (k1(w) and blk(k) <= 6),
 "backbone 1x1 only (stem not)": lambda k, w: k1(w) and blk(k) <= 9,
 "all split": lambda k, w: True,
 "none": lambda k, w: False,
}
for name, pick in subsets.items():
    o = O(pick)
    got, dec = se.run(o, frames)
    s = yo.parity_summary(ref, got, 0.64, dec_ref, dec, score_margin=2e-3)
    print(f"seed {SEED} {name:48s} split convs {o.n:3d}", {k: round(s[k], 4) for k in ("match_frac_clear_of_threshold", "anchor_box_err_px_p50", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max")}, flush=True)
