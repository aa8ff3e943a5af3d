# dev tool (GPU): every block of the 16-bit path recomputed on the CPU FROM THE GPU'S OWN INPUTS to that block (exact float32 weights, f32
# accumulation, f16 rounding where the kernels round - oracle/lowprec_oracle.py) and compared with the GPU's output of the block: a faithful
# kernel differs only where an f32 sum lands within accumulation-order noise of a rounding boundary (~1 % of the elements, by one ulp).
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CLEARCAM_TAP_BLOCKS"] = "1"
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
import oracle.yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle
fi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = sys.argv[2] if len(sys.argv) > 2 else "f16s"
sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
fr = np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)[fi:fi + 1]
class ActOnly(LowPrecOracle):                     # exact float32 weights, f16 activations
    def __init__(self):
        yo.YOLOv9Oracle.__init__(self, "c", 640, sd); self.t = torch.float16
e = ActOnly()
m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
m.detect_batch(fr)
g = {b: torch.from_numpy(m.get_tensor({15: "p3", 18: "p4", 21: "p5"}.get(b, f"b{b}"))).permute(0, 3, 1, 2).contiguous() for b in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 18, 19, 21)}
P = "model.list."
up = yo.YOLOv9Oracle.upsample
with torch.no_grad():
    want = {2: e.elan4(g[1], P + "2"), 3: e.down(g[2], P + "3"), 4: e.elan4(g[3], P + "4"), 5: e.down(g[4], P + "5"), 6: e.elan4(g[5], P + "6"),
            7: e.down(g[6], P + "7"), 8: e.elan4(g[7], P + "8"), 9: e.sppelan(g[8], P + "9"), 12: e.elan4(torch.cat((up(g[9]), g[6]), 1), P + "12"),
            15: e.elan4(torch.cat((up(g[12]), g[4]), 1), P + "15"), 16: e.down(g[15], P + "16"), 18: e.elan4(torch.cat((g[16], g[12]), 1), P + "18"),
            19: e.down(g[18], P + "19"), 21: e.elan4(torch.cat((g[19], g[9]), 1), P + "21")}
for b, w in want.items():
    d = (g[b] - w)
    print(f"{dt} block {b:2d} from the GPU's inputs: elements differing {float((d != 0).float().mean()):.4f}  rel rms {float(torch.sqrt((d ** 2).mean() / (w ** 2).mean())):.2e}  "
          f"max|d| {float(d.abs().max()):.3e}  mean d {float(d.mean()):+.2e}", flush=True)
