# dev tool (GPU): per-block comparison of the f16s path with (a) the f32 oracle and (b) the storage-rounding emulation with exact weights
# (f16 activations rounded where the kernels round) on one frame: where do the GPU's activations leave the emulation's?
import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CLEARCAM_TAP_BLOCKS"] = "1"
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
import oracle.yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle
fi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
fr = np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)[fi:fi + 1]
class ActOnly(LowPrecOracle):                     # exact float32 weights, f16 activations
    def __init__(self):
        yo.YOLOv9Oracle.__init__(self, "c", 640, sd); self.t = torch.float16
o, e = yo.YOLOv9Oracle("c", 640, sd), ActOnly()
with torch.no_grad():
    o.decode(o.head_raw(o.features(o.network_input(fr)))); e.decode(e.head_raw(e.features(e.network_input(fr))))
coff = {2: 0, 3: 0, 4: 0, 5: 0, 6: 0, 7: 0, 8: 0, 9: 0, 12: 0, 16: 0, 19: 0}
for dt in sys.argv[2:] or ["f16s"]:
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
    m.detect_batch(fr)
    for b in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 18, 21):
        name = {15: "p3", 18: "p4", 21: "p5"}.get(b, f"b{b}")
        g = torch.from_numpy(m.get_tensor(name)).permute(0, 3, 1, 2)
        r, q = o.block_outputs[b], e.block_outputs[b]
        # the view may sit at a channel offset of a wider buffer: find it by the best match
        best = None
        for c0 in range(0, g.shape[1] - r.shape[1] + 1, 8):
            d = float((g[:, c0:c0 + r.shape[1]] - q).abs().mean())
            if best is None or d < best[0]: best = (d, c0)
        gg = g[:, best[1]:best[1] + r.shape[1]]
        rel = lambda a, b_: float(torch.sqrt(((a - b_) ** 2).mean() / (b_ ** 2).mean()))    # noqa: E731
        print(f"{dt} block {b:2d} ({tuple(r.shape[1:])}, coff {best[1]}): gpu vs f32 {rel(gg, r):.2e}  emu vs f32 {rel(q, r):.2e}  gpu vs emu {rel(gg, q):.2e}  "
              f"elements differing gpu/emu {float((gg != q).float().mean()):.4f}  max|gpu-emu| {float((gg - q).abs().max()):.3e} (max|x| {float(r.abs().max()):.2f})", flush=True)
    m.close()
