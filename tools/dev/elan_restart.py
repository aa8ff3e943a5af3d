# dev tool (GPU): block_restart.py inside ONE RepNCSPELAN4 (block 6 = the third one: cat2, csp4, csp5), conv by conv
import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CLEARCAM_TAP_BLOCKS"] = "1"; os.environ["CLEARCAM_TAP_CSP"] = "1"; os.environ["CLEARCAM_FUSE_CSP"] = "0"
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
import oracle.yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle
fi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = sys.argv[2] if len(sys.argv) > 2 else "f16s"
sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
fr = np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)[fi:fi + 1]
class ActOnly(LowPrecOracle):
    def __init__(self):
        yo.YOLOv9Oracle.__init__(self, "c", 640, sd); self.t = torch.float16
e = ActOnly()
m = YOLOv9("c", 640, state_dict=sd, dtype=dt)
m.detect_batch(fr)
T = lambda n: torch.from_numpy(m.get_tensor(n)).permute(0, 3, 1, 2).contiguous()      # noqa: E731
x, out, cat = T("b5"), T("b6"), T("cat2")
h = 128
def cmp(name, got, want):
    d = got - want
    print(f"{dt} {name:44s} differing {float((d != 0).float().mean()):.4f}  rel rms {float(torch.sqrt((d ** 2).mean() / (want ** 2).mean())):.2e}  max|d| {float(d.abs().max()):.3e}", flush=True)
P = "model.list.6"
with torch.no_grad():
    cmp("cv1 (1x1 512->512)", cat[:, :4 * h], e.conv(x, P + ".cv1"))
    for k, (src, dst) in enumerate((((2 * h, 4 * h), (4 * h, 6 * h)), ((4 * h, 6 * h), (6 * h, 8 * h)))):
        br = f"{P}.cv{2 + k}"; r = br + ".list.0"; q = r + ".m.list.0"
        y1 = cat[:, src[0]:src[1]]
        ab, t, u = T(f"csp{4 + k}_ab"), T(f"csp{4 + k}_t"), T(f"csp{4 + k}_u")
        a0 = e.conv(y1, r + ".cv1")                       # the GPU's a was overwritten in place by the residual update: recompute it
        cmp(f"cv{2+k}: RepNCSP cv2 (b)", ab[:, h:], e.conv(y1, r + ".cv2"))
        cmp(f"cv{2+k}: bottleneck cv1 (t), from recomputed a", t, e.conv(a0, q + ".cv1"))
        cmp(f"cv{2+k}: a + cv2(t), from the GPU's t", ab[:, :h], e.q(a0 + e._silu_conv_f32(t, q + ".cv2")))
        cmp(f"cv{2+k}: RepNCSP cv3 (u)", u, e.conv(ab, r + ".cv3"))
        cmp(f"cv{2+k}: trailing 3x3", cat[:, dst[0]:dst[1]], e.conv(u, br + ".list.1"))
    cmp("cv4 (1x1 over the concat)", out, e.conv(cat, P + ".cv4"))
