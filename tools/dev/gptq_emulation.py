# dev tool (CPU): env CALIB = noise | smooth | blocks | lowc (calibration frames), NCAL (how many), DAMP, GPTQ_3X3, SEED (checkpoint), NF (test frames)
# CPU study: covariance-aware (GPTQ-style) rounding of the 1x1 convs' weights to f16, calibrated on 4 noise frames; 3x3 convs keep controlled rounding
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location("se", os.path.join(os.path.dirname(os.path.abspath(__file__)), "split_eval.py")); se = importlib.util.module_from_spec(spec); spec.loader.exec_module(se)
from clearcam_amd import weights as W
import oracle.yolov9_oracle as yo
from oracle.lowprec_oracle import LowPrecOracle
SEED = int(os.environ.get("SEED", "1234")); NF = int(os.environ.get("NF", "64"))
sd = W.conditioned_yolov9_state_dict("c", SEED, exact=False)
frames = np.random.default_rng(SEED + 1).integers(0, 256, (NF, 640, 640, 3), dtype=np.uint8)
calib = np.random.default_rng(4242).integers(0, 256, (int(os.environ.get("NCAL", "4")), 640, 640, 3), dtype=np.uint8)
CAL = os.environ.get("CALIB", "noise")
if CAL == "smooth":                                   # a DIFFERENT input distribution: heavily blurred noise (natural-image-like spectrum), contrast stretched
    import torch.nn.functional as F
    t = torch.from_numpy(calib).float().permute(0, 3, 1, 2)
    for _ in range(3): t = F.avg_pool2d(F.pad(t, (8, 8, 8, 8), mode="reflect"), 17, 1)
    t = (t - t.mean((2, 3), keepdim=True)) / t.std((2, 3), keepdim=True) * 50 + 128
    calib = t.clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy().copy()
elif CAL == "lowc":                                   # white noise of a quarter of the contrast around mid-grey
    calib = (128 + (calib.astype(np.int32) - 128) // 4).astype(np.uint8)
elif CAL == "blocks":                                 # piecewise-constant 32x32 blocks of random colour
    small = np.random.default_rng(77).integers(0, 256, (len(calib), 20, 20, 3), dtype=np.uint8)
    calib = np.repeat(np.repeat(small, 32, 1), 32, 2)
print("calibration frames:", CAL, calib.shape, float(calib.std()), flush=True)
ref, dec_ref = se.run(yo.YOLOv9Oracle("c", 640, sd), frames)
oc = yo.YOLOv9Oracle("c", 640, sd); H = {}
orig = oc._conv2d
def hooked(x, name, stride=1, groups=1):
    w = oc.sd[name + ".weight"]
    if name not in H and groups == 1 and (w.shape[2] == 1 or os.environ.get("GPTQ_3X3")):
        k = w.shape[2]
        if k == 1:
            X = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        else:
            X = torch.nn.functional.unfold(x, k, padding=k // 2, stride=stride).permute(0, 2, 1).reshape(-1, x.shape[1] * k * k)
        rows = int(os.environ.get("ROWS", "200000")) if k == 1 else 40000
        if X.shape[0] > rows: X = X[torch.randperm(X.shape[0], generator=torch.Generator().manual_seed(0))[:rows]]
        X = X.double()
        H[name] = (X.T @ X / X.shape[0])
    return orig(x, name, stride, groups)
oc._conv2d = hooked
with torch.no_grad():
    oc.decode(oc.head_raw(oc.features(oc.network_input(calib))))
print("calibrated", len(H), "convs", flush=True)
def gptq(w, Hm, damp=float(os.environ.get("DAMP", "0.01"))):
    """w (co, ci) f32 -> f16-representable (co, ci); column by column with error feedback through the inverse Hessian (GPTQ)."""
    Wm = w.double().clone(); ci = Wm.shape[1]
    Hd = Hm.clone(); Hd += torch.eye(ci, dtype=torch.float64) * damp * Hd.diag().mean()
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    Q = torch.empty_like(Wm)
    for i in range(ci):
        q = Wm[:, i].float().to(torch.float16).double()
        Q[:, i] = q
        err = (Wm[:, i] - q) / Hinv[i, i]
        if i + 1 < ci: Wm[:, i + 1:] -= err[:, None] * Hinv[i, i + 1:][None, :]
    return Q.float()
emu = LowPrecOracle.__new__(LowPrecOracle); emu.t = torch.float16
o0 = yo.YOLOv9Oracle("c", 640, sd)
names = [k[:-len(".weight")] for k, v in o0.sd.items() if k.endswith(".weight") and v.ndim == 4 and "dfl" not in k]
t0 = time.time(); plain, gq, spl = {}, {}, {}
for n in names:
    w = o0.sd[n + ".weight"]
    plain[n] = emu.q_feedback(w); spl[n] = se.split_f16(w)
    if n in H:
        gq[n] = gptq(w.reshape(w.shape[0], -1), H[n]).reshape(w.shape)
        # report the layer-level proxy: expected squared output error under the calibration covariance
print("rounded in", round(time.time() - t0), "s", flush=True)
def proxy(n, q):
    d = (q - o0.sd[n + ".weight"]).reshape(q.shape[0], -1).double(); return float((d @ H[n] * d).sum())
tot_p = sum(proxy(n, plain[n]) for n in H); tot_g = sum(proxy(n, gq[n]) for n in H)
nearest = {n: o0.sd[n + ".weight"].to(torch.float16).float() for n in H}
print("sum over the calibrated convs of E|dW x|^2 on the calibration frames: nearest %.3e  controlled %.3e  gptq %.3e" % (sum(proxy(n, nearest[n]) for n in H), tot_p, tot_g), flush=True)
blk = lambda n: int(n.split(".")[2])
class O(LowPrecOracle):
    def __init__(self, choose):
        yo.YOLOv9Oracle.__init__(self, "c", 640, sd); self.t = torch.float16
        for n in names: self.sd[n + ".weight"] = choose(n)
cfgs = {
 "controlled rounding, no split": lambda n: plain[n],
 "GPTQ, no split": lambda n: gq.get(n, plain[n]),
 "GPTQ + stem split": lambda n: spl[n] if blk(n) == 0 else gq.get(n, plain[n]),
 "all split": lambda n: spl[n],
}
for name, ch in cfgs.items():
    got, dec = se.run(O(ch), frames)
    s = yo.parity_summary(ref, got, 0.64, dec_ref, dec, score_margin=2e-3)
    print(f"seed {SEED} {name:36s}", {k: round(s[k], 4) for k in ("match_frac_clear_of_threshold", "anchor_box_err_px_p50", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max")}, flush=True)
