# dev tool (GPU): dtype "f16c" on a conditioned checkpoint (un-rounded float32 weights) against the f32 CPU oracle, by calibration-frame kind
# (None = the library's seeded noise, noise / smooth / blocks handed over through cc_yolo_calibrate).  argv: [seed] [frames];  env CLEARCAM_CALIB_DAMP
import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import conditioned_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from oracle.yolov9_oracle import YOLOv9Oracle, parity_summary, decoded_rows, tolerance_bars
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["none", "noise", "smooth", "blocks"]
def cal(kind, n=4):
    fr = np.random.default_rng(4242).integers(0, 256, (n, 640, 640, 3), dtype=np.uint8)
    if kind == "smooth":
        x = torch.from_numpy(fr).float().permute(0, 3, 1, 2)
        for _ in range(3): x = F.avg_pool2d(F.pad(x, (8, 8, 8, 8), mode="reflect"), 17, 1)
        x = (x - x.mean((2, 3), keepdim=True)) / x.std((2, 3), keepdim=True) * 50 + 128
        fr = x.clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy().copy()
    elif kind == "blocks":
        fr = np.ascontiguousarray(np.repeat(np.repeat(fr[:, ::32, ::32], 32, 1), 32, 2))
    return fr
sd = conditioned_yolov9_state_dict("c", seed, exact=False)
fr = np.random.default_rng(seed + 1).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
o = YOLOv9Oracle("c", 640, sd); det, dec = [], []
with torch.no_grad():
    for i in range(0, nf, 4):
        y = o.decode(o.head_raw(o.features(o.network_input(fr[i:i + 4]))))
        dec.append(decoded_rows(y)); det.append(o.scale_boxes((640, 640), o.postprocess(y), (640, 640)).numpy())
ref, dec_ref = np.concatenate(det), np.concatenate(dec)
keys = ("match_frac", "match_frac_clear_of_threshold", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max", "anchor_score_err_max")
for dt, kind in [("f16", None), ("f16h", None)] + [("f16c", k) for k in kinds]:
    m = YOLOv9("c", 640, state_dict=sd, dtype=dt, calibration_frames=None if kind in (None, "none") else cal(kind))
    got = m.detect_batch(fr); d = m.get_tensor("decoded"); m.close()
    p = parity_summary(ref, got, 0.64, dec_ref, d)
    print(f"damp {os.environ.get('CLEARCAM_CALIB_DAMP', '0.01')} checkpoint {seed} {dt} calib {kind}: bars {tolerance_bars(p)}", {k: round(float(p[k]), 4) for k in keys}, flush=True)
