# dev tool (GPU): which blocks need the low weight plane.  Detections of dtype "f16h" with the low plane kept up to block CLEARCAM_SPLIT_LAST
# (4, 6, 9 = the backbone, 12), next to "f16" (no block) and "f16s" (every block), against the f32 CPU oracle on conditioned checkpoints whose
# float32 weights are NOT pre-rounded; then the step time of each.  argv: [frames] [seeds, comma list]
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import conditioned_yolov9_state_dict, synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from oracle.yolov9_oracle import YOLOv9Oracle, parity_summary, decoded_rows
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "1234,7,99").split(",")]
MODES = [("f16", None, 0), ("f16h", 9, 0), ("f16h", 9, 1), ("f16h", 12, 0), ("f16h", 12, 1), ("f16h", 15, 0), ("f16h", 21, 0), ("f16s", None, 0)]
if os.environ.get("HYBRID_MODES"):                   # "dtype:last:rest_k,..."
    MODES = [(a, None if b == "" else int(b), int(c)) for a, b, c in (m.split(":") for m in os.environ["HYBRID_MODES"].split(","))]

def make(sd, dt, last, rest_k=0):
    if last is None: os.environ.pop("CLEARCAM_SPLIT_LAST", None)
    else: os.environ["CLEARCAM_SPLIT_LAST"] = str(last)
    os.environ["CLEARCAM_SPLIT_REST_K"] = str(rest_k)
    os.environ["CLEARCAM_FUSE_CSP"] = "0" if rest_k else "2"        # a fused RepNCSP takes one split flag for its four convs (same rows either way, not the same time)
    return YOLOv9("c", 640, state_dict=sd, dtype=dt)

for seed in seeds:
    sd = conditioned_yolov9_state_dict("c", seed, exact=False)
    fr = np.random.default_rng(int(os.environ.get("FRAME_SEED", seed + 1))).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    o = YOLOv9Oracle("c", 640, sd); det, dec = [], []
    with torch.no_grad():
        for i in range(0, nf, 4):
            y = o.decode(o.head_raw(o.features(o.network_input(fr[i:i + 4]))))
            dec.append(decoded_rows(y)); det.append(o.scale_boxes((640, 640), o.postprocess(y), (640, 640)).numpy())
    ref, dec_ref = np.concatenate(det), np.concatenate(dec)
    for dt, last, rk in MODES:
        m = make(sd, dt, last, rk)
        got = m.detect_batch(fr); d = m.get_tensor("decoded"); m.close()
        s = parity_summary(ref, got, 0.64, dec_ref, d)
        e = np.abs(d[..., :4] - dec_ref[..., :4]).max(-1)[(d[..., 4] > 0) & (dec_ref[..., 4] > 0)]
        print(f"seed {seed} {dt}{'' if last is None else ' <=' + str(last):6s}{' +1x1' if rk == 1 else '':5s} n {len(e)} p99.9 {np.quantile(e, 0.999):.3f} over0.5 {(e > 0.5).sum()}", {k: round(s[k], 4) for k in ("match_frac", "match_frac_clear_of_threshold", "anchor_box_err_px_p99", "anchor_box_err_px_max", "anchor_score_err_max")}, flush=True)

sd = synthetic_yolov9_state_dict("c", 1234)
B = 64
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
for dt, last, rk in MODES:
    m = make(sd, dt, last, rk)
    o = torch.empty(B, 300, 6, device="cuda")
    for _ in range(5): m.detect_batch_device(f, o)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): m.detect_batch_device(f, o)
    torch.cuda.synchronize(); one = (time.perf_counter() - t) / 20
    m.set_in_flight(3)
    outs = [torch.empty(B, 300, 6, device="cuda") for _ in range(3)]
    for i in range(6): m.wait(m.submit(f, outs[i % 3]))
    torch.cuda.synchronize(); t = time.perf_counter()
    tk = [m.submit(f, outs[i % 3]) for i in range(30)]
    for k in tk[-3:]: m.wait(k)
    torch.cuda.synchronize(); many = (time.perf_counter() - t) / 30
    print(f"{dt}{'' if last is None else ' <=' + str(last):6s}{' +1x1' if rk == 1 else '':5s}: back-to-back {one*1e3:.3f} ms ({B/one:.0f} frames/s), 3 in flight {many*1e3:.3f} ms ({B/many:.0f} frames/s)", flush=True)
    m.close()
