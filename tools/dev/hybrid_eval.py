# dev tool (GPU): which convs need the low weight plane.  Detections of dtype "f16h" with the plane kept in EVERY conv up to block A and in the
# 1x1 convs up to block B (CLEARCAM_SPLIT_ALL_LAST / CLEARCAM_SPLIT_1X1_LAST; default A = 0: the stem conv, B = 9: the backbone), next to "f16" (no
# conv) and "f16s" (every conv), against the f32 CPU oracle on the three conditioned checkpoints whose float32 weights are NOT pre-rounded; then
# the step time of each.  argv: [frames] [seeds, comma list]; HYBRID_MODES="dtype:A:B,..." overrides the list, FRAME_SEED the frames' seed.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import conditioned_yolov9_state_dict, synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from oracle.yolov9_oracle import YOLOv9Oracle, parity_summary, decoded_rows, tolerance_bars
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "1234,7,99").split(",")]
MODES = [("f16", -1, -1), ("f16h", 0, 9), ("f16h", -1, 9), ("f16h", 0, 6), ("f16h", 2, 9), ("f16h", 9, 9), ("f16s", 99, 99)]
if os.environ.get("HYBRID_MODES"):
    MODES = [(a, int(b), int(c)) for a, b, c in (m.split(":") for m in os.environ["HYBRID_MODES"].split(","))]

def make(sd, dt, all_last, k1_last):
    os.environ["CLEARCAM_SPLIT_ALL_LAST"] = str(all_last); os.environ["CLEARCAM_SPLIT_1X1_LAST"] = str(k1_last)
    return YOLOv9("c", 640, state_dict=sd, dtype=dt)

def label(dt, a, b):
    return f"{dt:5s}" + (f" all<={a:2d} 1x1<={b:2d}" if dt == "f16h" else " " * 17)

for seed in seeds:
    sd = conditioned_yolov9_state_dict("c", seed, exact=False)
    fr = np.random.default_rng(int(os.environ.get("FRAME_SEED", seed + 1))).integers(0, 256, (nf, 640, 640, 3), dtype=np.uint8)
    o = YOLOv9Oracle("c", 640, sd); det, dec = [], []
    with torch.no_grad():
        for i in range(0, nf, 4):
            y = o.decode(o.head_raw(o.features(o.network_input(fr[i:i + 4]))))
            dec.append(decoded_rows(y)); det.append(o.scale_boxes((640, 640), o.postprocess(y), (640, 640)).numpy())
    ref, dec_ref = np.concatenate(det), np.concatenate(dec)
    for dt, a, b in MODES:
        m = make(sd, dt, a, b)
        got = m.detect_batch(fr); d = m.get_tensor("decoded"); m.close()
        s = parity_summary(ref, got, 0.64, dec_ref, d)
        print(f"seed {seed} {label(dt, a, b)} bars {'ok ' if tolerance_bars(s)['all'] else 'NO '}", {k: round(s[k], 4) for k in ("match_frac", "match_frac_clear_of_threshold", "anchor_box_err_px_p99", "anchor_box_err_px_p999", "anchor_box_err_px_max", "anchor_score_err_max")}, flush=True)

sd = synthetic_yolov9_state_dict("c", 1234)
B = 64
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
for dt, a, b in MODES:
    m = make(sd, dt, a, b)
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
    print(f"{label(dt, a, b)}: back-to-back {one*1e3:.3f} ms ({B/one:.0f} frames/s), 3 in flight {many*1e3:.3f} ms ({B/many:.0f} frames/s)", flush=True)
    m.close()
