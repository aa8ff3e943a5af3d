# dev tool: every detector size, several resolutions / batch sizes, bf16 and f16 vs f32 on the same frames (gross-error sweep)
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
from oracle.yolov9_oracle import match_detections
rng = np.random.default_rng(5)
for size in (os.environ.get("SIZES", "t,s,m,c,e").split(",")):
    sd = synthetic_yolov9_state_dict(size, 1234)
    for res, (h, w), B in ((640, (640, 640), 3), (960, (540, 960), 2), (320, (180, 320), 5)):
        frames = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        ref = YOLOv9(size, res, state_dict=sd, dtype="f32").detect_batch(frames)
        row = []
        for dt in ("bf16", "f16"):
            got = YOLOv9(size, res, state_dict=sd, dtype=dt).detect_batch(frames)
            assert np.isfinite(got).all()
            m = [match_detections(ref[b], got[b], 0.5) for b in range(B)]      # (n_ref, n_got, n_matched, box_err, score_err)
            row.append(f"{dt}: {sum(x[2] for x in m)}/{sum(x[0] for x in m)} matched")
        print(size, res, (h, w), "B", B, " | ".join(row), flush=True)
