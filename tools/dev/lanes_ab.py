"""dev tool: the captured plan with its lanes as concurrent streams (default) against one chain (CLEARCAM_LANES=0), same process,
interleaved timing, outputs compared bit for bit.   python tools/dev/lanes_ab.py [dtype] [size]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
size = sys.argv[2] if len(sys.argv) > 2 else "c"
sd = synthetic_yolov9_state_dict(size, 1234)
for B in (64, 16, 8, 4, 1):
    f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    o = torch.empty(B, 300, 6, device="cuda")
    models, outs = [], {}
    for form, env in (("chain", "0"), ("heads", "2"), ("heads1", "4")):
        os.environ["CLEARCAM_LANES"] = env
        m = YOLOv9(size, 640, state_dict=sd, dtype=dtype)
        for _ in range(3):
            m.detect_batch_device(f, o)                 # the plan is built (and reads the switch) on the first call
        torch.cuda.synchronize()
        outs[form] = o.clone()
        models.append((form, m))
    times = {form: [] for form, _ in models}
    n = 10 if B == 64 else 100
    for _ in range(7):
        for form, m in models:
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(n):
                m.detect_batch_device(f, o)
            torch.cuda.synchronize()
            times[form].append((time.perf_counter() - t) / n * 1e3)
    for form, m in models:
        ts = sorted(times[form])
        print(f"yolov9-{size} {dtype} B={B:2} {form:6} median {ts[len(ts) // 2]:.3f} ms/step  min {ts[0]:.3f}   detections {int((outs[form][..., 4] > 0).sum())}  "
              f"bit-identical to chain: {bool(torch.equal(outs[form], outs['chain']))}", flush=True)
        m.close()
