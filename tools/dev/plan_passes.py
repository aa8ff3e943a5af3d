# dev tool: N+1 replays of the bench plan (YOLOv9-C, B=64, 640x640; storage mode argv[2], default $CLEARCAM_BENCH_DTYPE or f16h) and nothing else - the workload of the PMC passes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("CLEARCAM_BENCH_DTYPE", "f16h")
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype=dtype)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(64, 300, 6, device="cuda")
for _ in range(n + 1):
    m.detect_batch_device(f, o)
torch.cuda.synchronize()
