# dev tool: N batches of the bench plan through D slots of one handle (the workload of tools/dev/in_flight_trace.sh)
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype="f16")
if depth > 1:
    m.set_in_flight(depth)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)).cuda()
o = [torch.empty(64, 300, 6, device="cuda") for _ in range(depth)]
for k in range(depth + n):
    m.submit(f, o[k % depth])
torch.cuda.synchronize()
