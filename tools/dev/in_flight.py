"""dev tool: batches in flight (cc_yolo_set_in_flight / submit / wait) against back-to-back cc_yolo_detect calls, one handle.
    python tools/dev/in_flight.py [dtype] [B] [size]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
size = sys.argv[3] if len(sys.argv) > 3 else "c"
DMAX, N = 6, 24
sd = synthetic_yolov9_state_dict(size, 1234)
f = [torch.from_numpy(np.random.default_rng(i).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda() for i in range(DMAX)]
o = [torch.empty(B, 300, 6, device="cuda") for _ in range(DMAX)]
m = YOLOv9(size, 640, state_dict=sd, dtype=dtype)
ref = []
for i in range(DMAX):
    m.detect_batch_device(f[i], o[i]); torch.cuda.synchronize(); ref.append(o[i].clone())


def serial():
    for k in range(N):
        m.detect_batch_device(f[k % DMAX], o[k % DMAX])


def piped(depth):
    def go():
        for k in range(N):
            m.submit(f[k % depth], o[k % depth])
    return go


for depth in (0, 1, 2, 3, 4, 6):
    if depth:
        m.set_in_flight(depth)
    fn = serial if depth == 0 else piped(depth)
    for o_ in o:
        o_.zero_()
    fn(); torch.cuda.synchronize()                           # builds the slots' plans
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / N * 1e3)
    n = DMAX if depth == 0 else depth
    same = all(bool(torch.equal(o[i], ref[i])) for i in range(n))
    ts.sort()
    print(f"yolov9-{size} {dtype} B={B} {'cc_yolo_detect back to back' if depth == 0 else f'submit, {depth} in flight':28} median {ts[len(ts) // 2]:.3f} ms/step "
          f"= {B / ts[len(ts) // 2] * 1e3:.0f} frames/s  min {ts[0]:.3f}   bit-identical: {same}", flush=True)
