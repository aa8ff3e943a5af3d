"""dev tool: a handle created right after a handle whose slots carried pinned-host submissions was torn down replays its graph
3x slower (5.6 instead of 1.8 ms at B = 8) - which ingredient is it?   python tools/dev/slow_graph.py [host|dev] [pipe]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict  # noqa: E402
from clearcam_amd.yolov9 import YOLOv9  # noqa: E402

host = len(sys.argv) > 1 and sys.argv[1] == "host"
sd = synthetic_yolov9_state_dict("c", 1234)
B = 8
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(B, 300, 6, device="cuda")
fh = f.cpu().pin_memory()
oh = [torch.empty(B, 300, 6).pin_memory() for _ in range(4)]
side = torch.cuda.Stream()


def plain(tag):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    with torch.cuda.stream(side):
        for _ in range(10):
            m.detect_batch_device(f, o)
        torch.cuda.synchronize()
    print(f"{tag}: plain handle, detect on the GPU {m.last_gpu_ms():.2f} ms", flush=True)
    m.close()


plain("first")
for rnd in range(3):
    m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
    m.set_in_flight(3)
    ts = []
    for k in range(12):
        if len(ts) == 3:
            m.wait(ts.pop(0), host=True)
        ts.append(m.submit(fh if host else f, oh[k % 3] if host else o))
    for t in ts:
        m.wait(t, host=True)
    torch.cuda.synchronize()
    m.close()
    plain(f"after a 3-slot handle with {'pinned host' if host else 'device'} tensors (round {rnd})")
    plain("the one after")
