# dev tool (GPU): single-frame latency of the detector (the reference's batch-1 call) under an environment switch, e.g.
#   CLEARCAM_FUSE_ADOWN=1 python tools/dev/b1_latency.py [batch] [dtype]
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16h"
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype=dtype)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(B, 300, 6, device="cuda")
for _ in range(20): m.detect_batch_device(f, o)
torch.cuda.synchronize()
lat = []
for _ in range(200):
    torch.cuda.synchronize(); t = time.perf_counter(); m.detect_batch_device(f, o); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
lat.sort()
p = m.profile(iters=3)
print(f"B={B} {dtype} env { {k: v for k, v in os.environ.items() if k.startswith('CLEARCAM_') and k != 'CLEARCAM_PROFILE_CSV'} }: p50 {lat[100] * 1e3:.3f} ms  p10 {lat[20] * 1e3:.3f}  gpu {m.last_gpu_ms():.3f} ms  launches conv {p['conv_launches']}")
