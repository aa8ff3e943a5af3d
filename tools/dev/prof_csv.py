# dev tool: dump the per-launch profile CSV of the bench plan (env CLEARCAM_PROFILE_CSV=<path>); argv: [batch] [dtype]
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype=dtype)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(B, 300, 6, device="cuda")
for _ in range(3): m.detect_batch_device(f, o)
torch.cuda.synchronize()
print(m.profile(iters=5))
