import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
m = YOLOv9("c", 640, state_dict=synthetic_yolov9_state_dict("c", 1234), dtype="bf16")
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(64, 300, 6, device="cuda")
for _ in range(3): m.detect_batch_device(f, o)
torch.cuda.synchronize()
for w in (0, 1, 2):
    try: print(w, m.profile_graph(w, 10))
    except Exception as e: print(w, "ERR", e)
