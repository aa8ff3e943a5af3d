# dev tool: CLIP ViT-L/14 image tower, B=255 bf16, a few batches (run under rocprofv3 --kernel-trace --stats)
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14
from clearcam_amd.objects import OpenCLIP
from clearcam_amd.weights import synthetic_clip_state_dict
m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_L14, 4321), arch=CLIP_L14, dtype="bf16")
x = torch.rand(255, 3, 224, 224, device="cuda") * 2 - 1
for _ in range(2): m.precompute_embedding(x).numpy()
t = time.time()
for _ in range(5): m.precompute_embedding(x).numpy()
print("images/s", 5 * 255 / (time.time() - t))
