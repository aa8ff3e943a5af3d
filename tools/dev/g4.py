import numpy as np, time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14
from clearcam_amd.weights import synthetic_clip_state_dict
from clearcam_amd.objects import OpenCLIP
sd = synthetic_clip_state_dict(CLIP_L14, 4321)
m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype="bf16")
B = 64
x = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
out = torch.empty(B, 768, device="cuda")
for _ in range(4): m.precompute_embedding_device(x, out)
torch.cuda.synchronize()
