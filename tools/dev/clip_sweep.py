# dev tool: CLIP ViT-L/14 image rate over the reference's batch sweep (test/test_clip_speed.py:8-15) + 255
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14
from clearcam_amd.objects import OpenCLIP
from clearcam_amd.weights import synthetic_clip_state_dict
m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_L14, 4321), arch=CLIP_L14, dtype="bf16", device=0)
dev = torch.device("cuda", 0)
out = {}
for B in (8, 16, 32, 64, 128, 255):
    x = torch.rand(B, 3, 224, 224, device=dev) * 2 - 1
    emb = torch.empty(B, 768, device=dev)
    for _ in range(2):
        m.precompute_embedding_device(x, emb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        m.precompute_embedding_device(x, emb)
    torch.cuda.synchronize()
    out[B] = round(B / ((time.perf_counter() - t0) / 4), 1)
print("clip sweep", out, flush=True)
