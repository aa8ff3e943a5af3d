# dev tool: CLIP ViT-L/14 image tower rate at B=255 (bf16), for A/B runs under kernel-selection env switches
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14
from clearcam_amd.objects import OpenCLIP
from clearcam_amd.weights import synthetic_clip_state_dict
m = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_L14, 4321), arch=CLIP_L14, dtype="bf16", device=0)
dev = torch.device("cuda", 0)
for B in (255, 64):
    x = torch.rand(B, 3, 224, 224, device=dev) * 2 - 1
    emb = torch.empty(B, 768, device=dev)
    for _ in range(2):
        m.precompute_embedding_device(x, emb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        m.precompute_embedding_device(x, emb)
    torch.cuda.synchronize()
    r = B / ((time.perf_counter() - t0) / 4)
    print(f"clip L/14 B={B}: {r:.1f} img/s = {r * 162.03e9 / 1e12:.1f} TF", flush=True)
