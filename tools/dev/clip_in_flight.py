"""dev tool: CLIP ViT-L/14 image batches through one handle against D handles taking batches in turn on their own streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14  # noqa: E402
from clearcam_amd.objects import OpenCLIP  # noqa: E402
from clearcam_amd.weights import synthetic_clip_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 255
D = 4
dev = torch.device("cuda", 0)
sd = synthetic_clip_state_dict(CLIP_L14, 4321)
x = torch.rand(B, 3, 224, 224, device=dev) * 2 - 1
ms = [OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype="bf16", device=0) for _ in range(D)]
embs = [torch.empty(B, 768, device=dev) for _ in range(D)]
side = [torch.cuda.Stream() for _ in range(D)]
for m, e in zip(ms, embs):
    m.precompute_embedding_device(x, e)
torch.cuda.synchronize()
ref = embs[0].clone()
N = 12
for rnd in range(2):
    for depth in (1, 2, 3, 4):
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(N):
            j = k % depth
            with torch.cuda.stream(side[j]):
                ms[j].precompute_embedding_device(x, embs[j])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / N
        same = all(bool(torch.equal(embs[j], ref)) for j in range(depth))
        print(f"[{rnd}] CLIP L/14 bf16 B={B} {depth} in flight: {dt * 1e3:.2f} ms/batch = {B / dt:.0f} img/s   identical: {same}", flush=True)
