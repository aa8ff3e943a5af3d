import numpy as np, time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_L14
from clearcam_amd.weights import synthetic_clip_state_dict
from clearcam_amd.objects import OpenCLIP
sd = synthetic_clip_state_dict(CLIP_L14, 4321)
for dt in ("bf16",):
    m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype=dt)
    for B in (16, 64, 128, 256):
        x = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
        out = torch.empty(B, 768, device="cuda")
        for _ in range(2): m.precompute_embedding_device(x, out)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 5
        for _ in range(n): m.precompute_embedding_device(x, out)
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t) / n
        print(dt, "B", B, "ms", round(dt_ * 1e3, 2), "img/s", round(B / dt_), "TFLOP/s", round(B * 162.03e9 / dt_ / 1e12, 1), "gpu_ms", round(m.last_gpu_ms(), 2))
