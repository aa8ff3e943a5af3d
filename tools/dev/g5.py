# dev tool: crop preprocessing throughput and CLIP batch-size sweep around the tile-quantisation points
import numpy as np, time, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.arch import CLIP_B32, CLIP_L14
from clearcam_amd.weights import synthetic_clip_state_dict
from clearcam_amd.objects import OpenCLIP, preprocess_crops
rng = np.random.default_rng(0)
crops = [rng.integers(0, 256, (int(rng.integers(48, 400)), int(rng.integers(32, 300)), 3), dtype=np.uint8) for _ in range(1024)]
preprocess_crops(crops[:8])
t = time.perf_counter(); x = preprocess_crops(crops); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("crop preprocess (host pack + H2D + kernel): %d crops %.1f ms -> %.0f crops/s; %.1f MB packed" % (len(crops), dt * 1e3, len(crops) / dt, sum(c.size for c in crops) / 1e6))
sd = synthetic_clip_state_dict(CLIP_L14, 4321)
m = OpenCLIP(state_dict=sd, arch=CLIP_L14, dtype="bf16")
for B in [int(a) for a in (sys.argv[1:] or [248, 255, 256, 384, 510, 512])]:
    x = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
    out = torch.empty(B, 768, device="cuda")
    for _ in range(2): m.precompute_embedding_device(x, out)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 4
    for _ in range(n): m.precompute_embedding_device(x, out)
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t) / n
    print("B", B, "ms", round(dt_ * 1e3, 2), "img/s", round(B / dt_), "TFLOP/s", round(B * 162.03e9 / dt_ / 1e12, 1))

if os.environ.get("B32"):
    m.close()
    mb = OpenCLIP(state_dict=synthetic_clip_state_dict(CLIP_B32, 99), arch=CLIP_B32, dtype="bf16")
    for B in (512, 1024):
        x = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
        out = torch.empty(B, 512, device="cuda")
        for _ in range(2): mb.precompute_embedding_device(x, out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(4): mb.precompute_embedding_device(x, out)
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t) / 4
        print("ViT-B/32 B", B, "ms", round(dt_ * 1e3, 2), "img/s", round(B / dt_))
