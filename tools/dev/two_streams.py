# dev tool: one batch of 64 on one handle vs two handles (own streams, own hipGraphs) with 32 frames each, launched back to back
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.weights import synthetic_yolov9_state_dict
from clearcam_amd.yolov9 import YOLOv9
sd = synthetic_yolov9_state_dict("c", 1234)
f = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)).cuda()
o = torch.empty(64, 300, 6, device="cuda")
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
m = YOLOv9("c", 640, state_dict=sd, dtype="bf16")
print("1 x 64: %.3f ms" % timeit(lambda: m.detect_batch_device(f, o)))
for parts in [int(a) for a in (sys.argv[1:] or [2, 4, 8, 16, 4, 2])]:
    ms = [YOLOv9("c", 640, state_dict=sd, dtype="bf16") for _ in range(parts)]
    n = 64 // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def run():
        for i, (mm, s) in enumerate(zip(ms, streams)):
            with torch.cuda.stream(s):
                mm.detect_batch_device(f[i * n:(i + 1) * n], o[i * n:(i + 1) * n])
    print("%d x %d concurrent: %.3f ms" % (parts, n, timeit(run)))
    for mm in ms: mm.close()
