"""dev tool: host-to-device rate of one pinned 398 MB copy (64 x 1080p frames) by stream - is the rate a property of the stream (the
DMA engine its queue is bound to)?  And of 64 per-camera copies against one copy of the whole tick."""
import time

import torch

n, sz = 64, 1080 * 1920 * 3
host = torch.empty((n, sz), dtype=torch.uint8).pin_memory()
dev = torch.empty((n, sz), dtype=torch.uint8, device="cuda")


def rate(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n * sz * reps / (time.perf_counter() - t0) / 1e9


streams = [torch.cuda.Stream(priority=-1 if i % 2 == 0 else 0) for i in range(10)]
for rnd in range(2):
    out = []
    for i, st in enumerate(streams):
        def one(st=st):
            with torch.cuda.stream(st):
                dev.copy_(host, non_blocking=True)
        out.append(f"{rate(one):.1f}")
    print(f"round {rnd}: one 398 MB copy, streams 0..9 (even = high priority): " + " ".join(out) + " GB/s", flush=True)


def per_cam(sts):
    def f():
        for i in range(n):
            with torch.cuda.stream(sts[i % len(sts)]):
                dev[i].copy_(host[i], non_blocking=True)
    return f


for name, fn in (("64 copies, 1 stream", per_cam(streams[:1])), ("64 copies, 2 streams", per_cam(streams[:2]))):
    print(f"{name:32} {rate(fn):.1f} GB/s", flush=True)
