import time, torch
n, sz = 64, 1080*1920*3
host = torch.empty((n, sz), dtype=torch.uint8).pin_memory()
dev = torch.empty((n, sz), dtype=torch.uint8, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return n * sz * reps / (time.perf_counter() - t0) / 1e9
s1, s2, s3, s4 = [torch.cuda.Stream() for _ in range(4)]
def per_cam(streams):
    def f():
        for i in range(n):
            with torch.cuda.stream(streams[i % len(streams)]):
                dev[i].copy_(host[i], non_blocking=True)
    return f
def one():
    with torch.cuda.stream(s1):
        dev.copy_(host, non_blocking=True)
def halves():
    with torch.cuda.stream(s1): dev[:32].copy_(host[:32], non_blocking=True)
    with torch.cuda.stream(s2): dev[32:].copy_(host[32:], non_blocking=True)
for name, fn in (("64 copies, 1 stream", per_cam([s1])), ("64 copies, 2 streams", per_cam([s1, s2])), ("64 copies, 4 streams", per_cam([s1, s2, s3, s4])),
                 ("1 copy of 398 MB", one), ("2 copies of 199 MB, 2 streams", halves)):
    print(f"{name:32} {t(fn):.1f} GB/s", flush=True)
