# dev tool: latency of 1-query and 64-query searches over a 125k x 768 shard
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd.objects import EmbeddingIndex
N = 125_000
ix = EmbeddingIndex(768, N)
e = torch.randn(N, 768, device="cuda"); e /= e.norm(dim=1, keepdim=True); ix.add(e)
for Q in (1, 4, 8, 64, 256):
    q = torch.randn(Q, 768); q /= q.norm(dim=1, keepdim=True); qn = q.numpy()
    for _ in range(3): ix.search(qn, 100)
    lat = []
    for _ in range(20):
        t0 = time.perf_counter(); ix.search(qn, 100); lat.append(time.perf_counter() - t0)
    lat.sort(); print(f"Q={Q}: p50 {lat[10]*1e3:.3f} ms  ({lat[10]*1e3/Q:.4f} ms per query)")
