"""dev: the attention kernel alone (cc_attn_bench): whole kernel / staging only / tiles only, ViT-L/14 shape (B=255, L=257, H=16) and others"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
for (B, Lq, H, causal) in ((255, 257, 16, 0), (64, 257, 16, 0), (1024, 50, 12, 0), (64, 77, 12, 1)):
    out = []
    ms = C.c_float(); L.cc_attn_bench(2, B, Lq, H, causal, 64, 1, C.byref(ms))
    for abl in (0, 1, 2, 2 + 32, 128, 128 + 2):
        ms = C.c_float()
        _lib.check(L.cc_attn_bench(2, B, Lq, H, causal, abl, 20, C.byref(ms)))
        out.append(ms.value * 1e3)
    gf = 4.0 * Lq * Lq * 64 * H * B / 1e9
    gb = B * Lq * H * 64 * 2 * 4 / 1e9
    print(f"B={B} L={Lq} H={H} causal={causal}: full {out[0]:.1f} us ({gf / out[0] * 1e3:.0f} TF, {gb / out[0] * 1e3:.2f} TB/s)  staging only {out[1]:.1f} us  tiles only {out[2]:.1f} us  empty blocks {out[3]:.1f} us | one tile per wave round: full {out[4]:.1f}  tiles only {out[5]:.1f}", flush=True)
