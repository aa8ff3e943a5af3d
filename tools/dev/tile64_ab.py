"""dev: the persistent 3x3 64 -> 64 tile kernel (variant 12) against the wave-autonomous (8), weights-stationary (4) and generic (2) kernels
on the detector's shapes (cc_conv_bench, B = 64 and small batches).   python tools/dev/tile64_ab.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
names = {12: "tile64", 8: "wave", 4: "ws", 2: "generic", 0: "auto"}
print("shape (B H W) f16 | " + " ".join(f"{n:>8}" for n in names.values()) + "   (us per launch, 20 iterations, 3 rounds interleaved: min)")
for (B, H, W) in [(64, 160, 160), (64, 80, 80), (64, 40, 40), (16, 160, 160), (16, 80, 80)]:
    best = {v: float("inf") for v in names}
    for _ in range(3):
        for v in names:
            t = C.c_float()
            rc = L.cc_conv_bench(1, B, H, W, 64, 64, 3, 1, 1, v, 20, C.byref(t))
            if rc == 0:
                best[v] = min(best[v], t.value * 1e3)
    gf = 2.0 * B * H * W * 64 * 576 / 1e9
    print(f"{B:3d} {H:4d} {W:4d} | " + " ".join(f"{best[v]:8.1f}" for v in names) + f"   tile64 {gf / best[12] * 1e3:.0f} TFLOP/s", flush=True)
