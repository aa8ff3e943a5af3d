"""dev: 256 x 256 tiles on SIXTEEN waves of 64 x 64 (generic two-barrier loop, variant 14) against the eight-wave kernel (7) and the plan's choice (0).
python tools/dev/wave16_ab.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
shapes = [(1, 64, 80, 256, 256, 3, 1), (1, 64, 40, 256, 256, 3, 1), (1, 64, 40, 1024, 512, 1, 1), (1, 64, 80, 1024, 256, 1, 1), (1, 255, 16, 1024, 4096, 1, 1), (1, 255, 16, 4096, 1024, 1, 1)]
names = {7: "8-wave", 14: "16w 64KBx2", 15: "16w 32KBx4"}
for (dt, B, H, Cin, Cout, k, st) in shapes:
    best = {v: float("inf") for v in names}
    for _ in range(3):
        for v in names:
            t = C.c_float()
            if L.cc_conv_bench(dt, B, H, H, Cin, Cout, k, st, 1, v, 10, C.byref(t)) == 0: best[v] = min(best[v], t.value * 1e3)
    gf = 2.0 * B * (H // st) ** 2 * Cout * Cin * k * k / 1e9
    print(f"B {B} {H}x{H} {Cin}->{Cout} k{k}: " + "  ".join(f"{names[v]} {best[v]:7.1f} us ({gf / best[v] * 1e3:4.0f} TF)" for v in names), flush=True)
