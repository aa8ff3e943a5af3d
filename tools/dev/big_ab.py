"""dev: the 128-channel layers of the detector under the single-barrier kernels - 128 x 128 on four waves, two blocks per CU (variant 6, the plan's choice) against
256 x 128 on eight waves (variant 13) - and the generic kernel (2).   python tools/dev/big_ab.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
shapes = [(64, 80, 128, 128, 3, 1), (64, 40, 128, 128, 3, 1), (64, 20, 128, 128, 3, 1), (64, 160, 128, 128, 3, 2), (64, 80, 128, 128, 3, 2), (64, 320, 64, 128, 3, 2), (64, 40, 512, 256, 3, 1), (64, 20, 256, 256, 3, 1)]
names = {0: "auto", 6: "128x128/4w", 13: "256x128/8w", 2: "generic"}
for (B, H, Cin, Cout, k, st) in shapes:
    best = {v: float("inf") for v in names}
    for _ in range(3):
        for v in names:
            t = C.c_float()
            if L.cc_conv_bench(1, B, H, H, Cin, Cout, k, st, 1, v, 20, C.byref(t)) == 0: best[v] = min(best[v], t.value * 1e3)
    gf = 2.0 * B * (H // st) ** 2 * Cout * Cin * k * k / 1e9
    print(f"B {B} {H}x{H} {Cin}->{Cout} k{k} s{st}: " + "  ".join(f"{names[v]} {best[v]:7.1f} us ({gf / best[v] * 1e3:4.0f} TF)" for v in names), flush=True)
