"""dev: balanced rounds (conv_persist.hip) on / off on the detector's eight-wave shapes at B = 64 (cc_conv_bench, variant 0 = the plan's own choice).
python tools/dev/balance_ab.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
# (dtype code 1 = f16 / 3 = f16s two planes, B, H, Cin, Cout, k, stride, note)
shapes = [(1, 64, 40, 256, 256, 3, 1, "3x3 256->256 @40 (400 tiles)"), (1, 64, 40, 512, 256, 3, 1, "3x3 512->256 @40"), (1, 64, 80, 256, 256, 3, 2, "3x3 s2 256->256 -> 40"),
          (1, 64, 40, 1024, 512, 1, 1, "1x1 1024->512 @40 (800 tiles)"), (3, 64, 40, 512, 512, 1, 1, "1x1 512->512 @40 two planes"), (1, 64, 40, 768, 512, 1, 1, "1x1 768->512 @40"),
          (1, 64, 80, 256, 256, 3, 1, "3x3 256->256 @80 (1600 tiles)"), (1, 64, 80, 1024, 256, 1, 1, "1x1 1024->256 @80"), (3, 64, 80, 512, 512, 1, 1, "1x1 512->512 @80 two planes (3200 tiles)"),
          (1, 64, 20, 1024, 512, 1, 1, "1x1 1024->512 @20 (200 tiles)"), (1, 64, 20, 512, 512, 3, 1, "3x3 512->512 @20")]
tot = [0.0, 0.0]
for (dt, B, H, Cin, Cout, k, st, note) in shapes:
    best = [float("inf"), float("inf")]
    for _ in range(3):
        for on in (0, 1):
            L.cc_dev_set(b"balance", on)
            t = C.c_float()
            if L.cc_conv_bench(dt, B, H, H, Cin, Cout, k, st, 1, 0, 20, C.byref(t)) == 0: best[on] = min(best[on], t.value * 1e3)
    L.cc_dev_set(b"balance", -1)
    tot[0] += best[0]; tot[1] += best[1]
    print(f"{note:44s} plain {best[0]:7.1f} us   balanced {best[1]:7.1f} us   {best[1] / best[0]:.3f}", flush=True)
print(f"sum plain {tot[0]:.0f} balanced {tot[1]:.0f}")
