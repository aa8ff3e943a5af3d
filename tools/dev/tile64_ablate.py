"""dev (build with HIPCC_EXTRA=-DCC_TILE64_ABLATIONS): timing ablations of the two-group 3x3 64 -> 64 tile kernel at B = 64, 160 x 160.
python tools/dev/tile64_ablate.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
names = {0: "as built", 256: "K prio 1", 1280: "F prio 1", 64: "reads 2 ahead", 2: "no stores", 1: "no patch DMA", 4: "no act", 32: "K loop only", 16: "finish only"}
for (B, H, W) in [(64, 160, 160)]:
    best = {v: float("inf") for v in names}
    for _ in range(3):
        for v in names:
            L.cc_dev_set(b"tile64_abl", v)
            t = C.c_float()
            rc = L.cc_conv_bench(1, B, H, W, 64, 64, 3, 1, 1, 12, 20, C.byref(t))
            if rc == 0:
                best[v] = min(best[v], t.value * 1e3)
    L.cc_dev_set(b"tile64_abl", 0)
    for v, nme in names.items():
        print(f"B {B} {H}x{W}  abl {v:3d} {nme:28s} {best[v]:8.1f} us", flush=True)
