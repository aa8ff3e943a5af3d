"""Development: the DDetect entry convs (box 64 + class 256 channels fused into one 320-channel launch) against a 256 + 64 split, and the
64-channel part under every kernel that can run it.  usage: python tools/dev/head_split.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
DT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name, B, H, W, Cin, Cout, k, stride, v in [("3x3 256->320 @80", 64, 80, 80, 256, 320, 3, 1, 0), ("3x3 256->256 @80", 64, 80, 80, 256, 256, 3, 1, 0),
                                               ("3x3 256->64 @80 auto", 64, 80, 80, 256, 64, 3, 1, 0), ("3x3 256->64 @80 generic", 64, 80, 80, 256, 64, 3, 1, 2),
                                               ("3x3 256->64 @80 halo", 64, 80, 80, 256, 64, 3, 1, 3), ("3x3 256->64 @80 few-tile", 64, 80, 80, 256, 64, 3, 1, 9),
                                               ("3x3 512->320 @40", 64, 40, 40, 512, 320, 3, 1, 0), ("3x3 512->256 @40", 64, 40, 40, 512, 256, 3, 1, 0),
                                               ("3x3 512->64 @40 auto", 64, 40, 40, 512, 64, 3, 1, 0), ("3x3 512->64 @40 generic", 64, 40, 40, 512, 64, 3, 1, 2),
                                               ("3x3 512->64 @40 halo", 64, 40, 40, 512, 64, 3, 1, 3), ("3x3 512->64 @40 few-tile", 64, 40, 40, 512, 64, 3, 1, 9),
                                               ("3x3 512->320 @20", 64, 20, 20, 512, 320, 3, 1, 0), ("3x3 512->256 @20", 64, 20, 20, 512, 256, 3, 1, 0),
                                               ("3x3 512->64 @20 auto", 64, 20, 20, 512, 64, 3, 1, 0), ("3x3 512->64 @20 generic", 64, 20, 20, 512, 64, 3, 1, 2)]:
    try:
        for r in range(2):
            ms = C.c_float(); _lib.check(L.cc_conv_bench(DT, B, H, W, Cin, Cout, k, stride, 1, v, 20, C.byref(ms)))
        print(name, round(ms.value * 1e3, 1), "us", flush=True)
    except Exception as e:
        print(name, "error", e, flush=True)
