import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from clearcam_amd import _lib
L = _lib.lib()
for name, B, H, W, Cin, Cout, k, stride, v in [("3x3 256->320 @80", 64, 80, 80, 256, 320, 3, 1, 0), ("3x3 256->256 @80", 64, 80, 80, 256, 256, 3, 1, 0), ("3x3 256->64 @80", 64, 80, 80, 256, 64, 3, 1, 0),
                                               ("3x3 512->320 @40", 64, 40, 40, 512, 320, 3, 1, 0), ("3x3 512->256 @40", 64, 40, 40, 512, 256, 3, 1, 0), ("3x3 512->64 @40", 64, 40, 40, 512, 64, 3, 1, 0)]:
    for r in range(2):
        ms = C.c_float(); L.cc_conv_bench(2, B, H, W, Cin, Cout, k, stride, 1, v, 20, C.byref(ms))
    print(name, round(ms.value * 1e3, 1), "us")
