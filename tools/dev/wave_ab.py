# dev tool: narrow 3x3 layers - wave-autonomous kernel (variant 8) vs weights-stationary (4) vs generic (2) vs auto (0)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
SH = [("3x3 64->64 @80 B64", 64, 80, 80, 64, 64), ("3x3 64->64 @160 B64", 64, 160, 160, 64, 64), ("3x3 32->32 @160 B64", 64, 160, 160, 32, 32),
      ("3x3 64->64 @40 B64", 64, 40, 40, 64, 64), ("3x3 64->64 @80 B1", 1, 80, 80, 64, 64)]
for rnd in range(2):
    for name, B, H, W, Ci, Co in SH:
        row = []
        for v in (8, 4, 2):
            ms = C.c_float()
            rc = L.cc_conv_bench(2, B, H, W, Ci, Co, 3, 1, 1, v, 20, C.byref(ms))
            gf = 2.0 * B * H * W * Co * Ci * 9 / 1e9
            row.append(f"v{v}: {ms.value * 1e3:7.1f} us {gf / ms.value:6.0f} TF" if rc == 0 else f"v{v}: err")
        print(f"[{rnd}] {name:24}", "   ".join(row), flush=True)
