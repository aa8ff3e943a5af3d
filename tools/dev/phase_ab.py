# dev tool: A/B of conv kernel variants on GEMM / conv shapes through cc_conv_bench (random data, device time per launch)
#   python tools/dev/phase_ab.py [variants, e.g. 0,5,7]
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib  # noqa: E402

L = _lib.lib()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,5,7").split(",")]
# (name, B, H, W, Cin, Cout, k, stride)
SHAPES = [
    ("gemm 65536x256x2304 (1x1)", 64, 32, 32, 2304, 256, 1, 1),
    ("gemm 65535x1024x1024 (CLIP out-proj)", 255, 257, 1, 1024, 1024, 1, 1),
    ("gemm 65535x4096x1024 (CLIP fc)", 255, 257, 1, 1024, 4096, 1, 1),
    ("gemm 65535x1024x4096 (CLIP proj)", 255, 257, 1, 4096, 1024, 1, 1),
    ("3x3 256->256 @80x80 B64 (head)", 64, 80, 80, 256, 256, 3, 1),
    ("3x3 256->256 @40x40 B64", 64, 40, 40, 256, 256, 3, 1),
    ("3x3 512->256 @40x40 B64", 64, 40, 40, 512, 256, 3, 1),
    ("1x1 1024->512 @40x40 B64", 64, 40, 40, 1024, 512, 1, 1),
    ("1x1 1024->256 @80x80 B64", 64, 80, 80, 1024, 256, 1, 1),
    ("1x1 512->512 @80x80 B64", 64, 80, 80, 512, 512, 1, 1),
    ("1x1 256->256 @160x160 B64", 64, 160, 160, 256, 256, 1, 1),
]
for rnd in range(2):
    for name, B, H, W, Cin, Cout, k, stride in SHAPES:
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        gf = 2.0 * B * Ho * Wo * Cout * Cin * k * k / 1e9
        row = []
        for v in variants:
            ms = C.c_float()
            rc = L.cc_conv_bench(2, B, H, W, Cin, Cout, k, stride, 1, v, 20, C.byref(ms))
            row.append(f"v{v}: {ms.value * 1e3:8.1f} us {gf / ms.value:7.0f} TF" if rc == 0 else f"v{v}: error {L.cc_last_error().decode()[:60]}")
        print(f"[{rnd}] {name:40} {gf:8.1f} GF  " + "   ".join(row), flush=True)
