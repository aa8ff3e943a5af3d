# dev tool (GPU): the weights-resident streaming 1x1 kernel (conv_stream.hip, variant 10) against the kernel the selector took before it
# (variant 0 with cc_dev_set("stream", 0)) on the detector's thin 1x1 layers, B = 64, one and two weight planes (cc_conv_bench).
import ctypes as C, sys
from clearcam_amd import _lib
L = _lib.lib()
SHAPES = [(160, 64, 64), (160, 128, 128), (160, 256, 256), (80, 128, 128), (80, 256, 256), (80, 512, 256), (40, 128, 128), (40, 256, 256), (40, 512, 256),
          (20, 256, 256), (20, 512, 256)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print(f"B={B}: H Cin Cout planes | old ms (TB/s) | stream ms (TB/s)")
for H, ci, co in SHAPES:
    for dt, planes in ((1, 1), (3, 2)):
        if planes == 2 and ci * 2 > 512: continue
        gb = B * H * H * (ci + co) * 2 / 1e9
        ms = C.c_float()
        _lib.check(L.cc_dev_set(b"stream", 0))
        _lib.check(L.cc_conv_bench(dt, B, H, H, ci, co, 1, 1, 1, 0, 20, C.byref(ms))); old = ms.value
        _lib.check(L.cc_dev_set(b"stream", -1))
        _lib.check(L.cc_conv_bench(dt, B, H, H, ci, co, 1, 1, 1, 10, 20, C.byref(ms))); new = ms.value
        print(f"{H:4d} {ci:4d} {co:4d} {planes} | {old:.4f} ({gb / old:.2f}) | {new:.4f} ({gb / new:.2f}) | x{old / new:.2f}", flush=True)
