"""dev: a few conv shapes of the bench plan under the generic kernel's tile / stage knobs (CLEARCAM_CONV_CFG is read once per process,
so run one process per configuration):   CLEARCAM_CONV_CFG=256,2,128,2 python tools/dev/cfg_sweep.py [dtype code]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
DT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
out = []
for name, B, H, W, Cin, Cout, k, stride, v in [("3x3s2 64->128 @320", 64, 320, 320, 64, 128, 3, 2, 2), ("1x1 256->256 @160", 64, 160, 160, 256, 256, 1, 1, 2),
                                               ("1x1 128->128 @160", 64, 160, 160, 128, 128, 1, 1, 2), ("3x3 64->64 @160", 64, 160, 160, 64, 64, 3, 1, 2),
                                               ("3x3 128->128 @80", 64, 80, 80, 128, 128, 3, 1, 2), ("3x3s2 128->128 @160", 64, 159, 159, 128, 128, 3, 2, 2),
                                               ("1x1 64->64 @160", 64, 160, 160, 64, 64, 1, 1, 2), ("3x3 256->64 @80", 64, 80, 80, 256, 64, 3, 1, 2)]:
    ms = C.c_float()
    for r in range(2):
        rc = L.cc_conv_bench(DT, B, H, W, Cin, Cout, k, stride, 1, v, 20, C.byref(ms))
    out.append(f"{name} {ms.value * 1e3:.1f}" if rc == 0 else f"{name} err")
print(f"dtype {DT} cfg {os.environ.get('CLEARCAM_CONV_CFG', 'default'):14s} thin_k {os.environ.get('CLEARCAM_THIN_K', '-'):5s} | " + " | ".join(out), flush=True)
