"""dev: a few conv shapes under the generic kernel's 64-byte-row (BK=32, five blocks per CU) configuration forced for long K
(CLEARCAM_THIN_K, CLEARCAM_MID_SCHED=0) against the default selection.  Run twice: with and without the env."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
shapes = [(64, 40, 128, 128, 3, 1), (64, 40, 256, 256, 3, 1), (64, 40, 256, 256, 1, 1), (64, 20, 128, 128, 3, 1), (64, 20, 256, 256, 3, 1), (64, 20, 256, 256, 1, 1),
          (64, 40, 1024, 512, 1, 1), (64, 80, 128, 128, 3, 1), (64, 80, 128, 128, 1, 1), (64, 160, 64, 64, 3, 1)]
for B, H, Cin, Cout, k, st in shapes:
    out = []
    for v in (0, 2):
        t = C.c_float()
        rc = L.cc_conv_bench(2, B, H, H, Cin, Cout, k, st, 1, v, 30, C.byref(t))
        out.append(t.value * 1e3 if rc == 0 else float("nan"))
    print(f"B{B} {H}x{H} {Cin}->{Cout} k{k}: auto {out[0]:7.1f} us  generic(variant 2) {out[1]:7.1f} us   [THIN_K={os.environ.get('CLEARCAM_THIN_K')} MID={os.environ.get('CLEARCAM_MID_SCHED')}]", flush=True)
