"""dev: one conv shape under the environment's tile configuration (CLEARCAM_CONV_CFG, CLEARCAM_MID_SCHED ...), variant 0 and a forced one.
python tools/dev/cfg_ab.py B H Cin Cout k stride [variant]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
B, H, Cin, Cout, k, st = [int(v) for v in sys.argv[1:7]]
var = int(sys.argv[7]) if len(sys.argv) > 7 else 0
best = float("inf")
for _ in range(3):
    t = C.c_float()
    if L.cc_conv_bench(1, B, H, H, Cin, Cout, k, st, 1, var, 20, C.byref(t)) == 0: best = min(best, t.value * 1e3)
gf = 2.0 * B * (H // st) ** 2 * Cout * Cin * k * k / 1e9
env = {kk: v for kk, v in os.environ.items() if kk.startswith("CLEARCAM_")}
print(f"B {B} {H}x{H} {Cin}->{Cout} k{k} s{st} variant {var} env {env}: {best:.1f} us  {gf / best * 1e3:.0f} TFLOP/s", flush=True)
