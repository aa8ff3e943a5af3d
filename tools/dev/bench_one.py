"""dev: one conv shape under one forced variant, a few launches (the workload of a PMC pass).
python tools/dev/bench_one.py <dt 1=f16 2=bf16 3=f16s> B H W Cin Cout k stride variant [iters]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clearcam_amd import _lib
L = _lib.lib()
a = [int(v) for v in sys.argv[1:]]
dt, B, H, W, Cin, Cout, k, st, var = a[:9]
iters = a[9] if len(a) > 9 else 5
t = C.c_float()
rc = L.cc_conv_bench(dt, B, H, W, Cin, Cout, k, st, 1, var, iters, C.byref(t))
print(f"rc {rc}  {t.value * 1e3:.1f} us per launch")
