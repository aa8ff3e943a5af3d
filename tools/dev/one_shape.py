# dev tool: run ONE cc_conv_bench shape (for rocprofv3 counter passes): B H W Cin Cout k stride variant [iters]
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
B, H, W, Cin, Cout, k, stride, v = map(int, sys.argv[1:9])
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
ms = C.c_float()
rc = L.cc_conv_bench(2, B, H, W, Cin, Cout, k, stride, 1, v, iters, C.byref(ms))
print(rc, ms.value * 1e3, "us", 2.0 * B * ((H + 2 * (k // 2) - k) // stride + 1) * ((W + 2 * (k // 2) - k) // stride + 1) * Cout * Cin * k * k / 1e9 / ms.value, "TF")
