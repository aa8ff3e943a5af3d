import sys, os, numpy as np, torch, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
if os.environ.get("CC_LIB"): _lib.LIB_PATH = os.environ["CC_LIB"]
B, H, W, Cin, Cout, k, stride = map(int, sys.argv[1:8])
L = _lib.lib()
x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
Ho = (H + 2*(k//2) - k)//stride + 1; Wo = (W + 2*(k//2) - k)//stride + 1
out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
w = (np.random.default_rng(0).standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin*k*k)).astype(np.float32); b = np.zeros(Cout, np.float32)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(L.cc_conv2d_nhwc(2, _lib.ptr(x), B, H, W, Cin, _lib.ptr(w), _lib.ptr(b), Cout, k, stride, 1, 1, _lib.ptr(out), 0, None))
    ts.append(time.perf_counter() - t)
