import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
B, H, W, Cin, Cout, k, stride = map(int, sys.argv[1:8]); reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
L = _lib.lib()
x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
Ho = (H + 2*(k//2) - k)//stride + 1; Wo = (W + 2*(k//2) - k)//stride + 1
out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
w = (np.random.default_rng(0).standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin*k*k)).astype(np.float32); b = np.zeros(Cout, np.float32)
torch.cuda.synchronize()
for _ in range(reps):
    _lib.check(L.cc_conv2d_nhwc(2, _lib.ptr(x), B, H, W, Cin, _lib.ptr(w), _lib.ptr(b), Cout, k, stride, 1, 1, _lib.ptr(out), 0, None))
print("done", float(out.float().abs().mean()))
