# dev tool: time the 256x256 kernel on one GEMM shape (variant 5) with the library named by CC_LIB
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
if os.environ.get("CC_LIB"): _lib.LIB_PATH = os.environ["CC_LIB"]
L = _lib.lib()
M, N, K = 65536, 4096, 1024
x = torch.randn(1, M // 64, 64, K, device="cuda").to(torch.bfloat16); out = torch.empty(1, M // 64, 64, N, device="cuda", dtype=torch.bfloat16)
w = (np.random.default_rng(0).standard_normal((N, K, 1, 1)) / 32).astype(np.float32); b = np.zeros(N, np.float32)
for _ in range(4):
    _lib.check(L.cc_conv2d_nhwc(2, _lib.ptr(x), 1, M // 64, 64, K, _lib.ptr(w), _lib.ptr(b), N, 1, 1, 1, 0, _lib.ptr(out), 5, None))
torch.cuda.synchronize()
