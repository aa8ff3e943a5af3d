# dev tool: library GEMM (hipBLASLt via torch.matmul) vs conv_mfma 1x1 on the same shapes; read durations from rocprofv3 --kernel-trace --stats
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clearcam_amd import _lib
L = _lib.lib()
shapes = [(65536, 1024, 1024), (65536, 4096, 1024), (65536, 1024, 4096), (131072, 512, 2048)]
if os.environ.get("SHAPES"): shapes = [tuple(map(int, t.split("x"))) for t in os.environ["SHAPES"].split(",")]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch.matmul M{M} N{N} K{K}: {ms*1e3:.0f} us {2*M*N*K/ms/1e9:.0f} TF", flush=True)
    if K <= 4096 and M % 64 == 0:
        x = a.view(1, M // 64, 64, K); out = torch.empty(1, M // 64, 64, N, device="cuda", dtype=torch.bfloat16)
        wf = w.float().cpu().numpy().reshape(N, K, 1, 1).copy(); b = np.zeros(N, np.float32)
        for _ in range(3):
            _lib.check(L.cc_conv2d_nhwc(2, _lib.ptr(x), 1, M // 64, 64, K, _lib.ptr(wf), _lib.ptr(b), N, 1, 1, 1, 0, _lib.ptr(out), 0, None))
        torch.cuda.synchronize()
