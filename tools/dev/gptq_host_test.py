# dev test (CPU): tools/dev/gptq_host.cpp against the Python recursion of tools/dev/gptq_gpu.py on random layers
import ctypes as C, os, subprocess, sys, tempfile, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gptq_gpu import gptq_f16
so = os.path.join(tempfile.gettempdir(), "gptq_host.so")
subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "gptq_host.cpp")])
L = C.CDLL(so)
L.gptq_round_f16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
rng = np.random.default_rng(0)
for co, ci, rows in ((64, 128, 4000), (256, 512, 20000), (96, 1024, 6000)):
    base = rng.standard_normal((rows, 32)) @ rng.standard_normal((32, ci)) + 0.3 * rng.standard_normal((rows, ci)) + 0.5      # correlated channels, non-zero mean
    X = np.maximum(base, 0) * 0.2
    H = X.T @ X / rows
    w = (rng.standard_normal((co, ci)) / np.sqrt(ci)).astype(np.float32)
    out = np.empty_like(w)
    rc = L.gptq_round_f16(w.ctypes.data, co, ci, np.ascontiguousarray(H).ctypes.data, 0.01, out.ctypes.data)
    ref = gptq_f16(torch.from_numpy(w), torch.from_numpy(H), 0.01).numpy()
    same = float((out == ref).mean())
    t = torch.from_numpy(out)
    proxy = lambda q: float(np.einsum("oi,ij,oj->", (q - w).astype(np.float64), H, (q - w).astype(np.float64)))      # noqa: E731
    near = w.astype(np.float16).astype(np.float32)
    print(f"co {co} ci {ci}: rc {rc}, identical to the Python recursion on {same:.4f} of the weights; E|dW x|^2: nearest {proxy(near):.3e}  C++ {proxy(out):.3e}  Python {proxy(ref):.3e}")
    assert rc == 0 and torch.equal(t.to(torch.float16).float(), t) and same >= 0.995 and abs(proxy(out) / proxy(ref) - 1) < 0.02 and proxy(out) < 0.5 * proxy(near)
print("ok")
