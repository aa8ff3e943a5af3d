"""CPU tests: the C-ABI library loads and exports everything include/clearcam_hip.h declares; host shims."""
import ctypes
import os
import re

import numpy as np
import pytest

from clearcam_amd import _lib
from clearcam_amd.helpers import Tensor, as_numpy, jit_infer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "clearcam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported(lib_path):
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == declared                  # the Python binding covers the whole header
    L = ctypes.CDLL(lib_path)
    for s in declared:
        assert hasattr(L, s), f"{s} not exported by libclearcam_hip.so"


def test_no_cpu_fallback_without_gpu(lib_path, sd_t):
    """On a box without a GPU the product path must raise, not fall back (no compute happens here)."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from clearcam_amd.yolov9 import YOLOv9
    with pytest.raises(_lib.CCError):
        YOLOv9("t", 320, state_dict=sd_t)


def test_bad_arguments_report_errors(lib_path):
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.cc_yolo_create(ctypes.byref(h), b"x", 640, 2, 0) != 0
    assert b"unknown model size" in L.cc_last_error() or b"device" in L.cc_last_error().lower() or L.cc_last_error()
    assert L.cc_yolo_create(ctypes.byref(h), b"c", 641, 2, 0) != 0
    assert L.cc_yolo_create(ctypes.byref(h), b"c", 640, 7, 0) != 0


def test_missing_weights_message():
    from clearcam_amd.yolov9 import YOLOv9
    with pytest.raises(FileNotFoundError):
        YOLOv9("c", 640, weights="/nonexistent/yolov9-c.safetensors")
    with pytest.raises(FileNotFoundError):
        YOLOv9("e", 640, weights="/nonexistent/yolov9-e.safetensors")
    with pytest.raises(ValueError):
        YOLOv9("x", 640)


def test_tensor_shim_and_jit_infer():
    f = np.zeros((4, 5, 3), np.uint8)
    t = Tensor(f)
    assert t.shape == (4, 5, 3) and t.numpy() is f
    assert t.cast("float32").numpy().dtype == np.float32
    assert t.unsqueeze(0).shape == (1, 4, 5, 3)
    assert as_numpy(Tensor(Tensor(f))) is f
    calls, cache = [], {}
    fn = lambda x: calls.append(x.shape) or Tensor(np.zeros((300, 6), np.float32))   # noqa: E731
    out = jit_infer(fn, t, cache)
    jit_infer(fn, t, cache)
    jit_infer(fn, Tensor(np.zeros((8, 5, 3), np.uint8)), cache)
    assert out.numpy().shape == (300, 6)
    assert sorted(cache) == [(4, 5, 3), (8, 5, 3)] and len(calls) == 3      # one cache entry per input shape


def test_synthetic_weights_are_deterministic(sd_t):
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    again = synthetic_yolov9_state_dict("t", 1234)
    assert all(np.array_equal(sd_t[k], again[k]) for k in sd_t)
    other = synthetic_yolov9_state_dict("t", 99)
    assert not np.array_equal(sd_t["model.list.0.conv.weight"], other["model.list.0.conv.weight"])
    w = sd_t["model.list.4.cv4.conv.weight"]
    assert abs(float(w.mean(axis=(1, 2, 3)).max())) < 1e-6      # zero-sum filters


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_controlled_weight_rounding_host(lib_path, dtype):
    """cc_round_weights (the rounding cc_yolo_finalize applies to conv weights in the plain 16-bit modes; host code, no GPU):
    bit-identical to the emulation in oracle/lowprec_oracle.py; every value is the weight's lower or upper neighbour in the storage type;
    the output channel's total, every input channel's tap-sum and every tap's channel-sum of the rounded weights stay within ~two ulps of the
    float32 sums (round-to-nearest lets them random-walk); values the type already holds exactly are left alone."""
    import torch
    from oracle.lowprec_oracle import LowPrecOracle
    L = _lib.lib()
    t = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    code = {"f16": 1, "bf16": 2}[dtype]
    g = torch.Generator().manual_seed(5)
    w = torch.randn(48, 32, 3, 3, generator=g) * 0.05
    out = np.empty(w.numel(), np.float32)
    wn = np.ascontiguousarray(w.numpy())
    _lib.check(L.cc_round_weights(code, _lib.ptr(wn), 48, 32, 3, _lib.ptr(out)))
    got = torch.from_numpy(out).reshape(w.shape)
    emu = LowPrecOracle.__new__(LowPrecOracle)
    emu.t = t
    assert torch.equal(got, emu.q_feedback(w))                                  # library == emulation, bit for bit
    assert torch.equal(got, got.to(t).float())                                  # representable
    near = w.to(t).float()
    ulp = 2.0 ** (np.floor(np.log2(float(w.abs().max()))) - (10 if dtype == "f16" else 7))
    assert float((got - w).abs().max()) <= 1.001 * ulp                          # a neighbour, never further
    for name, dims in (("total", (1, 2, 3)), ("per input channel", (2, 3)), ("per tap", (1,))):
        err_c = (got.double().sum(dims) - w.double().sum(dims)).abs().max()
        err_n = (near.double().sum(dims) - w.double().sum(dims)).abs().max()
        assert float(err_c) <= 2.0 * ulp, (name, float(err_c), ulp)             # the margins are preserved ...
        if name != "per input channel":
            assert float(err_n) > 2 * float(err_c), (name, float(err_n), float(err_c))   # ... where nearest rounding drifts (sqrt(n) ulps)
    exact = near
    _lib.check(L.cc_round_weights(code, _lib.ptr(np.ascontiguousarray(exact.numpy())), 48, 32, 3, _lib.ptr(out)))
    assert np.array_equal(out.reshape(w.shape), exact.numpy())
    w1 = torch.randn(16, 40, 1, 1, generator=g)                                 # 1x1: one tap, the table has a single column
    o1 = np.empty(w1.numel(), np.float32)
    _lib.check(L.cc_round_weights(code, _lib.ptr(np.ascontiguousarray(w1.numpy())), 16, 40, 1, _lib.ptr(o1)))
    assert torch.equal(torch.from_numpy(o1).reshape(w1.shape), emu.q_feedback(w1))
    assert L.cc_round_weights(0, _lib.ptr(wn), 48, 32, 3, _lib.ptr(out)) != 0   # f32 storage has nothing to round


def test_calibration_aware_rounding_host(lib_path):
    """cc_gptq_round_f16 (the recursion cc_yolo_finalize runs for dtype "f16c"; host code, no GPU) against the restatement in
    oracle/lowprec_oracle.py on random layers with correlated, non-zero-mean inputs (what post-SiLU activations look like): every value is
    f16-representable, identical to the checker on >= 99.5 % of the weights (the two invert H with different factorisation code; a
    difference of one ulp of a double decides a tie now and then), the same expected output error E|dW x|^2 within 2 %, and that error at
    least halved against round-to-nearest.  A singular H (an input channel that is always zero) is made definite by the damping term."""
    import torch
    from oracle.lowprec_oracle import gptq_f16
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for co, ci, rows in ((64, 128, 4000), (96, 256, 6000)):
        base = rng.standard_normal((rows, 32)) @ rng.standard_normal((32, ci)) + 0.3 * rng.standard_normal((rows, ci)) + 0.5
        X = np.maximum(base, 0) * 0.2
        if ci == 256:
            X[:, 7] = 0.0                                                         # a dead input channel: H singular without the damping
        H = np.ascontiguousarray(X.T @ X / rows)
        w = (rng.standard_normal((co, ci)) / np.sqrt(ci)).astype(np.float32)
        out = np.empty_like(w)
        _lib.check(L.cc_gptq_round_f16(_lib.ptr(w), co, ci, _lib.ptr(H), 0.01, _lib.ptr(out)))
        ref = gptq_f16(torch.from_numpy(w), torch.from_numpy(H), 0.01).numpy()
        t = torch.from_numpy(out)
        assert torch.equal(t.to(torch.float16).float(), t)
        assert float((out == ref).mean()) >= 0.995
        proxy = lambda q: float(np.einsum("oi,ij,oj->", (q - w).astype(np.float64), H, (q - w).astype(np.float64)))   # noqa: E731
        near = w.astype(np.float16).astype(np.float32)
        assert abs(proxy(out) / proxy(ref) - 1) < 0.02 and proxy(out) < 0.5 * proxy(near), (proxy(out), proxy(ref), proxy(near))
    bad = -np.eye(8)                                                               # not positive definite even with the damping: reported, not rounded
    assert L.cc_gptq_round_f16(_lib.ptr(np.zeros((4, 8), np.float32)), 4, 8, _lib.ptr(bad), 0.01, _lib.ptr(np.zeros((4, 8), np.float32))) != 0


def test_calibration_aware_rounding_generalises_to_held_out_pixels(lib_path):
    """The point of dtype "f16c", checked on the CPU: second moments of a REAL 1x1 conv input (the f32 oracle's activations entering block 4's
    cv1 of the conditioned YOLOv9-C checkpoint, one noise frame) taken on half of the pixels; the weights cc_gptq_round_f16 returns for that
    conv have a smaller output error E|dW x|^2 than round-to-nearest on the OTHER half of the pixels (and on the pixels of another frame) -
    by the factor the recursion promises on its own calibration set, within a small margin."""
    import torch
    from clearcam_amd.weights import conditioned_yolov9_state_dict
    from oracle import yolov9_oracle as yo
    sd = conditioned_yolov9_state_dict("c", 1234, exact=False)
    name = "model.list.4.cv1.conv"
    o = yo.YOLOv9Oracle("c", 320, sd)
    seen = {}
    orig = o._conv2d
    def hooked(x, n, stride=1, groups=1):
        if n == name:
            seen.setdefault("x", []).append(x.clone())
        return orig(x, n, stride, groups)
    o._conv2d = hooked
    frames = np.random.default_rng(8).integers(0, 256, (2, 320, 320, 3), dtype=np.uint8)
    with torch.no_grad():
        o.features(o.network_input(frames[:1])); o.features(o.network_input(frames[1:]))
    X = [x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).double().numpy() for x in seen["x"]]
    cal, held, other = X[0][0::2], X[0][1::2], X[1]
    w = np.ascontiguousarray(sd[name + ".weight"][:, :, 0, 0])
    H = np.ascontiguousarray(cal.T @ cal / len(cal))
    q = np.empty_like(w)
    _lib.check(_lib.lib().cc_gptq_round_f16(_lib.ptr(w), w.shape[0], w.shape[1], _lib.ptr(H), 0.03, _lib.ptr(q)))
    near = w.astype(np.float16).astype(np.float32)
    err = lambda qq, xs: float((((qq - w).astype(np.float64) @ xs.T) ** 2).mean())      # noqa: E731  E|dW x|^2 over pixels and output channels
    gains = {k: err(near, xs) / err(q, xs) for k, xs in (("calibration", cal), ("held-out pixels", held), ("another frame", other))}
    print(gains)
    assert gains["calibration"] > 3.0 and gains["held-out pixels"] > 0.8 * gains["calibration"] and gains["another frame"] > 0.6 * gains["calibration"], gains
