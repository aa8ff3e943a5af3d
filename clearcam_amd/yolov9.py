"""Host-side mirror of ``detection/yolov9.py``: same class name, constructor and call surface,
compute in libclearcam_hip (HIP/MFMA) through the C ABI in include/clearcam_hip.h.

    model = YOLOv9("c", 640, state_dict=sd)          # reference: YOLOv9(size, res) + download
    preds = jit_infer(model, Tensor(frame), cache).numpy()     # (300,6) float32, clearcam.py:583

There is no CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .arch import YOLO_ARCH
from .helpers import Tensor, as_numpy
from .weights import load_safetensors

# "f16s": f16 activations with every conv weight carried as two f16 planes (W = W_hi + W_lo, ~22 significant bits, f32 accumulation):
# the 16-bit mode whose WEIGHTS are exact to f32 for any checkpoint.  "f16h" keeps the second plane only where it is needed - the 1x1
# convs of the backbone (blocks 0-9) and the stem conv; every other conv carries one f16 plane with controlled rounding (which balances a
# 3x3 filter's taps and has nothing to balance in a 1x1) - and stays as close to the f32 oracle on three independently calibrated
# un-rounded checkpoints at 1.4x the frame rate.  Plain f16 / bf16 round every weight
# to 11 / 8 bits: speed modes.  "f32" is the exact-arithmetic parity mode.
DTYPES = {"f32": 0, "float32": 0, "f16": 1, "float16": 1, "half": 1, "bf16": 2, "bfloat16": 2, "f16s": 3, "f16_split": 3, "f16h": 4, "f16c": 5}
MAX_DET = 300


class YOLOv9:
    def __init__(self, size: str = "t", res: int = 1280, state_dict: Optional[Dict[str, np.ndarray]] = None,
                 weights: Optional[str] = None, dtype: str = "f16h", device: int = 0, calibration_frames=None):
        """dtype: "f16h" (default) - f16 activations, split f16 weights in the backbone's 1x1 convs and the stem, one controlled-rounded
        f16 plane elsewhere: detections within the reference tolerance on un-rounded float32 checkpoints; "f16s" - split weights in every
        conv (weights exact to f32 whatever the checkpoint; 0.7x the rate); "f32" - the exact-arithmetic parity mode (5x slower); "f16" / "bf16" - speed modes whose 11 / 8-bit weight
        rounding (controlled rounding: filter sums preserved) moves boxes by up to a pixel / several pixels on the conditioned synthetic checkpoint.  f16
        storage saturates at 65504: a checkpoint whose activations exceed that needs "bf16" or "f32".
        "f16c" (round 5) - one f16 plane per conv (two in the stem conv) at plain f16's rate, the 1x1 convs' weights rounded by a
        calibration-aware recursion on `calibration_frames` ((B,H,W,3) BGR uint8 / float32 array: a few frames of the camera; default: four
        frames of seeded white noise) - "f16h"'s tolerance on inputs like the calibration frames."""
        if size not in YOLO_ARCH and size != "e":
            raise ValueError(f"unsupported size {size!r}: t, s, m, c, e")
        self.size, self.res, self.dtype, self.device = size, res, dtype, device
        if state_dict is None:
            # reference: safe_load(fetch(".../yolov9-{size}.safetensors")) (yolov9.py:372); no network here
            path = weights or os.path.join(os.environ.get("CLEARCAM_WEIGHTS_DIR", "weights"), f"yolov9-{size}.safetensors")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: pass state_dict= or weights= (the reference downloads "
                                        f"yolov9-{size}.safetensors from HuggingFace; there is no network here)")
            state_dict = load_safetensors(path)
        L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(L.cc_yolo_create(C.byref(self._h), size.encode(), res, DTYPES[dtype], device))
        for name, arr in state_dict.items():
            if name.endswith(("anchors", "strides")):      # Tensor attributes the reference recomputes per call
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(L.cc_yolo_load(self._h, name.encode(), _lib.ptr(a), shp, a.ndim))
        if calibration_frames is not None:
            if dtype != "f16c":
                raise ValueError('calibration_frames are for dtype "f16c"')
            f = np.ascontiguousarray(as_numpy(calibration_frames))
            if f.ndim != 4 or f.shape[3] != 3 or f.dtype not in (np.uint8, np.float32):
                raise ValueError(f"calibration_frames must be (B,H,W,3) uint8 or float32, got {f.shape} {f.dtype}")
            _lib.check(L.cc_yolo_calibrate(self._h, _lib.ptr(f), f.shape[0], f.shape[1], f.shape[2], int(f.dtype == np.float32)))
        _lib.check(L.cc_yolo_finalize(self._h))

    def calibration_info(self):
        """dtype "f16c": (packed 1x1 convs rounded by the calibration-aware recursion, convs left to the plain controlled rounding)."""
        a, b = C.c_int(), C.c_int()
        _lib.check(_lib.lib().cc_yolo_calibration_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- reference surface ------------------------------------------------------------------------
    def __call__(self, frame) -> Tensor:
        """``YOLOv9.__call__(frame)``: (H,W,3) BGR uint8/float32 -> Tensor (300,6)."""
        f = as_numpy(frame)
        if f.ndim != 3 or f.shape[2] != 3:
            raise ValueError(f"frame must be (H,W,3), got {f.shape}")
        return Tensor(self.detect_batch(f[None])[0])

    # -- batched extension (one camera per row) ---------------------------------------------------
    def detect_batch(self, frames) -> np.ndarray:
        """(B,H,W,3) BGR uint8/float32 host array or CUDA torch tensor -> (B,300,6) float32 ndarray."""
        L = _lib.lib()
        on_dev = bool(getattr(frames, "is_cuda", False))
        if on_dev:
            import torch
            f = frames.contiguous()
            is_f32 = f.dtype == torch.float32
            if not is_f32 and f.dtype != torch.uint8:
                raise TypeError("device frames must be uint8 or float32")
            B, H, W, ch = f.shape
            torch.cuda.current_stream(f.device).synchronize()
        else:
            f = as_numpy(frames)
            if f.dtype != np.uint8:
                f = f.astype(np.float32, copy=False)
            f = np.ascontiguousarray(f)
            is_f32 = f.dtype == np.float32
            B, H, W, ch = f.shape
        if ch != 3:
            raise ValueError("frames must be (B,H,W,3)")
        out = np.empty((B, MAX_DET, 6), np.float32)
        _lib.check(L.cc_yolo_detect(self._h, _lib.ptr(f), B, H, W, int(is_f32), int(on_dev), _lib.ptr(out), 0, None))
        return out

    def detect_batch_device(self, frames, out):
        """Device-resident variant for benchmarks: frames/out are CUDA torch tensors; no host copies, no sync."""
        import torch
        B, H, W, _ = frames.shape
        s = torch.cuda.current_stream(frames.device).cuda_stream
        _lib.check(_lib.lib().cc_yolo_detect(self._h, _lib.ptr(frames), B, H, W, int(frames.dtype == torch.float32), 1,
                                             _lib.ptr(out), 1, C.c_void_p(s)))
        return out

    # -- batches in flight (throughput mode: one GPU serving many cameras) --------------------------
    def set_in_flight(self, n: int) -> None:
        """n slots (own stream, arena and graph each): consecutive submit() calls overlap - the last layers of one batch run beside
        the first layers of the next.  Results are bit-identical to detect_batch_device; n = 1 is the default."""
        _lib.check(_lib.lib().cc_yolo_set_in_flight(self._h, n))
        self.in_flight = n

    def submit(self, frames, out) -> int:
        """Queue one batch on the next slot: `frames` (B,H,W,3) uint8 / float32 and `out` (B,300,6) float32 are torch tensors, each
        either on the device or in PINNED host memory (then the slot's stream also carries the upload / download, which overlap the
        other slots' kernels); numpy arrays are taken as pageable host memory (correct, but the runtime stages such copies and the
        call may block).  Device frames are taken as ready on the current torch stream, which does NOT wait for the result -
        wait(ticket) does.  `frames` and `out` must stay alive (and `out` unshared with other submissions in flight) until then."""
        B, H, W, _ = frames.shape
        t = C.c_longlong()
        if isinstance(frames, np.ndarray) or isinstance(out, np.ndarray):
            if not (isinstance(frames, np.ndarray) and isinstance(out, np.ndarray)):
                raise TypeError("frames and out must both be torch tensors or both numpy arrays")
            if not frames.flags.c_contiguous or not out.flags.c_contiguous or out.dtype != np.float32 or frames.dtype not in (np.uint8, np.float32):
                raise ValueError("frames: C-contiguous uint8 / float32, out: C-contiguous float32")
            _lib.check(_lib.lib().cc_yolo_submit(self._h, _lib.ptr(frames), B, H, W, int(frames.dtype == np.float32), 0, _lib.ptr(out), 0, None, C.byref(t)))
            return t.value
        import torch
        if not frames.is_contiguous() or not out.is_contiguous():
            raise ValueError("frames and out must be contiguous")
        if (not frames.is_cuda and not frames.is_pinned()) or (not out.is_cuda and not out.is_pinned()):
            raise ValueError("host tensors handed to submit() must be pinned")
        s = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream if frames.is_cuda else 0
        _lib.check(_lib.lib().cc_yolo_submit(self._h, _lib.ptr(frames), B, H, W, int(frames.dtype == torch.float32), int(frames.is_cuda),
                                             _lib.ptr(out), int(out.is_cuda), C.c_void_p(s), C.byref(t)))
        return t.value

    def wait(self, ticket: int, host: bool = False) -> None:
        """Make the current torch stream (host=True: the calling thread) wait for a submission's result."""
        import torch
        s = None if host else C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)
        _lib.check(_lib.lib().cc_yolo_wait(self._h, ticket, s))

    # -- parity taps --------------------------------------------------------------------------------
    def get_tensor(self, name: str) -> np.ndarray:
        L = _lib.lib()
        shp, nd = (C.c_int64 * 4)(), C.c_int()
        _lib.check(L.cc_yolo_get_tensor(self._h, name.encode(), None, shp, C.byref(nd)))
        out = np.empty(tuple(shp[i] for i in range(nd.value)), np.float32)
        _lib.check(L.cc_yolo_get_tensor(self._h, name.encode(), _lib.ptr(out), shp, C.byref(nd)))
        return out

    def nonfinite(self) -> int:
        """Anchors whose logits were not finite since the last query (f16 activations past 65504): non-zero means this checkpoint needs
        dtype "bf16" or "f32".  Host-output calls (detect_batch, __call__) raise by themselves; device-output / submit() calls do not."""
        n = C.c_int()
        _lib.check(_lib.lib().cc_yolo_nonfinite(self._h, C.byref(n)))
        return n.value

    def last_gpu_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().cc_yolo_last_gpu_ms(self._h, C.byref(ms)))
        return ms.value

    def profile(self, iters: int = 3) -> dict:
        """Per-kernel-family GPU time of the last plan (hipEvents around every launch, eager replay)."""
        ms, macs, n = (C.c_float * 5)(), C.c_double(), C.c_int()
        _lib.check(_lib.lib().cc_yolo_profile(self._h, iters, ms, C.byref(macs), C.byref(n)))
        return {"conv_ms": ms[0], "pool_ms": ms[1], "decode_ms": ms[2], "nms_ms": ms[3], "stem_ms": ms[4],
                "alg_macs_per_step": macs.value, "conv_launches": n.value}

    def profile_graph(self, which: int, iters: int = 10) -> float:
        """ms per replay of a subset of the last plan's launches captured into a hipGraph of its own (one event pair around `iters`
        replays): which = 0 the conv / GEMM launches, 1 every other launch, 2 the whole step."""
        ms = C.c_float()
        _lib.check(_lib.lib().cc_yolo_profile_graph(self._h, iters, which, C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().cc_yolo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
