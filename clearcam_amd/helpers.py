"""Host-side mirrors of the small helpers the reference's call sites use (utils/helpers.py)."""
from __future__ import annotations

import numpy as np


class Tensor:
    """Minimal stand-in for the ``tinygrad.Tensor`` objects crossing the hot-path boundary.

    The reference's callers only construct (``Tensor(frame)``, clearcam.py:582), cast
    (``.cast(dtypes.float32)``, test/run_mot.py:33), ``unsqueeze(0)`` (clearcam.py:1285) and read back
    (``.numpy()``); everything else happens inside the model call, which now runs in HIP.
    """

    def __init__(self, data):
        if isinstance(data, Tensor):
            data = data.data
        elif hasattr(data, "detach") and hasattr(data, "cpu"):      # torch tensor
            data = data.detach().cpu().numpy()
        self.data = np.asarray(data)

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def dtype(self):
        return self.data.dtype

    def cast(self, dtype):
        name = dtype if isinstance(dtype, str) else getattr(dtype, "name", None) or str(np.dtype(dtype))
        return Tensor(self.data.astype(np.dtype(name)))

    def unsqueeze(self, dim):
        return Tensor(np.expand_dims(self.data, dim))

    def numpy(self):
        return self.data

    def __matmul__(self, other):
        return Tensor(self.data @ as_numpy(other))

    @property
    def T(self):
        return Tensor(self.data.T)

    def __getitem__(self, i):
        return Tensor(self.data[i])

    def item(self):
        return self.data.item()

    @staticmethod
    def rand(*shape):
        return Tensor(np.random.rand(*shape).astype(np.float32))


def as_numpy(x) -> np.ndarray:
    """numpy view of whatever a reference call site passes (ndarray, Tensor shim, torch tensor, .numpy()-able)."""
    if isinstance(x, np.ndarray):
        return x
    if isinstance(x, Tensor):
        return x.data
    if hasattr(x, "detach") and hasattr(x, "cpu"):
        return x.detach().cpu().numpy()
    if hasattr(x, "numpy"):
        return np.asarray(x.numpy())
    return np.asarray(x)


def jit_infer(fn, x, jit_cache):
    """``utils/helpers.py:214-221``: one compiled graph per input shape, the callable passed on EVERY call.

    The reference stores ``TinyJit(lambda x, fn: fn(x))`` per shape and calls ``jit_cache[shape](x, fn)``: the cache
    entry never binds a model, so clearcam.py can close and re-create its CLIP model (settings toggle, :1250-1253) or
    run two models over inputs of one shape without ever clearing ``jit_cache``.  Here the per-shape plan (buffers +
    captured hipGraph) lives inside the model handle, so the entry only records that the shape has been seen
    (``jit_cache.clear()`` on a settings change, clearcam.py:1260-1262, keeps working)."""
    jit_cache.setdefault(tuple(x.shape), True)
    return fn(x)
