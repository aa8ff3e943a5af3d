"""A stand-in for the `tinygrad` package, just large enough to EXECUTE the reference's model code in this container.

TEST TOOLING, NOT PRODUCT.  tinygrad (the reference's pinned dependency, fe39cf14) is not installed and cannot be fetched, so the
reference's own detection/yolov9.py, models/objects.py (OpenCLIP), models/adaface.py and models/blazeface.py cannot run as
they are.  This package implements the slice of tinygrad's public `Tensor` / `nn` API those four files call, with the
semantics tinygrad documents, on top of PyTorch-CPU float32.  `tools/make_reference_run_golden.py` puts it on sys.path in
front of /root/reference, imports the reference's modules UNCHANGED, loads seeded synthetic weights through the reference's
own `load_state_dict(self, ...)` call and stores what the reference's classes return under tests/golden/refrun_*.npz.
The CPU oracle (oracle/*.py) is then checked against those files, which pins the oracle's layer wiring, concat orders,
decode, NMS and box scaling to the reference's code rather than to a reading of it.

What this does NOT pin: tinygrad's own kernels (replaced by torch ops of the same definition), i.e. summation order, the
uint8 fixed-point `interpolate` (restated below from tinygrad's Tensor.interpolate / Tensor.lerp) and the tie order of
`topk` (stable, lower index first, here).

Nothing under clearcam_amd/, bench.py or the GPU tests imports this package.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import dtype as _dtype_mod
from .dtype import dtypes


def _u(x):
    return x.t if isinstance(x, Tensor) else x


def _w(t):
    return Tensor(t) if isinstance(t, torch.Tensor) else t


def _shape_args(shape):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        return tuple(shape[0])
    return tuple(int(s) for s in shape)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Tensor:
    training = False

    def __init__(self, data=None, dtype=None, device=None, requires_grad=None):
        if isinstance(data, Tensor):
            t = data.t
        elif isinstance(data, torch.Tensor):
            t = data
        elif isinstance(data, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(data))
        elif isinstance(data, bool):
            t = torch.tensor(data)
        elif isinstance(data, int):
            t = torch.tensor(data, dtype=torch.int32)
        elif isinstance(data, float):
            t = torch.tensor(data, dtype=torch.float32)
        elif isinstance(data, (list, tuple)):
            a = np.array(data)
            if a.dtype == np.float64:
                a = a.astype(np.float32)
            elif a.dtype == np.int64:
                a = a.astype(np.int32)
            t = torch.from_numpy(a)
        else:
            raise TypeError(f"refshim Tensor: unsupported data {type(data)}")
        if dtype is not None:
            t = t.to(_dtype_mod.to_torch(dtype))
        self.t = t

    # ---- creation ---------------------------------------------------------------------------------------------------
    @staticmethod
    def empty(*shape, **kw):
        return Tensor(torch.zeros(_shape_args(shape), dtype=torch.float32))

    @staticmethod
    def zeros(*shape, **kw):
        return Tensor(torch.zeros(_shape_args(shape), dtype=torch.float32))

    @staticmethod
    def ones(*shape, **kw):
        return Tensor(torch.ones(_shape_args(shape), dtype=torch.float32))

    @staticmethod
    def full(shape, fill_value, **kw):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        dt = torch.float32 if isinstance(fill_value, float) else torch.int32
        return Tensor(torch.full(shape, fill_value, dtype=dt))

    @staticmethod
    def arange(start, stop=None, step=1, **kw):
        if stop is None:
            start, stop = 0, start
        isf = any(isinstance(v, float) for v in (start, stop, step))
        return Tensor(torch.arange(start, stop, step, dtype=torch.float32 if isf else torch.int32))

    def zeros_like(self, **kw):
        return Tensor(torch.zeros_like(_u(self)))

    # ---- properties -------------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.ndim

    @property
    def dtype(self):
        return _dtype_mod.from_torch(self.t.dtype)

    @property
    def device(self):
        return "CPU"

    @property
    def T(self):
        return Tensor(self.t.transpose(-1, -2)) if self.t.ndim >= 2 else self

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def numel(self):
        return self.t.numel()

    def numpy(self):
        return self.t.detach().contiguous().numpy().copy()

    def item(self):
        return self.t.item()

    def tolist(self):
        return self.t.tolist()

    def realize(self, *a):
        return self

    def contiguous(self):
        return Tensor(self.t.contiguous().clone())

    def detach(self):
        return self

    def clone(self):
        return Tensor(self.t.clone())

    def to(self, *a, **k):
        return self

    def cast(self, dt):
        return Tensor(self.t.to(_dtype_mod.to_torch(dt)))

    def float(self):
        return Tensor(self.t.float())

    def half(self):
        return Tensor(self.t.half())

    def int(self):
        return Tensor(self.t.to(torch.int32))

    def replace(self, other):
        o = _u(other)
        assert tuple(o.shape) == tuple(self.t.shape), f"replace: shape {tuple(o.shape)} != {tuple(self.t.shape)}"
        self.t = o
        return self

    def assign(self, other):
        self.t = _u(other).to(self.t.dtype).reshape(self.t.shape).clone()
        return self

    # ---- arithmetic -------------------------------------------------------------------------------------------------
    def _bin(self, other, fn, rev=False):
        a, b = self.t, _u(other)
        if isinstance(b, np.generic):
            b = b.item()
        return Tensor(fn(b, a) if rev else fn(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return Tensor(torch.as_tensor(_u(o)) - self.t)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)
    def __rtruediv__(self, o): return Tensor(torch.true_divide(torch.as_tensor(_u(o)), self.t))
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __neg__(self): return Tensor(-self.t)
    def __matmul__(self, o): return Tensor(self.t @ _u(o))
    def __and__(self, o): return self._bin(o, torch.logical_and if self.t.dtype == torch.bool else torch.bitwise_and)
    def __or__(self, o): return self._bin(o, torch.logical_or if self.t.dtype == torch.bool else torch.bitwise_or)
    def __invert__(self): return Tensor(~self.t)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __eq__(self, o): return self._bin(o, torch.eq)      # elementwise, like tinygrad
    def __ne__(self, o): return self._bin(o, torch.ne)
    __hash__ = object.__hash__

    def _ibin(self, o, fn):
        r = fn(self.t, _u(o))
        self.t = r.to(self.t.dtype) if self.t.is_floating_point() else r
        return self

    def __iadd__(self, o): return self._ibin(o, torch.add)
    def __isub__(self, o): return self._ibin(o, torch.sub)
    def __imul__(self, o): return self._ibin(o, torch.mul)
    def __itruediv__(self, o): return self._ibin(o, torch.true_divide)

    def add(self, o): return self + o
    def sub(self, o): return self - o
    def mul(self, o): return self * o
    def div(self, o): return self / o
    def pow(self, o): return self ** o
    def matmul(self, o): return self @ o
    def dot(self, o): return self @ o
    def _promoted(self, o):
        a, b = self.t, torch.as_tensor(_u(o))
        dt = torch.promote_types(a.dtype, b.dtype)
        return torch.broadcast_tensors(a.to(dt), b.to(dt))

    def maximum(self, o): return Tensor(torch.maximum(*self._promoted(o)))
    def minimum(self, o): return Tensor(torch.minimum(*self._promoted(o)))

    def sqrt(self): return Tensor(torch.sqrt(self.t))
    def rsqrt(self): return Tensor(torch.rsqrt(self.t))
    def exp(self): return Tensor(torch.exp(self.t))
    def log(self): return Tensor(torch.log(self.t))
    def tanh(self): return Tensor(torch.tanh(self.t))
    def abs(self): return Tensor(torch.abs(self.t))
    def relu(self): return Tensor(torch.relu(self.t))
    def sigmoid(self): return Tensor(torch.sigmoid(self.t))
    def silu(self): return Tensor(self.t * torch.sigmoid(self.t))                       # tinygrad: x * sigmoid(x)
    def gelu(self):                                                                     # tinygrad's gelu is the tanh form
        x = self.t
        return Tensor(0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3))))
    def clip(self, min_=None, max_=None): return Tensor(torch.clamp(self.t, min_, max_))
    clamp = clip

    def where(self, a, b):
        c = self.t if self.t.dtype == torch.bool else self.t != 0
        a, b = _u(a), _u(b)
        if not isinstance(a, torch.Tensor) and not isinstance(b, torch.Tensor):
            a = torch.tensor(a, dtype=torch.float32 if isinstance(a, float) or isinstance(b, float) else torch.int32)
        if not isinstance(a, torch.Tensor):
            a = torch.tensor(a, dtype=b.dtype)
        if not isinstance(b, torch.Tensor):
            b = torch.tensor(b, dtype=a.dtype)
        return Tensor(torch.where(c, a, b))

    def masked_fill(self, mask, value):
        return Tensor(self.t.masked_fill(_u(mask), value))

    def lerp(self, end, weight):
        """tinygrad Tensor.lerp: uint8 inputs with a float weight take a 7-bit fixed-point path with an int8 difference."""
        a, b, w = self.t, _u(end), _u(weight)
        if a.dtype == torch.uint8 and isinstance(w, torch.Tensor) and w.is_floating_point():
            W = 7
            wi = (w * (1 << W) + 0.5).to(torch.int16)
            d = (b.to(torch.int16) - a.to(torch.int16)).to(torch.int8)                  # (end - self) in uint8, then cast(int8): wraps
            v = (d.to(torch.int16) * wi + (1 << (W - 1))).to(torch.int32) & 0xFFFF       # int16 product, viewed as uint16
            return Tensor(((a.to(torch.int32) + (v >> W)) & 0xFF).to(torch.uint8))
        return Tensor(a + (b - a) * w)

    # ---- reductions -------------------------------------------------------------------------------------------------
    def _red(self, fn, axis, keepdim):
        if axis is None:
            r = fn(self.t.reshape(-1), 0)
            r = r[0] if isinstance(r, tuple) else r
            return Tensor(r.reshape((1,) * self.t.ndim) if keepdim else r)
        r = fn(self.t, axis, keepdim)
        return Tensor(r[0] if isinstance(r, tuple) else r)

    def sum(self, axis=None, keepdim=False):
        t = self.t.to(torch.int32) if self.t.dtype == torch.bool else self.t
        if axis is None:
            r = t.sum()
            return Tensor(r.reshape((1,) * t.ndim) if keepdim else r)
        return Tensor(t.sum(axis, keepdim=keepdim))

    def mean(self, axis=None, keepdim=False):
        return Tensor(self.t.mean() if axis is None else self.t.mean(axis, keepdim=keepdim))

    def max(self, axis=None, keepdim=False):
        return Tensor(self.t.max()) if axis is None else Tensor(self.t.max(axis, keepdim=keepdim)[0])

    def min(self, axis=None, keepdim=False):
        return Tensor(self.t.min()) if axis is None else Tensor(self.t.min(axis, keepdim=keepdim)[0])

    def argmax(self, axis=None, keepdim=False):
        # first occurrence of the maximum, like tinygrad
        t = self.t
        if axis is None:
            flat = t.reshape(-1)
            return Tensor((flat == flat.max()).to(torch.int32).argmax().to(torch.int32))
        m = t.max(axis, keepdim=True)[0]
        return Tensor((t == m).to(torch.int32).argmax(axis, keepdim=keepdim).to(torch.int32))

    def softmax(self, axis=-1):
        return Tensor(torch.softmax(self.t, axis))

    def topk(self, k, dim=-1, largest=True, sorted_=True):
        v, i = torch.sort(self.t, dim=dim, descending=largest, stable=True)             # ties: lower index first
        sl = [slice(None)] * self.t.ndim
        sl[dim] = slice(0, k)
        return Tensor(v[tuple(sl)]), Tensor(i[tuple(sl)].to(torch.int32))

    def triu(self, diagonal=0): return Tensor(torch.triu(self.t, diagonal))
    def tril(self, diagonal=0): return Tensor(torch.tril(self.t, diagonal))

    def layernorm(self, axis=-1, eps=1e-5):
        y = self.t - self.t.mean(axis, keepdim=True)
        return Tensor(y * torch.rsqrt((y * y).mean(axis, keepdim=True) + eps))

    def batchnorm(self, weight, bias, mean, invstd, axis=1):
        shape = [1] * self.t.ndim
        shape[axis] = -1
        x = self.t - _u(mean).reshape(shape)
        if weight is not None:
            x = x * _u(weight).reshape(shape)
        r = x * (_u(invstd).reshape(shape) if _u(invstd).ndim == 1 else _u(invstd))
        return Tensor(r + _u(bias).reshape(shape) if bias is not None else r)

    def linear(self, weight, bias=None):
        x = self.t @ _u(weight)
        return Tensor(x + _u(bias) if bias is not None else x)

    def sequential(self, ll):
        x = self
        for f in ll:
            x = f(x)
        return x

    # ---- movement ---------------------------------------------------------------------------------------------------
    def reshape(self, *shape): return Tensor(self.t.reshape(_shape_args(shape)))
    def view(self, *shape): return Tensor(self.t.reshape(_shape_args(shape)))
    def permute(self, *order): return Tensor(self.t.permute(_shape_args(order)))
    def transpose(self, a=1, b=0): return Tensor(self.t.transpose(a, b))
    def unsqueeze(self, dim): return Tensor(self.t.unsqueeze(dim))
    def squeeze(self, dim=None): return Tensor(self.t.squeeze() if dim is None else self.t.squeeze(dim))
    def flatten(self, start=0, end=-1): return Tensor(self.t.flatten(start, end))
    def expand(self, *shape): return Tensor(self.t.expand(_shape_args(shape)))
    def repeat(self, *reps): return Tensor(self.t.repeat(_shape_args(reps)))
    def repeat_interleave(self, n, dim=None): return Tensor(self.t.repeat_interleave(n, dim=dim))
    def flip(self, *axis): return Tensor(self.t.flip(_shape_args(axis)))

    def chunk(self, n, dim=0): return tuple(Tensor(c) for c in self.t.chunk(n, dim))

    def split(self, sizes, dim=0):
        return tuple(Tensor(c) for c in self.t.split(sizes if isinstance(sizes, int) else list(sizes), dim))

    def cat(self, *args, dim=0):
        ts = [_u(self)] + [_u(a) for a in args]
        dt = torch.result_type(ts[0], ts[1]) if len(ts) > 1 else ts[0].dtype
        for t in ts[2:]:
            dt = torch.promote_types(dt, t.dtype)
        return Tensor(torch.cat([t.to(dt) for t in ts], dim))

    def stack(self, *args, dim=0):
        ts = list(self) if isinstance(self, (tuple, list)) else [self]
        ts = [_u(a) for a in ts + list(args)]
        dt = ts[0].dtype
        for t in ts[1:]:
            dt = torch.promote_types(dt, t.dtype)
        return Tensor(torch.stack([t.to(dt) for t in ts], dim))

    def pad(self, padding, mode="constant", value=0.0):
        nd = self.t.ndim
        if len(padding) and isinstance(padding[0], (tuple, list, type(None))):          # ((before, after), ...) per axis
            per = [(0, 0) if p is None else tuple(p) for p in padding]
            assert len(per) == nd
        else:                                                                            # flat, last axis first (torch order)
            assert len(padding) % 2 == 0
            per = [(0, 0)] * nd
            for i in range(len(padding) // 2):
                per[nd - 1 - i] = (padding[2 * i], padding[2 * i + 1])
        flat = []
        for b, a in reversed(per):
            flat += [int(b), int(a)]
        return Tensor(F.pad(self.t, flat, mode="constant", value=value))

    def gather(self, dim, index): return Tensor(torch.gather(self.t, dim, _u(index).to(torch.int64)))

    # ---- pooling / conv / resize -----------------------------------------------------------------------------------
    def avg_pool2d(self, kernel_size=(2, 2), stride=None, dilation=1, padding=0, ceil_mode=False, count_include_pad=True):
        assert dilation == 1
        return Tensor(F.avg_pool2d(self.t, _pair(kernel_size), _pair(stride if stride is not None else kernel_size), _pair(padding),
                                   ceil_mode=ceil_mode, count_include_pad=count_include_pad))

    def max_pool2d(self, kernel_size=(2, 2), stride=None, dilation=1, padding=0, ceil_mode=False):
        return Tensor(F.max_pool2d(self.t, _pair(kernel_size), _pair(stride if stride is not None else kernel_size), _pair(padding),
                                   _pair(dilation), ceil_mode=ceil_mode))

    def conv2d(self, weight, bias=None, groups=1, stride=1, dilation=1, padding=0):
        return Tensor(F.conv2d(self.t, _u(weight), _u(bias) if bias is not None else None, _pair(stride), _pair(padding), _pair(dilation), groups))

    def interpolate(self, size, mode="linear", align_corners=False):
        """tinygrad Tensor.interpolate: the trailing len(size) axes, one axis at a time (last axis first)."""
        x = self
        nd = self.t.ndim
        for i in range(-1, -len(size) - 1, -1):
            n_in, n_out = x.shape[i], int(size[i])
            ax = nd + i
            if mode == "nearest":
                idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * (n_in / n_out)).to(torch.int64)
                x = Tensor(x.t.index_select(ax, idx))
                continue
            assert mode == "linear"
            scale = (n_in - int(align_corners)) / (n_out - int(align_corners))
            index = torch.arange(n_out, dtype=torch.float32) * scale if align_corners else (torch.arange(n_out, dtype=torch.float32) + 0.5) * scale - 0.5
            index = index.clamp(0, n_in - 1)
            low, high = index.floor().to(torch.int64), index.ceil().to(torch.int64)
            perc = index - low.to(torch.float32)
            shp = [1] * nd
            shp[ax] = n_out
            x = Tensor(x.t.index_select(ax, low)).lerp(Tensor(x.t.index_select(ax, high)), Tensor(perc.reshape(shp)))
        return x

    # ---- indexing ---------------------------------------------------------------------------------------------------
    def _index(self, idx):
        """(torch-style index, dims to flip afterwards)."""
        idx = idx if isinstance(idx, tuple) else (idx,)
        idx = tuple(_u(i).to(torch.int64) if isinstance(_u(i), torch.Tensor) and _u(i).dtype != torch.bool else _u(i) for i in idx)
        n_spec = sum(1 for i in idx if i is not None and i is not Ellipsis)
        out, flips, od = [], [], 0
        for i in idx:
            if i is Ellipsis:
                k = self.t.ndim - n_spec
                out += [slice(None)] * k
                od += k
            elif isinstance(i, slice) and i.step is not None and i.step < 0:
                assert i.start is None and i.stop is None and i.step == -1, "refshim: only [::-1] is supported"
                out.append(slice(None)); flips.append(od); od += 1
            elif isinstance(i, int):
                out.append(i)
            else:
                out.append(i); od += 1
        assert not flips or not any(isinstance(i, (list, torch.Tensor)) for i in out), "refshim: [::-1] next to fancy indexing"
        return tuple(out), flips

    def __getitem__(self, idx):
        ti, flips = self._index(idx)
        r = self.t[ti]
        return Tensor(r.flip(flips) if flips else r)

    def __setitem__(self, idx, value):
        ti, flips = self._index(idx)
        assert not flips
        v = _u(value)
        self.t = self.t.clone()
        self.t[ti] = v.to(self.t.dtype) if isinstance(v, torch.Tensor) else v

    def __len__(self):
        return self.t.shape[0]

    def __iter__(self):
        return (Tensor(r) for r in self.t)

    def __bool__(self):
        return bool(self.t)

    def __repr__(self):
        return f"<refshim Tensor {tuple(self.t.shape)} {self.t.dtype}>"


class TinyJit:
    """Pass-through: calling the wrapped function directly is what TinyJit's first (un-captured) call does."""
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *a, **k):
        return self.fn(*a, **k)

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        fn = self.fn
        return lambda *a, **k: fn(obj, *a, **k)


class _Device:
    DEFAULT = "CPU"

    def __getitem__(self, k):
        return self


Device = _Device()


def getenv(key, default=0):
    import os
    return type(default)(os.getenv(key, default))


from . import nn  # noqa: E402,F401
