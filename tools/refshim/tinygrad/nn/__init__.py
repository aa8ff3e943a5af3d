"""tinygrad.nn stand-in (test tooling only, see tinygrad/__init__.py).  Layer definitions follow tinygrad's documented ones:
parameters are plain attributes named weight / bias / running_mean / running_var / num_batches_tracked, so that
`nn.state.get_state_dict` produces the same key names the reference's checkpoints use."""
import torch

from .. import Tensor
from . import state  # noqa: F401


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d:
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        self.kernel_size = _pair(kernel_size)
        self.stride, self.dilation, self.groups, self.padding = stride, dilation, groups, padding
        self.weight = Tensor.empty(out_channels, in_channels // groups, *self.kernel_size)
        self.bias = Tensor.empty(out_channels) if bias else None

    def __call__(self, x):
        return x.conv2d(self.weight, self.bias, self.groups, self.stride, self.dilation, self.padding)


class Linear:
    def __init__(self, in_features, out_features, bias=True):
        self.weight = Tensor.empty(out_features, in_features)
        self.bias = Tensor.empty(out_features) if bias else None

    def __call__(self, x):
        return x.linear(self.weight.transpose(), self.bias)


class Embedding:
    def __init__(self, vocab_size, embed_size):
        self.vocab_sz, self.embed_sz = vocab_size, embed_size
        self.weight = Tensor.empty(vocab_size, embed_size)

    def __call__(self, idx):
        return Tensor(self.weight.t[idx.t.to(torch.int64)])


class LayerNorm:
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        self.normalized_shape = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
        self.axis, self.eps = tuple(-1 - i for i in range(len(self.normalized_shape))), eps
        self.weight = Tensor.ones(*self.normalized_shape) if elementwise_affine else None
        self.bias = Tensor.zeros(*self.normalized_shape) if elementwise_affine else None

    def __call__(self, x):
        x = x.layernorm(eps=self.eps, axis=self.axis)
        return x if self.weight is None else x * self.weight + self.bias


class BatchNorm:
    def __init__(self, sz, eps=1e-5, affine=True, track_running_stats=True, momentum=0.1):
        self.eps, self.track_running_stats, self.momentum = eps, track_running_stats, momentum
        self.weight = Tensor.ones(sz) if affine else None
        self.bias = Tensor.zeros(sz) if affine else None
        self.num_batches_tracked = Tensor(torch.zeros(1, dtype=torch.int64))
        if track_running_stats:
            self.running_mean, self.running_var = Tensor.zeros(sz), Tensor.ones(sz)

    def __call__(self, x):
        assert not Tensor.training                                 # inference: running statistics
        invstd = (self.running_var + self.eps).rsqrt()
        return x.batchnorm(self.weight, self.bias, self.running_mean, invstd)


BatchNorm2d = BatchNorm3d = BatchNorm
