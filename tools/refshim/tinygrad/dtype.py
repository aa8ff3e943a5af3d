"""dtype names of the tinygrad stand-in (see tinygrad/__init__.py: test tooling only)."""
import torch


class DType:
    def __init__(self, name, tt):
        self.name, self.tt = name, tt

    def __repr__(self):
        return f"dtypes.{self.name}"


class dtypes:
    float32 = float = DType("float", torch.float32)
    float16 = half = DType("half", torch.float16)
    bfloat16 = DType("bfloat16", torch.bfloat16)
    float64 = double = DType("double", torch.float64)
    int8 = char = DType("char", torch.int8)
    uint8 = uchar = DType("uchar", torch.uint8)
    int16 = short = DType("short", torch.int16)
    int32 = int = DType("int", torch.int32)
    int64 = long = DType("long", torch.int64)
    bool = DType("bool", torch.bool)
    default_float = float32
    default_int = int32


_BY_NAME = {"float32": dtypes.float32, "float": dtypes.float32, "float16": dtypes.float16, "half": dtypes.float16,
            "uint8": dtypes.uint8, "int8": dtypes.int8, "int16": dtypes.int16, "int32": dtypes.int32, "int": dtypes.int32,
            "int64": dtypes.int64, "bool": dtypes.bool, "bfloat16": dtypes.bfloat16, "float64": dtypes.float64}
_BY_TORCH = {d.tt: d for d in _BY_NAME.values()}


def to_torch(dt):
    if isinstance(dt, str):
        dt = _BY_NAME[dt]
    return dt.tt if isinstance(dt, DType) else dt


def from_torch(tt):
    return _BY_TORCH[tt]
