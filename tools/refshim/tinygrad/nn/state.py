"""tinygrad.nn.state stand-in (test tooling only).  `get_state_dict` walks an object tree the way tinygrad documents it
(attributes of objects, items of lists / tuples / dicts, dotted key names); `load_state_dict` is strict about missing keys
and shapes, so loading a synthetic checkpoint through the reference's own constructors checks its key layout.
`safe_load` does no file I/O: it returns the state dict registered under the file name by the fixture generator."""
import os
from collections import OrderedDict

import numpy as np

from .. import Tensor

INJECTED = {}          # file name (basename of the URL / path the reference asks for) -> {key: ndarray}
LOADED = []            # (file name, unused keys) per load_state_dict call, for the generator's report


def inject(filename, state_dict):
    INJECTED[filename] = state_dict


def safe_load(fn):
    name = os.path.basename(str(fn))
    if name not in INJECTED:
        raise FileNotFoundError(f"refshim: no state dict injected for {name}")
    d = {k: Tensor(np.asarray(v)) for k, v in INJECTED[name].items()}
    d["__refshim_name__"] = name
    return d


def safe_save(tensors, fn, metadata=None):
    raise RuntimeError("refshim: safe_save is not available")


def get_state_dict(obj, prefix="", tensor_type=Tensor):
    if isinstance(obj, tensor_type):
        return {prefix.strip("."): obj}
    if hasattr(obj, "_asdict"):
        return get_state_dict(obj._asdict(), prefix, tensor_type)
    if isinstance(obj, OrderedDict):
        return get_state_dict(dict(obj), prefix, tensor_type)
    if hasattr(obj, "__dict__"):
        return get_state_dict(obj.__dict__, prefix, tensor_type)
    out = {}
    if isinstance(obj, (list, tuple)):
        for i, x in enumerate(obj):
            out.update(get_state_dict(x, f"{prefix}{i}.", tensor_type))
    elif isinstance(obj, dict):
        for k, v in obj.items():
            out.update(get_state_dict(v, f"{prefix}{k}.", tensor_type))
    return out


def load_state_dict(model, state_dict, strict=True, verbose=False, consume=False):
    name = state_dict.pop("__refshim_name__", "?") if isinstance(state_dict, dict) else "?"
    model_sd = get_state_dict(model)
    for k, v in model_sd.items():
        if k not in state_dict:
            if k.endswith("num_batches_tracked"):                  # a counter no inference path reads; the synthetic checkpoints omit it
                continue
            if strict:
                raise KeyError(f"refshim load_state_dict: checkpoint {name} has no '{k}'")
            continue
        src = state_dict[k]
        if tuple(src.shape) != tuple(v.shape):
            raise ValueError(f"refshim load_state_dict: shape of '{k}': checkpoint {tuple(src.shape)} vs model {tuple(v.shape)}")
        v.replace(src.cast(v.dtype) if src.dtype is not v.dtype else src)
    LOADED.append((name, sorted(k for k in state_dict if k not in model_sd)))
    return model
