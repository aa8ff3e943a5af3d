from . import Tensor  # noqa: F401  (clearcam.py imports `tinygrad.tensor.Tensor`)
