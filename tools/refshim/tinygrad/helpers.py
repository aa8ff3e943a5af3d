"""tinygrad.helpers stand-in (test tooling only): `fetch` does no I/O, it hands the URL to `nn.state.safe_load`."""
import os


def fetch(url, name=None, subdir=None, **kw):
    return url


def getenv(key, default=0):
    return type(default)(os.getenv(key, default))


def prod(xs):
    r = 1
    for x in xs:
        r *= x
    return r


def round_up(n, a):
    return (n + a - 1) // a * a


def partition(lst, fxn):
    a, b = [], []
    for x in lst:
        (a if fxn(x) else b).append(x)
    return a, b
