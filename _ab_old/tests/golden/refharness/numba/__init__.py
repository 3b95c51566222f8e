"""Identity-decorator stand-in for ``numba`` (test harness only, never shipped in the product).

The reference's generated FastFunc modules begin with ``from numpy import *`` so under this
stub they run as the same array expressions in plain NumPy.
Used only by ``tests/golden/make_golden.py`` inside the build container.
"""
from . import typed  # noqa: F401


def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        fn = args[0]
        fn.py_func = fn
        return fn

    def deco(fn):
        fn.py_func = fn
        return fn

    return deco


njit = _identity
jit = _identity


class _TypeToken:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


float64 = _TypeToken()
int32 = _TypeToken()
int64 = _TypeToken()
boolean = _TypeToken()
