"""Legendre-Gauss-Lobatto namespace of the oracle (mirrors ``pockit.lobatto``)."""
from .phase import Phase as _Phase
from .system import System as _System
from .variable import Variable, constant_guess, linear_guess  # noqa: F401


class Phase(_Phase):
    scheme = "lgl"


class System(_System):
    Phase = Phase
