"""Legendre-Gauss-Radau namespace of the oracle (mirrors ``pockit.radau``)."""
from .phase import Phase as _Phase
from .system import System as _System
from .variable import Variable, constant_guess, linear_guess  # noqa: F401


class Phase(_Phase):
    scheme = "lgr"


class System(_System):
    Phase = Phase
