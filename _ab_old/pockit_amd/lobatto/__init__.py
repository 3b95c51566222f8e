"""Legendre-Gauss-Lobatto transcription on MI355X (API of ``pockit.lobatto``)."""
from ..model import PhaseBase, SystemBase
from ..variable import Variable, constant_guess, linear_guess

__all__ = ["Phase", "System", "Variable", "constant_guess", "linear_guess"]


class Phase(PhaseBase):
    scheme = "lgl"


class System(SystemBase):
    _class_phase = Phase
