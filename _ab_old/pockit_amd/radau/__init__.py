"""Legendre-Gauss-Radau transcription on MI355X (API of ``pockit.radau``)."""
from ..model import PhaseBase, SystemBase
from ..variable import Variable, constant_guess, linear_guess

__all__ = ["Phase", "System", "Variable", "constant_guess", "linear_guess"]


class Phase(PhaseBase):
    scheme = "lgr"


class System(SystemBase):
    _class_phase = Phase
