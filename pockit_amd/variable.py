"""Guess / solution container for one phase (host-side API plumbing either side of the hot path).

Mirrors the accessors of the reference's ``Variable`` (/root/reference/pockit/base/variablebase.py:92-140,
319-363) and its two guess helpers (:393-470).  Interpolation/adaptation matrices (V_x, D_x, adapt)
are outside the accelerated path and not provided yet.
"""
from __future__ import annotations

import numpy as np

from .model import FIXED


class _BatchView:
    def __init__(self, data, left, right):
        self._data, self._left, self._right = data, left, right

    def __getitem__(self, i):
        return self._data[self._left[i]: self._right[i]]

    def __setitem__(self, i, value):
        self._data[self._left[i]: self._right[i]] = value

    def __len__(self):
        return len(self._left)


class Variable:
    def __init__(self, phase, data):
        self._data = data
        nx = phase.n_x
        self._x = _BatchView(data, phase.l_v[:nx], phase.r_v[:nx])
        self._u = _BatchView(data, phase.l_v[nx:], phase.r_v[nx:])
        self._tx01, self._tu01 = phase.t_x, phase.t_u

    x = property(lambda self: self._x)
    u = property(lambda self: self._u)
    data = property(lambda self: self._data)
    t_x = property(lambda self: self._tx01 * (self.t_f - self.t_0) + self.t_0)
    t_u = property(lambda self: self._tu01 * (self.t_f - self.t_0) + self.t_0)

    @property
    def t_0(self):
        return self._data[-2]

    @t_0.setter
    def t_0(self, value):
        self._data[-2] = value

    @property
    def t_f(self):
        return self._data[-1]

    @t_f.setter
    def t_f(self, value):
        self._data[-1] = value


def _guess_times(v, phase):
    if phase.info_t_0.t == FIXED:
        v.t_0 = phase.t_0
    else:
        v.t_0 -= 0.5
    if phase.info_t_f.t == FIXED:
        v.t_f = phase.t_f
    else:
        v.t_f += 0.5
    return v


def constant_guess(phase, value: float = 1.0) -> Variable:
    """All variables ``value`` except FIXED boundary values/times."""
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(value), dtype=np.float64))
    for i in range(phase.n_x):
        if phase.info_bc_0[i].t == FIXED:
            v.x[i][0] = phase.bc_0[i]
        if phase.info_bc_f[i].t == FIXED:
            v.x[i][-1] = phase.bc_f[i]
    return _guess_times(v, phase)


def linear_guess(phase, default: float = 1.0) -> Variable:
    """States interpolated linearly between FIXED boundary values, ``default`` elsewhere."""
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(default), dtype=np.float64))
    for i in range(phase.n_x):
        fixed0 = phase.info_bc_0[i].t == FIXED
        fixedf = phase.info_bc_f[i].t == FIXED
        if fixed0 and fixedf:
            v.x[i] = v._tx01 * (phase.bc_f[i] - phase.bc_0[i]) + phase.bc_0[i]
        elif fixed0:
            v.x[i] = phase.bc_0[i]
        elif fixedf:
            v.x[i] = phase.bc_f[i]
    return _guess_times(v, phase)
