"""Minimal solution/guess container and the two guess helpers (oracle).

Restates the parts of /root/reference/pockit/base/variablebase.py the path's callers need:
``Variable`` accessors (:65-140,319-363) and ``constant_guess`` / ``linear_guess`` (:393-470).
Interpolation / adapt (V_x, D_x, adapt) are off the hot path and not restated.
"""
from __future__ import annotations

import numpy as np

from .phase import FIXED


class _Slices:
    def __init__(self, data, l, r):
        self._d, self._l, self._r = data, l, r

    def __getitem__(self, i):
        return self._d[self._l[i]: self._r[i]]

    def __setitem__(self, i, value):
        self._d[self._l[i]: self._r[i]] = value

    def __len__(self):
        return len(self._l)


class Variable:
    def __init__(self, phase, data):
        self._data = data
        nx = phase.n_x
        self.x = _Slices(data, phase.l_v[:nx], phase.r_v[:nx])
        self.u = _Slices(data, phase.l_v[nx:], phase.r_v[nx:])
        self._t_x, self._t_u = phase.t_x, phase.t_u

    data = property(lambda self: self._data)
    t_x = property(lambda self: self._t_x * (self.t_f - self.t_0) + self.t_0)
    t_u = property(lambda self: self._t_u * (self.t_f - self.t_0) + self.t_0)

    @property
    def t_0(self):
        return self._data[-2]

    @t_0.setter
    def t_0(self, v):
        self._data[-2] = v

    @property
    def t_f(self):
        return self._data[-1]

    @t_f.setter
    def t_f(self, v):
        self._data[-1] = v


def _finish_times(v, phase):
    if phase.info_t_0.kind == FIXED:
        v.t_0 = phase.t_0
    else:
        v.t_0 -= 0.5
    if phase.info_t_f.kind == FIXED:
        v.t_f = phase.t_f
    else:
        v.t_f += 0.5
    return v


def constant_guess(phase, value=1.0):
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(value)))
    for i in range(phase.n_x):
        if phase.info_bc_0[i].kind == FIXED:
            v.x[i][0] = phase.bc_0[i]
        if phase.info_bc_f[i].kind == FIXED:
            v.x[i][-1] = phase.bc_f[i]
    return _finish_times(v, phase)


def linear_guess(phase, default=1.0):
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(default)))
    for i in range(phase.n_x):
        f0 = phase.info_bc_0[i].kind == FIXED
        ff = phase.info_bc_f[i].kind == FIXED
        if f0 and ff:
            v.x[i] = v._t_x * (phase.bc_f[i] - phase.bc_0[i]) + phase.bc_0[i]
        elif f0:
            v.x[i] = phase.bc_0[i]
        elif ff:
            v.x[i] = phase.bc_f[i]
    return _finish_times(v, phase)
