"""Guess packing / solution unpacking shared by the solver adapters.

Mirrors /root/reference/pockit/optimizer/_common.py:9-63 (argument checks, x0 layout
[phase data ... | static], re-application of FIXED/FUNC boundary values to the returned x)."""
from __future__ import annotations

import numpy as np

from ..variable import Variable


def preprocess(system, guess, optimizer_options=None):
    if not system.ok:
        raise ValueError("system is not fully configured")
    if optimizer_options is None:
        optimizer_options = {}
    guess_is_variable = isinstance(guess, Variable)
    if guess_is_variable:
        guess = [guess]
    if not system.n_s and len(guess) != system.n_p:
        raise ValueError("len(guess) must be equal to the number of phases")
    elif system.n_s and len(guess) != system.n_p + 1:
        raise ValueError("len(guess) must be equal to the number of phases + 1 (for static variables)")
    x_0 = np.zeros(system.L)
    for i in range(system.n_p):
        x_0[system.l_p[i]: system.r_p[i]] = guess[i].data
    if system.n_s > 0:
        x_0[system.l_s: system.r_s] = np.array(list(guess[-1]), dtype=np.float64)
    return x_0, guess_is_variable, optimizer_options


def postprocess(system, x, guess_is_variable):
    x = np.array(x, dtype=np.float64)
    s = x[system.l_s: system.r_s]
    result = []
    for i, p in enumerate(system.p):
        x_ = x[system.l_p[i]: system.r_p[i]]
        for j in range(p.n_x):
            x_[p.l_v[j]] = p._value_boundary_condition(p.info_bc_0[j], x_[p.l_v[j]], s)
            x_[p.r_v[j] - 1] = p._value_boundary_condition(p.info_bc_f[j], x_[p.r_v[j] - 1], s)
        x_[-2] = p._value_boundary_condition(p.info_t_0, x_[-2], s)
        x_[-1] = p._value_boundary_condition(p.info_t_f, x_[-1], s)
        result.append(Variable(p, x_))
    if system.n_s > 0:
        result.append(s)
    return result[0] if guess_is_variable else result
