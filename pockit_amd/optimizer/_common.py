"""What the solver adapters share: a guess becomes the start vector of the NLP, the solver's answer becomes
``Variable`` objects again.

Behaviour kept from the reference's helpers (/root/reference/pockit/optimizer/_common.py:9-63): the accepted shapes
of ``guess`` (one ``Variable`` for a one-phase system without static parameters, else a list with one entry per phase
plus the static values), the ``ValueError`` texts for a half-configured system or a guess of the wrong length, and
the rule that slots whose value the transcription dictates (a FIXED number, or a FUNC of the static parameters:
phasebase.py:830-847) come back holding that value, not whatever the solver left in the dead variable.

Written on the transcription plan: the NLP is ``[phase blocks ... | static parameters]`` with the offsets of
``SystemPlan`` (``l_p``, ``r_p``, ``l_s``, ``r_s``), and the dictated slots are tabulated once per plan
(``dictated_slots``) so that re-applying them is one fancy assignment plus one call per FUNC slot.
"""
from __future__ import annotations

import numpy as np

from ..model import FIXED, FUNC
from ..variable import Variable


def dictated_slots(system):
    """``(index, value)`` arrays of the NLP slots holding a FIXED boundary value / time, and ``[(index, phase, info)]``
    of those that are functions of the static parameters."""
    plan = system.plan
    fixed_at, fixed_to, func = [], [], []
    for k, phase in enumerate(system.p):
        lay, base = phase.layout, int(plan.l_p[k])
        slots = [(int(lay.l_v[i]), phase.info_bc_0[i]) for i in range(phase.n_x)]
        slots += [(int(lay.r_v[i]) - 1, phase.info_bc_f[i]) for i in range(phase.n_x)]
        slots += [(lay.L - 2, phase.info_t_0), (lay.L - 1, phase.info_t_f)]
        for local, info in slots:
            if info.t == FIXED:
                fixed_at.append(base + local)
                fixed_to.append(float(info.v))
            elif info.t == FUNC:
                func.append((base + local, phase, info))
    return np.array(fixed_at, dtype=np.int64), np.array(fixed_to, dtype=np.float64), func


def preprocess(system, guess, optimizer_options=None):
    """-> (start vector of the NLP, whether ``guess`` was a bare ``Variable``, options dict)."""
    if not system.ok:
        raise ValueError("system is not fully configured")
    bare = isinstance(guess, Variable)
    parts = [guess] if bare else list(guess)
    if system.n_s:
        if len(parts) != system.n_p + 1:
            raise ValueError("len(guess) must be equal to the number of phases + 1 (for static variables)")
    elif len(parts) != system.n_p:
        raise ValueError("len(guess) must be equal to the number of phases")
    plan = system.plan
    start = np.zeros(plan.n)
    for k in range(system.n_p):
        start[plan.l_p[k]: plan.r_p[k]] = parts[k].data
    if system.n_s:
        # (a static guess of the wrong length is an error, as in the reference's slice assignment, _common.py:33)
        start[plan.l_s: plan.r_s] = np.array([float(v) for v in parts[-1]], dtype=np.float64)
    return start, bare, ({} if optimizer_options is None else optimizer_options)


def postprocess(system, x, bare):
    """The solver's vector as ``[Variable per phase ... (, static values)]`` (or the one ``Variable``)."""
    plan = system.plan
    sol = np.array(x, dtype=np.float64)          # (a copy: the solver keeps its own vector)
    static = sol[plan.l_s: plan.r_s]
    fixed_at, fixed_to, func = dictated_slots(system)
    sol[fixed_at] = fixed_to
    for at, phase, info in func:
        sol[at] = phase._value_boundary_condition(info, sol[at], static)
    out = [Variable(phase, sol[plan.l_p[k]: plan.r_p[k]]) for k, phase in enumerate(system.p)]
    if system.n_s:
        out.append(static)
    return out[0] if bare else out
