"""Reference-side binding: a configured ``pockit`` system -> the MI355X evaluator's plan.

``plan_from_reference_system(system)`` is what INTEGRATION.md section 2 hands to ``Evaluator``: a pockit maintainer
who keeps the reference's own ``System`` / ``Phase`` objects (``pockit.base.systembase.SystemBase``,
``pockit.base.phasebase.PhaseBase``) swaps only the five callbacks.  The function reads what the reference's setters
stored -- the raw SymPy expressions and settings, NOT its compiled functions or index arrays -- and replays them on
this package's modeling API, symbol by symbol:

  PhaseBase._expr_dynamics / _expr_integral / _expr_phase_constraint      phasebase.py:264,295,349-366
  PhaseBase._variable_bounds_phase / _time_bounds_phase / _static_parameter_bounds_phase   phasebase.py:345-362
  PhaseBase._initial_value / _terminal_value / _initial_time / _terminal_time              phasebase.py:474-477
  PhaseBase._mesh / _num_point                                                              phasebase.py:575-577
  SystemBase._expr_objective, _system_constraint_user (+ bounds)                            systembase.py:198,245-247

The transcription compiler then rebuilds layout and triplet order from scratch; that they equal the reference's is what
the golden vectors pin (tests/test_adapter_reference.py compares converted plans with tests/golden in the build
container; nothing of the reference ships or is imported by this module).
The bang-bang flags of phase constraints (they feed the switch-point refinement) are recovered from the SymPy source of the
reference's scaled constraint functions.  Not carried over: FastFunc cache directories.
"""
from __future__ import annotations

import importlib

import numpy as np
import sympy as sp


def _strip(name: str) -> str:
    """``speed^{(0)}`` -> ``speed`` (the phase identifier suffix is re-attached by ``new_phase``)."""
    k = name.rfind("^{(")
    return name[:k] if k >= 0 else name


def _bang_bang_flags(rp, conv, exprs, lb, ub):
    """Which of the rebuilt phase constraints the reference treats as bang-bang.  The reference keeps no flags, only one
    compiled function ``(c - lb) / (ub - lb)`` per flagged constraint, in the user's order (phasebase.py:388-412); its
    SymPy source survives in ``FastFunc._function`` with the arguments renamed (fastfunc.py:140-149).  A rebuilt
    constraint is bang-bang when its own scaled form equals one of those sources."""
    funcs = list(getattr(rp, "_func_bang_bang_control", []))
    if not funcs:
        return False
    scaled = []
    for f in funcs:
        back = dict(zip(f._args, rp._symbols))
        scaled.append(conv(sp.sympify(f._function).xreplace(back)))
    flags = []
    for e, lo, hi in zip(exprs, lb, ub):
        hit = False
        if np.isfinite(lo) and np.isfinite(hi) and hi > lo:
            mine = (sp.sympify(e) - lo) / (hi - lo)
            for k, g in enumerate(scaled):
                if g is not None and (mine == g or sp.expand(mine - g) == 0 or sp.simplify(mine - g) == 0):
                    hit, scaled[k] = True, None      # (every compiled function marks one constraint)
                    break
        flags.append(hit)
    if any(g is not None for g in scaled):
        raise ValueError("a bang-bang constraint of the reference phase could not be matched to a phase constraint")
    return flags


def system_from_reference(ref):
    """A ``pockit_amd`` System mirroring the configured reference system ``ref`` (radau or lobatto by its module)."""
    scheme = "lobatto" if ".lobatto" in type(ref).__module__ else "radau"
    ns = importlib.import_module(f"pockit_amd.{scheme}")
    system = ns.System([str(n) for n in ref._name_static_parameter], simplify=ref._simplify, fastmath=ref._fastmath)
    smap = dict(zip(ref._symbol_static_parameter, system.s))
    gmap = dict(smap)                                          # + integral symbols, for the system-level expressions
    phases = []
    for rp in ref._phase:
        p = system.new_phase([_strip(n) for n in rp._name_state], [_strip(n) for n in rp._name_control])
        m = dict(smap)
        m.update(zip(rp._symbol_state, p.x))
        m.update(zip(rp._symbol_control, p.u))
        m[rp._symbol_time] = p.t

        def conv(e, m=m):
            return sp.sympify(e).xreplace(m)

        def bc(v, m=m):
            if v is None:
                return None
            if isinstance(v, sp.Expr) and not v.is_number:
                return v.xreplace(m)
            return float(v)

        p.set_dynamics([conv(e) for e in rp._expr_dynamics])
        if getattr(rp, "_integral_set", False):
            p.set_integral([conv(e) for e in rp._expr_integral])
            gmap.update(zip(rp._symbol_integral, p.I))
        if getattr(rp, "_phase_constraint_set", False):
            exprs = [conv(e) for e in rp._expr_phase_constraint]
            lb = [float(v) for v in np.asarray(rp._lower_bound_phase_constraint, dtype=np.float64)]
            ub = [float(v) for v in np.asarray(rp._upper_bound_phase_constraint, dtype=np.float64)]
            variables = list(p.x) + list(p.u)
            for i, lo, hi in rp._variable_bounds_phase:        # bare-symbol constraints became variable bounds
                exprs.append(variables[i]); lb.append(lo); ub.append(hi)
            for lo, hi in rp._time_bounds_phase:
                exprs.append(p.t); lb.append(lo); ub.append(hi)
            for i, lo, hi in rp._static_parameter_bounds_phase:
                exprs.append(system.s[i]); lb.append(lo); ub.append(hi)
            p.set_phase_constraint(exprs, lb, ub, _bang_bang_flags(rp, conv, exprs, lb, ub))
        p.set_boundary_condition([bc(v) for v in rp._initial_value], [bc(v) for v in rp._terminal_value],
                                 bc(rp._initial_time), bc(rp._terminal_time))
        p.set_discretization(np.asarray(rp._mesh, dtype=np.float64), np.asarray(rp._num_point, dtype=np.int64))
        phases.append(p)
    system.set_phase(phases)
    system.set_objective(sp.sympify(ref._expr_objective).xreplace(gmap))
    if getattr(ref, "_system_constraint_set", False):
        system.set_system_constraint([sp.sympify(e).xreplace(gmap) for e in ref._system_constraint_user],
                                     list(ref._system_constraint_user_lower_bound),
                                     list(ref._system_constraint_user_upper_bound))
    return system


def plan_from_reference_system(ref):
    """The ``SystemPlan`` (layout, triplet structure in the reference's order, value expressions) of a configured
    reference system: ``Evaluator.checked(plan_from_reference_system(system))`` serves its five callbacks on the GPU (the fused kernel verified against
    the stand-alone ones at set-up)."""
    return system_from_reference(ref).plan
