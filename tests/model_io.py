"""Test infrastructure: a configured ``pockit_amd`` System <-> a JSON-able description (SymPy expressions as ``srepr``
text, settings as numbers), so that a model built in the build container -- e.g. by running one of the reference's example
PROGRAMS against this package (tests/golden/make_examples.py) -- can be rebuilt on the GPU box through the package's own
modeling API, where neither the reference nor its programs exist.  The description holds what the user handed to the
setters of pockit_amd.model (mathematics and settings), nothing else."""
import importlib

import numpy as np
import sympy as sp


def _strip(name):
    k = name.rfind("^{(")
    return name[:k] if k >= 0 else name


def _num(v):
    v = float(v)
    return "inf" if v == np.inf else ("-inf" if v == -np.inf else v)


def _unnum(v):
    return float(v)


def _bc(raw):
    if raw is None:
        return None
    if isinstance(raw, sp.Expr) and not raw.is_number:
        return {"expr": sp.srepr(raw)}
    return float(raw)


def dump_system(system):
    phases = []
    for p in system.p:
        variables = list(p.x) + list(p.u)
        exprs = [f.expr for f in p._func_phase_constraint]
        lb = [float(v) for v in p._lower_bound_phase_constraint]
        ub = [float(v) for v in p._upper_bound_phase_constraint]
        bang = {(kind, idx) for kind, idx, _, _ in p._bang_bang}
        flags = [("path", j) in bang for j in range(len(exprs))]
        symbols = list(p._symbols)
        for i, lo, hi in p._variable_bounds_phase:            # bare-symbol constraints (they became variable bounds)
            exprs.append(variables[i]); lb.append(lo); ub.append(hi); flags.append(("symbol", symbols.index(variables[i])) in bang)
        for lo, hi in p._time_bounds_phase:
            exprs.append(p.t); lb.append(lo); ub.append(hi); flags.append(("symbol", symbols.index(p.t)) in bang)
        for i, lo, hi in p._static_parameter_bounds_phase:
            exprs.append(system.s[i]); lb.append(lo); ub.append(hi); flags.append(("symbol", symbols.index(system.s[i])) in bang)
        phases.append({
            "state": [_strip(s.name) for s in p.x], "control": [_strip(s.name) for s in p.u],
            "dynamics": [sp.srepr(f.expr) for f in p._func_dynamics],
            "integral": [sp.srepr(f.expr) for f in p._func_integral],
            "constraint": [sp.srepr(sp.sympify(e)) for e in exprs], "lb": [_num(v) for v in lb], "ub": [_num(v) for v in ub],
            "bang_bang": flags,
            "bc": [[_bc(v) for v in p._initial_value], [_bc(v) for v in p._terminal_value], _bc(p._initial_time), _bc(p._terminal_time)],
            "mesh": [float(v) for v in p._mesh], "num_point": [int(v) for v in p._num_point]})
    return {"scheme": "lobatto" if system.p and system.p[0].scheme == "lgl" else "radau",
            "static": [s.name for s in system.s], "simplify": bool(system._simplify), "fastmath": bool(system._fastmath),
            "phases": phases, "objective": sp.srepr(system._expr_objective),
            "system_constraint": [sp.srepr(sp.sympify(e)) for e in system._system_constraint_user],
            "system_lb": [_num(v) for v in system._system_constraint_user_lower_bound],
            "system_ub": [_num(v) for v in system._system_constraint_user_upper_bound]}


def load_system(d, namespace=None):
    """Rebuild the System through the modeling API of ``namespace`` (default: pockit_amd.radau / .lobatto by the scheme;
    the oracle's namespaces take the same calls)."""
    ns = namespace or importlib.import_module(f"pockit_amd.{d['scheme']}")
    system = ns.System(list(d["static"]), simplify=d["simplify"], fastmath=d["fastmath"])
    expr = lambda t: sp.sympify(t)  # noqa: E731  (srepr text; symbols are identified by name)

    def bc(v):
        if v is None:
            return None
        return expr(v["expr"]) if isinstance(v, dict) else float(v)

    phases = []
    for pd in d["phases"]:
        p = system.new_phase(list(pd["state"]), list(pd["control"]))
        p.set_dynamics([expr(t) for t in pd["dynamics"]])
        if pd["integral"]:
            p.set_integral([expr(t) for t in pd["integral"]])
        if pd["constraint"]:
            p.set_phase_constraint([expr(t) for t in pd["constraint"]], [_unnum(v) for v in pd["lb"]], [_unnum(v) for v in pd["ub"]],
                                   list(pd["bang_bang"]) if any(pd["bang_bang"]) else False)
        b0, bf, t0, tf = pd["bc"]
        p.set_boundary_condition([bc(v) for v in b0], [bc(v) for v in bf], bc(t0), bc(tf))
        p.set_discretization(np.asarray(pd["mesh"], dtype=np.float64), np.asarray(pd["num_point"], dtype=np.int64))
        phases.append(p)
    system.set_phase(phases)
    system.set_objective(expr(d["objective"]))
    if d["system_constraint"]:
        system.set_system_constraint([expr(t) for t in d["system_constraint"]], [_unnum(v) for v in d["system_lb"]],
                                     [_unnum(v) for v in d["system_ub"]])
    return system
