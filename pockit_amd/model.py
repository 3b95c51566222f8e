"""Modeling API of the MI355X evaluator: ``System`` / ``Phase`` with pockit's names and semantics.

The classes here only *describe* the optimal-control problem (SymPy expressions, boundary kinds,
mesh) and validate input exactly where the reference does; nothing is evaluated on the host.
``System`` then hands the description to the transcription compiler / HIP code generator
(pockit_amd/transcription.py, codegen.py) and serves the cyipopt ``problem_obj`` protocol from
the GPU (pockit_amd/evaluator.py).

Reference interface mirrored (same method names, argument meaning, chaining, ValueErrors):
  * /root/reference/pockit/base/phasebase.py:41-630    Phase construction and setters
  * /root/reference/pockit/base/systembase.py:53-255   System construction and setters
  * /root/reference/pockit/base/systembase.py:602-835  the seven NLP callbacks (+ hessian_o/_c)
Symbol naming (``name^{(id)}``, ``t^{(id)}``, ``I_k^{(id)}``) follows phasebase.py:85-117,301-304.
"""
from __future__ import annotations

import weakref
from typing import Iterable, Optional

import numpy as np
import sympy as sp

from .layout import MeshLayout
from .symbolic import SparseFunc

FREE, FIXED, FUNC = 0, 1, 2


class BcInfo:
    """Boundary quantity: ``None`` -> FREE, number -> FIXED, SymPy expression of s -> FUNC."""

    __slots__ = ("t", "v", "raw")

    def __init__(self, raw, static_symbols, simplify):
        self.raw = raw
        if raw is None:
            self.t, self.v = FREE, None
        elif isinstance(raw, float):
            self.t, self.v = FIXED, raw
        elif isinstance(raw, sp.Expr):
            self.t, self.v = FUNC, SparseFunc(raw, static_symbols, simplify)
        else:
            raise ValueError("boundary condition must be None, number or sp.Expr")


def _names(spec, prefix, identifier, what):
    if isinstance(spec, int):
        return [f"{prefix}_{i}^{{({identifier})}}" for i in range(spec)]
    if isinstance(spec, list):
        if "t" in spec:
            raise ValueError(f'Symbol "t" is reserved for time. Use a different name for {what} variables')
        return [s + f"^{{({identifier})}}" for s in spec]
    raise ValueError(f"{what} must be int or list of str")


class PhaseBase:
    """One phase: states x, controls u, time t, static parameters s, integrals I."""

    scheme: str = ""  # "lgr" or "lgl"

    def __init__(self, identifier, state, control, symbol_static_parameter, simplify=False, fastmath=False):
        self._identifier = identifier
        self._symbol_state = [sp.Symbol(n) for n in _names(state, "x", identifier, "state")]
        self._symbol_control = [sp.Symbol(n) for n in _names(control, "u", identifier, "control")]
        self._symbol_time = sp.Symbol(f"t^{{({identifier})}}")
        self._symbol_static_parameter = list(symbol_static_parameter)
        self._symbols = (self._symbol_state + self._symbol_control + [self._symbol_time]
                         + self._symbol_static_parameter)
        self._simplify, self._fastmath = simplify, fastmath
        self._dynamics_set = self._boundary_condition_set = self._discretization_set = False
        self._version = 0
        self.set_integral([])
        self.set_phase_constraint([], [], [])

    def _changed(self):
        self._version += 1
        self._discontinuous_check_passed = False      # reference: phasebase.py:229-231,827-828
        return self

    # ------------------------------------------------------------------ setters
    def set_dynamics(self, dynamics, *, cache: Optional[str] = None):
        if len(dynamics) != self.n_x:
            raise ValueError("the number of dynamics must be equal to the number of state variables")
        self._func_dynamics = [SparseFunc(d, self._symbols, self._simplify) for d in dynamics]
        self._dynamics_set = True
        return self._changed()

    def set_integral(self, integral, *, cache: Optional[str] = None):
        self._func_integral = [SparseFunc(e, self._symbols, self._simplify) for e in integral]
        self._symbol_integral = [sp.Symbol(f"I_{i}^{{({self._identifier})}}") for i in range(len(integral))]
        return self._changed()

    def set_phase_constraint(self, phase_constraint, lower_bound, upper_bound,
                             bang_bang_control=False, *, cache: Optional[str] = None):
        phase_constraint, lower_bound, upper_bound = list(phase_constraint), list(lower_bound), list(upper_bound)
        if not len(phase_constraint) == len(lower_bound) == len(upper_bound):
            raise ValueError("phase_constraint, lower_bound and upper_bound must have the same length")
        self._variable_bounds_phase, self._time_bounds_phase, self._static_parameter_bounds_phase = [], [], []
        exprs, lo, hi = [], [], []
        for c, lb, ub in zip(phase_constraint, lower_bound, upper_bound):
            if c.is_symbol:
                i = self._symbols.index(c)
                if i < self.n:
                    self._variable_bounds_phase.append((i, lb, ub))
                elif i == self.n:
                    self._time_bounds_phase.append((lb, ub))
                else:
                    self._static_parameter_bounds_phase.append((i - self.n - 1, lb, ub))
            else:
                exprs.append(sp.sympify(c))
                lo.append(lb)
                hi.append(ub)
        if isinstance(bang_bang_control, bool):
            flags = [bang_bang_control] * len(phase_constraint)
        else:
            flags = list(bang_bang_control)
        # bang-bang constraints, scaled to [0, 1] at check time (reference: phasebase.py:388-412): where the value
        # of each one comes from -- a path-constraint row block or a variable / time / static-parameter slot
        self._bang_bang = []
        n_path = 0
        for c, lb, ub, bb in zip(phase_constraint, lower_bound, upper_bound, flags):
            if bb:
                if np.isinf(lb) or np.isinf(ub):
                    raise ValueError("lower_bound and upper_bound must be finite for bang-bang control constraint")
                if ub <= lb + 1e-10:
                    raise ValueError(
                        "lower_bound must be strictly less than upper_bound for bang-bang control constraint")
                self._bang_bang.append(("symbol", self._symbols.index(c), lb, ub) if c.is_symbol
                                       else ("path", n_path, lb, ub))
            n_path += 0 if c.is_symbol else 1
        self._func_phase_constraint = [SparseFunc(e, self._symbols, self._simplify) for e in exprs]
        self._lower_bound_phase_constraint = np.array(lo, dtype=np.float64)
        self._upper_bound_phase_constraint = np.array(hi, dtype=np.float64)
        return self._changed()

    def set_boundary_condition(self, initial_value, terminal_value, initial_time, terminal_time,
                               *, cache: Optional[str] = None):
        if not len(initial_value) == len(terminal_value) == self.n_x:
            raise ValueError(
                "initial_value, terminal_value must have the same length as number of state variables")

        def num(v):
            return float(v) if isinstance(v, int) else v

        self._initial_value = [num(v) for v in initial_value]
        self._terminal_value = [num(v) for v in terminal_value]
        self._initial_time, self._terminal_time = num(initial_time), num(terminal_time)
        mk = lambda raw: BcInfo(raw, self._symbol_static_parameter, self._simplify)  # noqa: E731
        self.info_bc_0 = [mk(v) for v in self._initial_value]
        self.info_bc_f = [mk(v) for v in self._terminal_value]
        self.info_t_0, self.info_t_f = mk(self._initial_time), mk(self._terminal_time)
        self._boundary_condition_set = True
        return self._changed()

    def set_discretization(self, mesh, num_point):
        if isinstance(mesh, (int, np.integer)):
            if mesh < 1:
                raise ValueError("mesh must contain at least one interval")
            mesh_new = np.linspace(0, 1, int(mesh) + 1, endpoint=True)
        else:
            mesh_new = np.array(list(mesh), dtype=np.float64)
            if mesh_new.ndim != 1 or len(mesh_new) < 2:
                raise ValueError("mesh must contain at least two points")
            if not np.all(np.isfinite(mesh_new)):
                raise ValueError("mesh points must be finite")
            if np.any(np.diff(mesh_new) <= 0):
                raise ValueError("mesh points must be strictly increasing")
            mesh_new = (mesh_new - mesh_new[0]) / (mesh_new[-1] - mesh_new[0])
        n_int = len(mesh_new) - 1
        if isinstance(num_point, (int, np.integer)):
            k_new = np.full(n_int, int(num_point), dtype=np.int64)
        else:
            vals = np.array(list(num_point))
            if vals.ndim != 1:
                raise ValueError("num_point must be a one-dimensional iterable")
            if not np.issubdtype(vals.dtype, np.integer):
                raise ValueError("num_point entries must be integers")
            k_new = vals.astype(np.int64)
        if len(k_new) != n_int:
            raise ValueError("num_point must have the same length as mesh intervals (= len(mesh) - 1)")
        k_min = 2 if self.scheme == "lgl" else 1
        if np.any(k_new < k_min):
            raise ValueError(f"num_point entries must be at least {k_min}")
        if np.any(k_new > np.iinfo(np.int32).max):
            raise ValueError("num_point entries are too large")
        layout = MeshLayout(self.scheme, mesh_new, k_new.astype(np.int32), self.n_x, self.n_u)
        # commit only after everything validated (atomic update)
        self._mesh, self._num_point, self._num_interval = mesh_new, k_new.astype(np.int32), n_int
        self.layout = layout
        self._discretization_set = True
        return self._changed()

    # ------------------------------------------------------------------ mesh error check / refinement
    # (reference: phasebase.py:1374-1437 check_continuous, 1522-1617 refine_continuous).  The error data come from
    # the GPU evaluator of the system the phase belongs to (pk_err); the decision logic is pockit_amd/refine.py.
    def _owner(self):
        system = self._system() if getattr(self, "_system", None) is not None else None
        if system is None:
            raise ValueError("the phase must be part of a System (System.set_phase) before its mesh error can be "
                             "evaluated: the evaluator lives at system level")
        return system

    def _substitute_boundary(self, data, s):
        """Write the FIXED / FUNC boundary values and times into ``data`` in place (the reference does this to the
        caller's array whenever it evaluates a phase, phasebase.py:839-851; ``refine_continuous`` then adapts the
        substituted values)."""
        lay = self.layout
        for i in range(self.n_x):
            data[lay.l_v[i]] = self._value_boundary_condition(self.info_bc_0[i], data[lay.l_v[i]], s)
            data[lay.r_v[i] - 1] = self._value_boundary_condition(self.info_bc_f[i], data[lay.r_v[i] - 1], s)
        data[-2] = self._value_boundary_condition(self.info_t_0, data[-2], s)
        data[-1] = self._value_boundary_condition(self.info_t_f, data[-1], s)

    def _error_data(self, variable, static_parameter):
        if self.n_s and static_parameter is None:
            raise ValueError("phase has static parameters, but the value of static parameters is not given")
        self._substitute_boundary(variable.data, [] if static_parameter is None else list(static_parameter))
        system = self._owner()
        k = next(i for i, p in enumerate(system.p) if p is self)
        plan = system.plan
        x = np.zeros(plan.n)
        for i, p in enumerate(system.p):        # other phases: any finite point on their own time axis
            x[plan.r_p[i] - 1] = 1.0
        x[plan.l_p[k]: plan.r_p[k]] = variable.data
        if self.n_s:
            x[plan.l_s: plan.r_s] = np.asarray(list(static_parameter), dtype=np.float64)
        return system.evaluator.mesh_error(x)[k]

    def check_continuous(self, variable, static_parameter=None, absolute_tolerance_continuous=1.0e-8,
                         relative_tolerance_continuous=1.0e-8, tolerance_mesh=1.0e-4) -> bool:
        from . import refine

        T, I = self._error_data(variable, static_parameter)
        return bool(np.all(refine.interval_ok(self.layout, T, I, absolute_tolerance_continuous,
                                              relative_tolerance_continuous, tolerance_mesh)))

    def _refine_from(self, T, I, atol, rtol, num_point_min, num_point_max, mesh_length_min, mesh_length_max):
        from . import refine

        ok = refine.interval_ok(self.layout, T, I, atol, rtol, mesh_length_min)
        if np.all(ok):
            return
        mesh, num_point = refine.refined_discretization(self.layout, T, I, ok, rtol, num_point_min, num_point_max,
                                                        mesh_length_min, mesh_length_max)
        passed = self._discontinuous_check_passed       # survives a continuous refinement (phasebase.py:1615-1617)
        self.set_discretization(mesh, num_point)
        self._discontinuous_check_passed = passed

    # bang-bang (discontinuous) check: reference phasebase.py:1368-1400,1439-1474
    n_b = property(lambda self: len(self._bang_bang))

    def _bang_bang_values(self, data, s, g_path):
        """(n_b, L_m): every bang-bang constraint scaled to [0, 1] at the collocation nodes.  ``g_path``: this
        phase's path-constraint values (n_c, L_m) as the constraints callback returned them."""
        lay = self.layout
        out = np.empty((self.n_b, lay.L_m))
        for b, (kind, idx, lb, ub) in enumerate(self._bang_bang):
            if kind == "path":
                v = g_path[idx]
            elif idx < self.n:
                v = data[lay.l_v[idx]: lay.l_v[idx] + lay.L_m]
            elif idx == self.n:
                v = (lay.tau - 0.5) * (data[-1] - data[-2]) + (data[-1] + data[-2]) / 2
            else:
                v = np.full(lay.L_m, s[idx - self.n - 1])
            out[b] = (v - lb) / (ub - lb)
        return out

    def _discontinuous_ok(self, f_bb, dtol, mtol):
        lay = self.layout
        ok = np.ones(lay.N, dtype=bool)
        for j in range(lay.N):
            if lay.width[j] < mtol:
                continue
            part = f_bb[:, lay.lm[j]: lay.rm[j]]
            ok[j] = bool(np.all(np.all(part < dtol, axis=1) | np.all(part > 1 - dtol, axis=1)))
        return ok

    def _require_radau(self):
        if self.scheme == "lgl":
            raise NotImplementedError("Lobatto nodes cannot approximate discontinuous functions precisely.")

    def _path_values(self, variable, static_parameter):
        """This phase's path-constraint values at the nodes, from the system's constraints callback (GPU)."""
        if self.n_s and static_parameter is None:
            raise ValueError("phase has static parameters, but the value of static parameters is not given")
        s = [] if static_parameter is None else [float(v) for v in static_parameter]
        self._substitute_boundary(variable.data, s)
        if not any(kind == "path" for kind, *_ in self._bang_bang):
            return None, s
        system = self._owner()
        k = next(i for i, p in enumerate(system.p) if p is self)
        plan = system.plan
        x = np.zeros(plan.n)
        for i in range(len(system.p)):
            x[plan.r_p[i] - 1] = 1.0
        x[plan.l_p[k]: plan.r_p[k]] = variable.data
        x[plan.l_s: plan.r_s] = s
        g = system.evaluator.constraints(x)
        L_m = self.layout.L_m
        return g[plan.path_off[k]: plan.path_off[k] + self.n_c * L_m].reshape(self.n_c, L_m), s

    def check_discontinuous(self, variable, static_parameter=None, tolerance_discontinuous=1e-3,
                            tolerance_mesh=1e-4) -> bool:
        self._require_radau()
        g_path, s = self._path_values(variable, static_parameter)
        f_bb = self._bang_bang_values(variable.data, s, g_path)
        passed = bool(np.all(self._discontinuous_ok(f_bb, tolerance_discontinuous, tolerance_mesh)))
        if passed:
            self._discontinuous_check_passed = True
        return passed

    def check(self, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
              relative_tolerance_continuous=1e-8, tolerance_discontinuous=1e-3, tolerance_mesh=1e-4) -> bool:
        if self.scheme == "lgr" and not self._discontinuous_check_passed and not self.check_discontinuous(
                variable, static_parameter, tolerance_discontinuous, tolerance_mesh):
            return False
        return self.check_continuous(variable, static_parameter, absolute_tolerance_continuous,
                                     relative_tolerance_continuous, tolerance_mesh)

    def refine_discontinuous(self, variable, static_parameter=None, tolerance_discontinuous=1e-3, num_point_min=6,
                             num_point_max=12, mesh_length_min=1e-3, mesh_length_max=1.0) -> None:
        """Move / add mesh points to the switching times of the bang-bang constraints (reference:
        phasebase.py:1619-1868; logic in pockit_amd/refine.py).  Call ``System.update()`` afterwards."""
        from . import refine

        self._require_radau()
        if self.check_discontinuous(variable, static_parameter, tolerance_discontinuous, mesh_length_min):
            return
        g_path, s = self._path_values(variable, static_parameter)
        f_bb = self._bang_bang_values(variable.data, s, g_path)
        mesh, num_point = refine.switch_point_discretization(self.layout, f_bb, tolerance_discontinuous, num_point_min,
                                                             num_point_max, mesh_length_min, mesh_length_max)
        self.set_discretization(mesh, num_point)

    def refine(self, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
               relative_tolerance_continuous=1e-8, tolerance_discontinuous=1e-3, num_point_min=6, num_point_max=12,
               mesh_length_min=1e-3, mesh_length_max=1.0) -> None:
        """At most one refinement: for the bang-bang error if that check fails, else for the continuous error."""
        if self.scheme == "lgr" and not self._discontinuous_check_passed and not self.check_discontinuous(
                variable, static_parameter, tolerance_discontinuous, mesh_length_min):
            self.refine_discontinuous(variable, static_parameter, tolerance_discontinuous, num_point_min,
                                      num_point_max, mesh_length_min, mesh_length_max)
        else:
            self.refine_continuous(variable, static_parameter, absolute_tolerance_continuous,
                                   relative_tolerance_continuous, num_point_min, num_point_max, mesh_length_min,
                                   mesh_length_max)

    def refine_continuous(self, variable, static_parameter=None, absolute_tolerance_continuous=1.0e-8,
                          relative_tolerance_continuous=1.0e-8, num_point_min=6, num_point_max=12,
                          mesh_length_min=1.0e-3, mesh_length_max=1.0) -> None:
        """Adjust mesh and interpolation degrees in place (call ``System.update()`` afterwards, as with the
        reference)."""
        T, I = self._error_data(variable, static_parameter)
        self._refine_from(T, I, absolute_tolerance_continuous, relative_tolerance_continuous, num_point_min,
                          num_point_max, mesh_length_min, mesh_length_max)

    # ------------------------------------------------------------------ read-only views
    n_x = property(lambda self: len(self._symbol_state))
    n_u = property(lambda self: len(self._symbol_control))
    n = property(lambda self: len(self._symbol_state) + len(self._symbol_control))
    n_s = property(lambda self: len(self._symbol_static_parameter))
    n_d = property(lambda self: len(self._symbol_state))
    n_I = property(lambda self: len(self._func_integral))
    n_c = property(lambda self: len(self._func_phase_constraint))
    x = property(lambda self: self._symbol_state)
    u = property(lambda self: self._symbol_control)
    t = property(lambda self: self._symbol_time)
    s = property(lambda self: self._symbol_static_parameter)
    I = property(lambda self: self._symbol_integral)  # noqa: E741
    F_d = property(lambda self: self._func_dynamics)
    F_I = property(lambda self: self._func_integral)
    F_c = property(lambda self: self._func_phase_constraint)
    c_lb = property(lambda self: self._lower_bound_phase_constraint)
    c_ub = property(lambda self: self._upper_bound_phase_constraint)
    s_b = property(lambda self: self._static_parameter_bounds_phase)
    bc_0 = property(lambda self: self._initial_value)
    bc_f = property(lambda self: self._terminal_value)
    t_0 = property(lambda self: self._initial_time)
    t_f = property(lambda self: self._terminal_time)
    N = property(lambda self: self._num_interval)
    ok = property(lambda self: self._dynamics_set and self._boundary_condition_set and self._discretization_set)
    # mesh layout views
    l_v = property(lambda self: self.layout.l_v)
    r_v = property(lambda self: self.layout.r_v)
    l_d = property(lambda self: self.layout.l_d)
    r_d = property(lambda self: self.layout.r_d)
    l_m = property(lambda self: self.layout.lm)
    r_m = property(lambda self: self.layout.rm)
    L_m = property(lambda self: self.layout.L_m)
    t_m = property(lambda self: self.layout.tau)
    w_m = property(lambda self: self.layout.w)
    t_x = property(lambda self: self.layout.t_x)
    t_u = property(lambda self: self.layout.tau)
    # per-interval windows on the state / control time axes and the one-more-point ("aug") views the refinement
    # uses (reference: radau/discretization.py:505-521,593-613, lobatto/discretization.py:431-441)
    l_x = property(lambda self: self.layout.lm)
    r_x = property(lambda self: self.layout.rm + (1 if self.scheme == "lgr" else 0))
    l_u = property(lambda self: self.layout.lm)
    r_u = property(lambda self: self.layout.rm)

    @property
    def l_m_aug(self):
        k = self.layout.K + 1
        step = k if self.scheme == "lgr" else k - 1
        return np.concatenate(([0], np.cumsum(step[:-1])))

    @property
    def r_m_aug(self):
        return self.l_m_aug + self.layout.K + 1

    @property
    def L_m_aug(self):
        return int(self.r_m_aug[-1])

    @property
    def t_m_aug(self):
        """Positions in [0, 1] of the nodes with one more point per interval (LGL: shared ends listed once)."""
        from . import collocation

        lay, lgr = self.layout, self.scheme == "lgr"
        nodes = collocation.lgr_nodes_weights if lgr else collocation.lgl_nodes_weights
        parts = []
        for j in range(lay.N):
            xa = lay.mesh[j] + (nodes(int(lay.K[j]) + 1)[0] + 1.0) * 0.5 * lay.width[j]
            parts.append(xa if lgr or j == lay.N - 1 else xa[:-1])
        return np.concatenate(parts)

    @property
    def w_aug(self):
        from . import collocation

        nodes = collocation.lgr_nodes_weights if self.scheme == "lgr" else collocation.lgl_nodes_weights
        return [nodes(int(k))[1] for k in self.layout.K]

    @property
    def P(self):
        """K -> matrix turning values at the K nodes of an interval into monomial coefficients (highest power
        first) of their interpolant on [-1, 1]."""
        from . import collocation

        nodes = collocation.lgr_nodes_weights if self.scheme == "lgr" else collocation.lgl_nodes_weights
        return lambda K: np.linalg.inv(np.vander(nodes(int(K))[0]))

    L_x = property(lambda self: int(self.layout.r_v[self.n_x - 1]))
    L_xu = property(lambda self: int(self.layout.r_v[-1]))
    L = property(lambda self: self.layout.L)

    @property
    def v_lb(self):
        return self._variable_bounds()[0]

    @property
    def v_ub(self):
        return self._variable_bounds()[1]

    def _variable_bounds(self):
        """Bare-symbol path constraints become bounds over the whole trajectory
        (reference: phasebase.py:632-659)."""
        lo = np.full(self.L, -np.inf, dtype=np.float64)
        hi = np.full(self.L, np.inf, dtype=np.float64)
        for i, lb, ub in self._variable_bounds_phase:
            sl = slice(self.l_v[i], self.r_v[i])
            lo[sl] = np.maximum(lo[sl], lb)
            hi[sl] = np.minimum(hi[sl], ub)
        for lb, ub in self._time_bounds_phase:
            lo[-2:] = np.maximum(lo[-2:], lb)
            hi[-2:] = np.minimum(hi[-2:], ub)
        return lo, hi

    @staticmethod
    def _value_boundary_condition(info, x, s):
        """Host-side helper used by the solver adapters' post-processing
        (reference: phasebase.py:830-837, optimizer/_common.py:50-56)."""
        if info.t == FREE:
            return x
        if info.t == FIXED:
            return info.v
        fn = sp.lambdify(info.v.args, info.v.expr, modules="math")
        return float(fn(*[float(v) for v in s]))


class SystemBase:
    """A complete multi-phase problem; implements the cyipopt ``problem_obj`` protocol on the GPU."""

    _class_phase = PhaseBase

    def __init__(self, static_parameter, simplify=False, fastmath=False):
        if isinstance(static_parameter, int):
            names = [f"s_{i}" for i in range(static_parameter)]
        elif isinstance(static_parameter, list):
            names = static_parameter
        else:
            raise ValueError("static_parameter must be int or list of str")
        self._symbol_static_parameter = [sp.Symbol(n) for n in names]
        self._simplify, self._fastmath = simplify, fastmath
        self._identifier_phase = 0
        self._phase = []
        self._phase_set = self._objective_set = self._system_constraint_set = False
        self._evaluator = None
        self._built_for = None
        self._hessian_layout = "reference"
        self._jacobian_layout = "reference"
        self.set_phase([])
        self.set_system_constraint([], np.array([]), np.array([]))

    # ------------------------------------------------------------------ modeling API
    def new_phase(self, state, control):
        self._identifier_phase += 1
        return self._class_phase(self._identifier_phase - 1, state, control, self._symbol_static_parameter,
                                 self._simplify, self._fastmath)

    def set_phase(self, phase):
        for i, p in enumerate(phase):
            if not p.ok:
                raise ValueError(
                    f"Dynamics, boundary conditions, or discretization scheme of phase {i} are not fully set")
        self._phase = list(phase)
        for p in self._phase:
            p._system = weakref.ref(self)
        self._phase_set = True
        return self._invalidate()

    def set_objective(self, objective, *, cache: Optional[str] = None):
        self._expr_objective = sp.sympify(objective)
        self._objective_set = True
        return self._invalidate()

    def set_system_constraint(self, system_constraint, lower_bound: Iterable[float],
                              upper_bound: Iterable[float], *, cache: Optional[str] = None):
        lower_bound, upper_bound = list(lower_bound), list(upper_bound)
        if not len(system_constraint) == len(lower_bound) == len(upper_bound):
            raise ValueError("system_constraint, lower_bound and upper_bound must have the same length")
        self._system_constraint_user = list(system_constraint)
        self._system_constraint_user_lower_bound = lower_bound
        self._system_constraint_user_upper_bound = upper_bound
        self._system_constraint_set = True
        return self._invalidate()

    def set_hessian_layout(self, layout: str):
        """``"reference"`` (default): the reference's triplet list, duplicates included (drop-in).
        ``"compact"``: one triplet per distinct (row, col) of every node -- 10-20x fewer values for IPOPT to
        receive and assemble; ``hessianstructure()`` / ``hessian()`` switch together, the matrices are equal."""
        if layout not in ("reference", "compact"):
            raise ValueError('layout must be "reference" or "compact"')
        self._hessian_layout = layout
        return self

    @property
    def writable_results(self):
        """False (default): ``jacobian()`` returns a READ-ONLY array where the landing block keeps the x-independent entries of J
        from iterate to iterate (they never cross PCIe again; ``.copy()`` gives a private array).  True: every callback returns
        a writable array, as the reference does, and the block has those entries filled in again before it is reused (one
        host pass over them per iterate).  ``POCKIT_AMD_WRITABLE_RESULTS=1`` sets the default."""
        if getattr(self, "_writable_results", None) is not None:
            return self._writable_results
        return self._evaluator.writable_results if self._evaluator is not None else False

    @writable_results.setter
    def writable_results(self, value):
        self._writable_results = bool(value)
        if self._evaluator is not None:
            self._evaluator.writable_results = self._writable_results

    def set_jacobian_layout(self, layout: str):
        """``"reference"`` (default): the reference's triplet list (drop-in).  ``"compact"``: derivative entries of the
        dynamics whose column is the same on every node (t_0, t_f, static parameters) are contracted with the
        integration block -- one value per defect row instead of one per nonzero of the integration matrix
        (phasebase.py:885-887,1120-1124 emit K per row); ``jacobianstructure()`` / ``jacobian()`` switch together, the
        matrices are equal."""
        if layout not in ("reference", "compact"):
            raise ValueError('layout must be "reference" or "compact"')
        self._jacobian_layout = layout
        if self._evaluator is not None:
            self._evaluator.set_jacobian_layout(layout == "compact")
        return self

    def update(self) -> None:
        """Re-transcribe after changing any phase (e.g. a new mesh)."""
        self._invalidate()

    def _invalidate(self):
        self._plan = None
        if self._evaluator is not None:
            self._evaluator.close()
        self._evaluator = None
        return self

    # ------------------------------------------------------------------ plan / evaluator (lazy)
    def _stamp(self):
        return tuple([p._version for p in self._phase])

    @property
    def plan(self):
        if self._plan is not None and self._built_for == self._stamp():
            return self._plan
        from .transcription import SystemPlan

        if self._plan is None or self._built_for != self._stamp():
            if self._evaluator is not None:
                self._evaluator.close()
                self._evaluator = None
            self._plan = SystemPlan(self)
            self._built_for = self._stamp()
        return self._plan

    @property
    def evaluator(self):
        """The GPU evaluator; built (code generation + hipcc + upload) on first use.
        Raises RuntimeError when the HIP library or a GPU is missing -- there is no CPU path."""
        ev = self._evaluator       # (a solver calls this five times per iterate: the common case first)
        if ev is not None and self._plan is not None and self._built_for == self._stamp():
            return ev
        from .evaluator import Evaluator

        plan = self.plan
        if self._evaluator is None:
            self._evaluator = Evaluator.checked(plan)      # (fused kernel verified against the stand-alone ones)
            self._evaluator.set_jacobian_layout(self._jacobian_layout == "compact")
            if getattr(self, "_writable_results", None) is not None:
                self._evaluator.writable_results = self._writable_results
        return self._evaluator

    # ------------------------------------------------------------------ layout views (host, no GPU)
    n_s = property(lambda self: len(self._symbol_static_parameter))
    s = property(lambda self: self._symbol_static_parameter)
    n_p = property(lambda self: len(self._phase))
    N = property(lambda self: len(self._phase))
    p = property(lambda self: self._phase)
    ok = property(lambda self: self._phase_set and self._objective_set and self._system_constraint_set)
    l_p = property(lambda self: self.plan.l_p)
    r_p = property(lambda self: self.plan.r_p)
    l_s = property(lambda self: self.plan.l_s)
    r_s = property(lambda self: self.plan.r_s)
    L = property(lambda self: self.plan.n)
    n_c = property(lambda self: self.plan.n_sys)
    v_lb = property(lambda self: self.plan.v_lb)
    v_ub = property(lambda self: self.plan.v_ub)
    c_lb = property(lambda self: self.plan.c_lb)
    c_ub = property(lambda self: self.plan.c_ub)

    # ------------------------------------------------------------------ cyipopt problem_obj protocol
    def objective(self, x):
        return self.evaluator.objective(x)

    def gradient(self, x):
        return self.evaluator.gradient(x)

    def constraints(self, x):
        return self.evaluator.constraints(x)

    def jacobianstructure(self):
        if self._jacobian_layout == "compact":
            self.plan.jacc  # noqa: B018  (builds the compact plan)
            return self.plan.jacc_row, self.plan.jacc_col
        return self.plan.jac_row, self.plan.jac_col

    def jacobian(self, x):
        return self.evaluator.jacobian(x)

    def hessianstructure(self):
        if self._hessian_layout == "compact":
            self.plan.hessc  # noqa: B018  (builds the compact plan)
            return self.plan.hessc_row, self.plan.hessc_col
        return self.plan.hess_row, self.plan.hess_col

    def hessian(self, x, lagrange, obj_factor):
        if self._hessian_layout == "compact":
            return self.evaluator.hessian_compact(x, lagrange, obj_factor)
        return self.evaluator.hessian(x, lagrange, obj_factor)

    # split Hessians used by the SciPy adapter (reference: systembase.py:726-809)
    def hessianstructure_o(self):
        if self._hessian_layout == "compact":
            return self.hessianstructure()
        n = self.plan.nnz_H_obj
        return self.plan.hess_row[:n], self.plan.hess_col[:n]

    def hessian_o(self, x):
        m = len(self.plan.c_lb)
        if self._hessian_layout == "compact":       # same pattern for both parts; the values split by sigma / lambda
            return self.evaluator.hessian_compact(x, np.zeros(m), 1.0)
        return self.evaluator.hessian(x, np.zeros(m), 1.0)[: self.plan.nnz_H_obj]

    def hessianstructure_c(self):
        if self._hessian_layout == "compact":
            return self.hessianstructure()
        n = self.plan.nnz_H_obj
        return self.plan.hess_row[n:], self.plan.hess_col[n:]

    def hessian_c(self, x, fct_c):
        if self._hessian_layout == "compact":
            return self.evaluator.hessian_compact(x, fct_c, 0.0)
        return self.evaluator.hessian(x, fct_c, 0.0)[self.plan.nnz_H_obj:]

    # ------------------------------------------------------------------ CSR hand-off
    def jacobian_csr(self, x):
        """The constraint Jacobian as ``scipy.sparse.csr_array`` (values gathered into CSR order on the GPU)."""
        ev = self.evaluator
        return ev.csr_map("jac").to_scipy(ev.jacobian_csr(x))

    def hessian_csr(self, x, lagrange, obj_factor):
        """The lower triangle of the Hessian of the Lagrangian as ``scipy.sparse.csr_array`` (repeated triplets of
        the reference layout summed on the GPU)."""
        ev = self.evaluator
        return ev.csr_map("hess").to_scipy(ev.hessian_csr(x, lagrange, obj_factor))

    # ------------------------------------------------------------------ mesh error check / refinement
    # (reference: systembase.py:837-889 check_continuous, 982-1069 refine_continuous)
    def _split_value(self, value):
        from .variable import Variable

        if not self.ok:
            raise ValueError("system is not fully configured")
        single = isinstance(value, Variable)
        if single:
            value = [value]
        if not self.n_s and len(value) != self.n_p:
            raise ValueError("len(value) must be equal to the number of phases")
        if self.n_s and len(value) != self.n_p + 1:
            raise ValueError("len(value) must be equal to the number of phases + 1 (for static variables)")
        return list(value), single

    def _mesh_error(self, value):
        plan = self.plan
        x = np.empty(plan.n)
        s = [float(v) for v in value[-1]] if self.n_s else []
        for k in range(self.n_p):
            self._phase[k]._substitute_boundary(value[k].data, s)
            x[plan.l_p[k]: plan.r_p[k]] = value[k].data
        if self.n_s:
            x[plan.l_s: plan.r_s] = np.array(list(value[-1]), dtype=np.float64)
        return self.evaluator.mesh_error(x)

    def check_continuous(self, value, absolute_tolerance_continuous=1.0e-8, relative_tolerance_continuous=1.0e-8,
                         tolerance_mesh=1.0e-4) -> bool:
        from . import refine

        value, _ = self._split_value(value)
        data = self._mesh_error(value)
        return all(bool(np.all(refine.interval_ok(p.layout, T, I, absolute_tolerance_continuous,
                                                  relative_tolerance_continuous, tolerance_mesh)))
                   for p, (T, I) in zip(self._phase, data))

    def refine_continuous(self, value, absolute_tolerance_continuous=1.0e-8, relative_tolerance_continuous=1.0e-8,
                          num_point_min=6, num_point_max=12, mesh_length_min=1.0e-3, mesh_length_max=1.0):
        """One hp-refinement sweep over all phases from a single error-estimation launch; returns the values
        interpolated onto the new discretization (the input itself when every interval already passes)."""
        from . import refine

        original = value
        value, single = self._split_value(value)
        data = self._mesh_error(value)
        if all(bool(np.all(refine.interval_ok(p.layout, T, I, absolute_tolerance_continuous,
                                              relative_tolerance_continuous, mesh_length_min)))
               for p, (T, I) in zip(self._phase, data)):
            return original
        adapted = []
        for p, v, (T, I) in zip(self._phase, value, data):
            p._refine_from(T, I, absolute_tolerance_continuous, relative_tolerance_continuous, num_point_min,
                           num_point_max, mesh_length_min, mesh_length_max)
            adapted.append(v.adapt(p))
        self.update()
        if single:
            return adapted[0]
        return adapted + value[self.n_p:]

    def _static_of(self, value):
        return np.array(list(value[-1]), dtype=np.float64) if self.n_s else None

    def check_discontinuous(self, value, tolerance_discontinuous=1.0e-3, tolerance_mesh=1.0e-4) -> bool:
        """Bang-bang check of every phase (reference: systembase.py:891-939); Radau only, as in the reference."""
        value, _ = self._split_value(value)
        s = self._static_of(value)
        return bool(np.all([p.check_discontinuous(v, s, tolerance_discontinuous, tolerance_mesh)
                            for p, v in zip(self._phase, value)]))

    def check(self, value, absolute_tolerance_continuous=1.0e-8, relative_tolerance_continuous=1.0e-8,
              tolerance_discontinuous=1.0e-3, tolerance_mesh=1.0e-4) -> bool:
        """Continuous and (Radau) discontinuous error check (reference: systembase.py:941-980,
        lobatto/system.py:31-58)."""
        lgl = any(p.scheme == "lgl" for p in self._phase)
        return self.check_continuous(value, absolute_tolerance_continuous, relative_tolerance_continuous,
                                     tolerance_mesh) and (lgl or self.check_discontinuous(
                                         value, tolerance_discontinuous, tolerance_mesh))

    def refine_discontinuous(self, value, tolerance_discontinuous=1.0e-3, num_point_min=6, num_point_max=12,
                             mesh_length_min=1.0e-3, mesh_length_max=1.0):
        original = value
        if self.check_discontinuous(value, tolerance_discontinuous, mesh_length_min):
            return original
        value, single = self._split_value(value)
        s = self._static_of(value)
        adapted = []
        for p, v in zip(self._phase, value):
            p.refine_discontinuous(v, s, tolerance_discontinuous, num_point_min, num_point_max, mesh_length_min,
                                   mesh_length_max)
            adapted.append(v.adapt(p))
        self.update()
        return adapted[0] if single else adapted + value[self.n_p:]

    def refine(self, value, absolute_tolerance_continuous=1.0e-8, relative_tolerance_continuous=1.0e-8,
               tolerance_discontinuous=1.0e-3, num_point_min=6, num_point_max=12, mesh_length_min=1.0e-3,
               mesh_length_max=1.0):
        """One refinement sweep (reference: systembase.py:1134-1212): per phase the bang-bang refinement if that
        check fails, else the continuous one; returns the values on the new discretization."""
        original = value
        if self.check(value, absolute_tolerance_continuous, relative_tolerance_continuous, tolerance_discontinuous,
                      mesh_length_min):
            return original
        value, single = self._split_value(value)
        s = self._static_of(value)
        adapted = []
        for p, v in zip(self._phase, value):
            p.refine(v, s, absolute_tolerance_continuous, relative_tolerance_continuous, tolerance_discontinuous,
                     num_point_min, num_point_max, mesh_length_min, mesh_length_max)
            adapted.append(v.adapt(p))
        self.update()
        return adapted[0] if single else adapted + value[self.n_p:]
