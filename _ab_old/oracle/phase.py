"""One phase of the transcribed optimal-control problem (oracle, NumPy).

Restates the callback half of /root/reference/pockit/base/phasebase.py:
  * modeling setters ................................... phasebase.py:41-630
  * variable bounds .................................... phasebase.py:632-659
  * derivative-node graph (front / middle / back) ...... phasebase.py:125-194,580-626,661-825
  * middle-stage vector, boundary substitution ......... phasebase.py:830-852
  * J/H index builders ................................. phasebase.py:854-995
  * values, gradients, Hessians ........................ phasebase.py:997-1337
Unlike the reference, ``x`` is never modified in place (a substituted copy is evaluated) and the
dependency tracking (autoupdate.py) is replaced by one lazy ``prepare()``.
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .chain import DNode, eval_gradient, eval_hessian, link_gradient, link_hessian
from .mesh import Mesh
from .symfunc import SymFunc

FREE, FIXED, FUNC = 0, 1, 2


class Boundary:
    """kind/value of one boundary quantity: None -> FREE, number -> FIXED, Expr(s) -> FUNC."""

    def __init__(self, raw, static_symbols, compile_args):
        self.raw = raw
        if raw is None:
            self.kind, self.v = FREE, None
        elif isinstance(raw, float):
            self.kind, self.v = FIXED, raw
        elif isinstance(raw, sp.Expr):
            self.kind, self.v = FUNC, SymFunc(raw, static_symbols, *compile_args)
        else:
            raise ValueError("boundary condition must be None, number or sp.Expr")

    def value(self, current, s):
        if self.kind == FREE:
            return current
        if self.kind == FIXED:
            return self.v
        return self.v.F(s, 1)[0]


def _cat(parts, dtype):
    return np.concatenate(parts).astype(dtype, copy=False) if parts else np.array([], dtype=dtype)


class Phase:
    scheme = None  # "lgr" / "lgl", bound by the radau / lobatto namespaces

    def __init__(self, identifier, state, control, symbol_static_parameter, simplify=False, fastmath=False):
        def names(spec, prefix, what):
            if isinstance(spec, int):
                return [f"{prefix}_{i}^{{({identifier})}}" for i in range(spec)]
            if isinstance(spec, list):
                if "t" in spec:
                    raise ValueError(f'Symbol "t" is reserved for time. Use a different name for {what} variables')
                return [nm + f"^{{({identifier})}}" for nm in spec]
            raise ValueError(f"{what} must be int or list of str")

        self._id = identifier
        self.x = [sp.Symbol(nm) for nm in names(state, "x", "state")]
        self.u = [sp.Symbol(nm) for nm in names(control, "u", "control")]
        self.t = sp.Symbol(f"t^{{({identifier})}}")
        self.s = list(symbol_static_parameter)
        self.n_x, self.n_u, self.n_s = len(self.x), len(self.u), len(self.s)
        self.n = self.n_x + self.n_u
        self._symbols = self.x + self.u + [self.t] + self.s
        self._compile = (simplify, fastmath)
        self._have = set()
        self._ready = False
        self.set_integral([])
        self.set_phase_constraint([], [], [])

    # ------------------------------------------------------------------ modeling API
    def _touch(self, what):
        self._have.add(what)
        self._ready = False
        return self

    def set_dynamics(self, dynamics, *, cache=None):
        if len(dynamics) != self.n_x:
            raise ValueError("the number of dynamics must be equal to the number of state variables")
        self.F_d = [SymFunc(sp.sympify(d), self._symbols, *self._compile) for d in dynamics]
        return self._touch("dynamics")

    def set_integral(self, integral, *, cache=None):
        self.F_I = [SymFunc(sp.sympify(e), self._symbols, *self._compile) for e in integral]
        self.n_I = len(self.F_I)
        self.I = [sp.Symbol(f"I_{i}^{{({self._id})}}") for i in range(self.n_I)]
        return self._touch("integral")

    def set_phase_constraint(self, phase_constraint, lower_bound, upper_bound, bang_bang_control=False, *, cache=None):
        cons, lo, hi = list(phase_constraint), list(lower_bound), list(upper_bound)
        if not len(cons) == len(lo) == len(hi):
            raise ValueError("phase_constraint, lower_bound and upper_bound must have the same length")
        self._var_bounds, self._time_bounds, self.s_b = [], [], []
        exprs, elo, ehi = [], [], []
        for c, lb, ub in zip(cons, lo, hi):
            if c.is_symbol:                     # a bare symbol is a variable bound, not a row
                k = self._symbols.index(c)
                if k < self.n:
                    self._var_bounds.append((k, lb, ub))
                elif k == self.n:
                    self._time_bounds.append((lb, ub))
                else:
                    self.s_b.append((k - self.n - 1, lb, ub))
            else:
                exprs.append(sp.sympify(c))
                elo.append(lb)
                ehi.append(ub)
        self.F_c = [SymFunc(e, self._symbols, *self._compile) for e in exprs]
        self.n_c = len(self.F_c)
        self.c_lb = np.array(elo, dtype=np.float64)
        self.c_ub = np.array(ehi, dtype=np.float64)
        return self._touch("constraint")

    def set_boundary_condition(self, initial_value, terminal_value, initial_time, terminal_time, *, cache=None):
        if not len(initial_value) == len(terminal_value) == self.n_x:
            raise ValueError("initial_value, terminal_value must have the same length as number of state variables")

        def num(v):
            return float(v) if isinstance(v, int) else v

        self.bc_0 = [num(v) for v in initial_value]
        self.bc_f = [num(v) for v in terminal_value]
        self.t_0, self.t_f = num(initial_time), num(terminal_time)
        mk = lambda raw: Boundary(raw, self.s, self._compile)  # noqa: E731
        self.info_bc_0 = [mk(v) for v in self.bc_0]
        self.info_bc_f = [mk(v) for v in self.bc_f]
        self.info_t_0, self.info_t_f = mk(self.t_0), mk(self.t_f)
        return self._touch("boundary")

    def set_discretization(self, mesh, num_point):
        if isinstance(mesh, int):
            if mesh < 1:
                raise ValueError("mesh must contain at least one interval")
            pts = np.linspace(0, 1, mesh + 1, endpoint=True)
        else:
            pts = np.array(list(mesh), dtype=np.float64)
            if pts.ndim != 1 or len(pts) < 2:
                raise ValueError("mesh must contain at least two points")
            if not np.all(np.isfinite(pts)):
                raise ValueError("mesh points must be finite")
            if np.any(np.diff(pts) <= 0):
                raise ValueError("mesh points must be strictly increasing")
            pts = (pts - pts[0]) / (pts[-1] - pts[0])
        if isinstance(num_point, int):
            K = np.full(len(pts) - 1, num_point, dtype=np.int64)
        else:
            K = np.array(list(num_point))
            if K.ndim != 1:
                raise ValueError("num_point must be a one-dimensional iterable")
            if not np.issubdtype(K.dtype, np.integer):
                raise ValueError("num_point entries must be integers")
            K = K.astype(np.int64)
        if len(K) != len(pts) - 1:
            raise ValueError("num_point must have the same length as mesh intervals (= len(mesh) - 1)")
        kmin = 2 if self.scheme == "lgl" else 1
        if np.any(K < kmin):
            raise ValueError(f"num_point entries must be at least {kmin}")
        d = Mesh(self.scheme, pts, K.astype(np.int32), self.n_x, self.n_u)
        self._mesh, self._num_point, self.N, self.d = pts, K.astype(np.int32), len(K), d
        for nm in ("l_v", "r_v", "l_d", "r_d", "l_m", "r_m", "L_m", "t_m", "w_m", "t_x", "t_u",
                   "l_x", "r_x", "l_u", "r_u"):
            setattr(self, nm, getattr(d, nm))
        self.L = int(d.r_v[-1]) + 2
        return self._touch("mesh")

    @property
    def ok(self):
        return {"dynamics", "boundary", "mesh"} <= self._have

    # ------------------------------------------------------------------ mesh error check / refinement
    def check_continuous(self, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
                         relative_tolerance_continuous=1e-8, tolerance_mesh=1e-4):
        from . import refine

        return refine.check_continuous(self, variable, static_parameter, absolute_tolerance_continuous,
                                       relative_tolerance_continuous, tolerance_mesh)

    def refine_continuous(self, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
                          relative_tolerance_continuous=1e-8, num_point_min=6, num_point_max=12,
                          mesh_length_min=1e-3, mesh_length_max=1.0):
        from . import refine

        refine.refine_continuous(self, variable, static_parameter, absolute_tolerance_continuous,
                                 relative_tolerance_continuous, num_point_min, num_point_max, mesh_length_min,
                                 mesh_length_max)

    # ------------------------------------------------------------------ structure (setup)
    def prepare(self):
        if self._ready:
            return
        d, nx, nu, ns = self.d, self.n_x, self.n_u, self.n_s
        pm = d.part_mstage
        self._pm = pm

        # variable bounds (phasebase.py:632-659)
        lo = np.full(self.L, -np.inf)
        hi = np.full(self.L, np.inf)
        for k, lb, ub in self._var_bounds:
            sl = slice(self.l_v[k], self.r_v[k])
            lo[sl] = np.maximum(lo[sl], lb)
            hi[sl] = np.minimum(hi[sl], ub)
        for lb, ub in self._time_bounds:
            lo[-2:] = np.maximum(lo[-2:], lb)
            hi[-2:] = np.minimum(hi[-2:], ub)
        self.v_lb, self.v_ub = lo, hi

        # leaf / boundary nodes
        static = [DNode().leaf(-ns + i) for i in range(ns)]

        def boundary_node(info, slot):
            nd = DNode()
            if info.kind == FREE:
                nd.leaf(slot)
            elif info.kind == FUNC:
                nd.args = static
                nd.local(info.v)
            return nd

        x_f = [boundary_node(self.info_bc_0[i], self.l_v[i]) for i in range(nx)]
        x_b = [boundary_node(self.info_bc_f[i], self.r_v[i] - 1) for i in range(nx)]
        ps, pc = d.part_state, d.part_control
        x_m = [DNode(ps.L_m).leaf(np.arange(self.l_v[i] + ps.l_m, self.l_v[i] + ps.r_m)) for i in range(nx)]
        u_m = [DNode(pc.L_m).leaf(np.arange(self.l_v[nx + i] + pc.l_m, self.l_v[nx + i] + pc.r_m))
               for i in range(nu)]
        u_f = [DNode().leaf(self.l_v[nx + i]) if pm.f else DNode() for i in range(nu)]
        u_b = [DNode().leaf(self.r_v[nx + i] - 1) if pm.b else DNode() for i in range(nu)]
        t_f_ = boundary_node(self.info_t_0, self.L - 2)
        t_b_ = boundary_node(self.info_t_f, self.L - 1)
        t_m_ = DNode(pm.L_m)
        t_m_.args = [t_f_, t_b_]
        t_m_.lg_arg = np.array([0, 1], dtype=np.int32)
        t_m_.lg = np.array([1.0 - self.t_m[pm.m], self.t_m[pm.m]])
        dt_ = DNode()
        dt_.args = [t_f_, t_b_]
        dt_.lg_arg = np.array([0, 1], dtype=np.int32)
        dt_.lg = np.array([[-1.0], [1.0]])
        s_m = []
        for i in range(ns):
            nd = DNode(pm.L_m)
            nd.args = [static[i]]
            nd.lg_arg = np.array([0], dtype=np.int32)
            nd.lg = np.ones((1, pm.L_m))
            s_m.append(nd)
        self._n_static, self._n_xf, self._n_xb = static, x_f, x_b
        self._n_tf, self._n_tb, self._n_dt = t_f_, t_b_, dt_
        self._basic = static + x_m + u_f + u_m + u_b + s_m + x_f + x_b + [t_f_, t_b_, t_m_, dt_]
        link_gradient(self._basic)
        link_hessian(self._basic, "phase")
        self._args = {
            "f": x_f + u_f + [t_f_] + static,
            "m": x_m + u_m + [t_m_] + s_m,
            "b": x_b + u_b + [t_b_] + static,
        }

        self._dyn = self._function_nodes(self.F_d, scaled=True)
        self._int = self._function_nodes(self.F_I, scaled=True)
        self._con = self._function_nodes(self.F_c, scaled=False)
        self._index_dynamic()
        self._index_path()
        self._ready = True

    def _function_nodes(self, funcs, scaled):
        """Per function: raw nodes at front/middle/back and (optionally) their dt-scaled twins."""
        pm = self._pm
        raw = {w: [] for w in "fmb"}
        for fn in funcs:
            for w in "fmb":
                nd = DNode(pm.L_m if w == "m" else 1).local(fn)
                nd.args = self._args[w]
                raw[w].append(nd)
        order = raw["f"] + raw["m"] + raw["b"]
        link_gradient(order)
        link_hessian(order, "phase")
        out = {"raw": raw, "all": list(order)}
        if scaled:
            sc = {w: [] for w in "fmb"}
            for w in "fmb":
                for r in raw[w]:
                    nd = DNode(r.n)
                    nd.args = [r, self._n_dt]
                    nd.lg_arg = np.array([0, 1], dtype=np.int32)
                    nd.lh_r = np.array([1], dtype=np.int32)
                    nd.lh_c = np.array([0], dtype=np.int32)
                    nd.lh = np.array([[1.0]])
                    sc[w].append(nd)
                link_gradient(sc[w])
                link_hessian(sc[w], "phase")
            out["scaled"] = sc
            out["all"] += sc["f"] + sc["m"] + sc["b"]
        return out

    def _index_dynamic(self):
        d, pm = self.d, self._pm
        T, I = d.T_coo, d.I_coo
        jr, jc, hr, hc = [], [], [], []
        for i in range(self.n_x):                                  # translation part
            if self._n_xf[i].Gi:
                gi = np.concatenate(self._n_xf[i].Gi)
                jr.append(self.l_d[i] + np.repeat(T.f.row, len(gi)))
                jc.append(np.tile(gi, T.f.nnz))
            jr.append(self.l_d[i] + T.m.row)
            jc.append(self.l_v[i] + T.m.col)
            if self._n_xb[i].Gi:
                gi = np.concatenate(self._n_xb[i].Gi)
                jr.append(self.l_d[i] + np.repeat(T.b.row, len(gi)))
                jc.append(np.tile(gi, T.b.nnz))
        sc = self._dyn["scaled"]
        for i in range(self.n_x):                                  # integration part
            if pm.f and sc["f"][i].Gi:
                gi = np.concatenate(sc["f"][i].Gi)
                jr.append(self.l_d[i] + np.repeat(I.f.row, len(gi)))
                jc.append(np.tile(gi, I.f.nnz))
            for gi in sc["m"][i].Gi:
                jr.append(self.l_d[i] + I.m.row)
                jc.append(gi[I.m.col - pm.l_m])
            if pm.b and sc["b"][i].Gi:
                gi = np.concatenate(sc["b"][i].Gi)
                jr.append(self.l_d[i] + np.repeat(I.b.row, len(gi)))
                jc.append(np.tile(gi, I.b.nnz))
        self.jac_dyn_row, self.jac_dyn_col = _cat(jr, np.int64), _cat(jc, np.int64)

        for i in range(self.n_x):
            for nd, part in ((self._n_xf[i], T.f), (self._n_xb[i], T.b)):
                if nd.Hr:
                    hr.append(np.tile(np.concatenate(nd.Hr), part.nnz))
                    hc.append(np.tile(np.concatenate(nd.Hc), part.nnz))
        for i in range(self.n_x):
            if pm.f and sc["f"][i].Hr:
                hr.append(np.tile(np.concatenate(sc["f"][i].Hr), I.f.nnz))
                hc.append(np.tile(np.concatenate(sc["f"][i].Hc), I.f.nnz))
            for r_, c_ in zip(sc["m"][i].Hr, sc["m"][i].Hc):
                hr.append(r_[I.m.col - pm.l_m])
                hc.append(c_[I.m.col - pm.l_m])
            if pm.b and sc["b"][i].Hr:
                hr.append(np.tile(np.concatenate(sc["b"][i].Hr), I.b.nnz))
                hc.append(np.tile(np.concatenate(sc["b"][i].Hc), I.b.nnz))
        self.hess_dyn_row, self.hess_dyn_col = _cat(hr, np.int64), _cat(hc, np.int64)

    def _index_path(self):
        pm = self._pm
        raw = self._con["raw"]
        jr, jc, hr, hc = [], [], [], []
        base = 0
        for j in range(self.n_c):
            if pm.f:
                for gi in raw["f"][j].Gi:
                    jr.append(np.array([base]))
                    jc.append(gi)
            for gi in raw["m"][j].Gi:
                jr.append(np.arange(base + pm.l_m, base + pm.r_m))
                jc.append(gi)
            if pm.b:
                for gi in raw["b"][j].Gi:
                    jr.append(np.array([base + self.L_m - 1]))
                    jc.append(gi)
            base += self.L_m
        for j in range(self.n_c):
            for w, on in (("f", pm.f), ("m", True), ("b", pm.b)):
                if on:
                    hr.extend(raw[w][j].Hr)
                    hc.extend(raw[w][j].Hc)
        self.jac_path_row, self.jac_path_col = _cat(jr, np.int64), _cat(jc, np.int64)
        self.hess_path_row, self.hess_path_col = _cat(hr, np.int64), _cat(hc, np.int64)

    # ------------------------------------------------------------------ values (every callback)
    def mstage(self, x, s):
        """Boundary-substituted copy of x, middle-stage vector vb and dt (phasebase.py:839-852)."""
        x = np.array(x, dtype=np.float64, copy=True)
        for i in range(self.n_x):
            x[self.l_v[i]] = self.info_bc_0[i].value(x[self.l_v[i]], s)
            x[self.r_v[i] - 1] = self.info_bc_f[i].value(x[self.r_v[i] - 1], s)
        x[-2] = self.info_t_0.value(x[-2], s)
        x[-1] = self.info_t_f.value(x[-1], s)
        dt = x[-1] - x[-2]
        tm = (x[-1] + x[-2]) / 2
        t_nodes = (self.t_m - 0.5) * dt + tm
        vb = np.concatenate([self.d.to_mstage(x[:-2]), t_nodes, np.repeat(s, self.L_m)])
        return x, vb, dt

    def _load_boundary_locals(self, s, hess):
        for info, nd in ([*zip(self.info_bc_0, self._n_xf), *zip(self.info_bc_f, self._n_xb),
                          (self.info_t_0, self._n_tf), (self.info_t_f, self._n_tb)]):
            if info.kind == FUNC:
                nd.lg = info.v.G(s, 1)
                if hess:
                    nd.lh = info.v.H(s, 1)
        eval_gradient(self._basic)
        if hess:
            eval_hessian(self._basic, "phase")

    def _load_function_locals(self, group, funcs, which, vb, dt, hess):
        """Evaluate F/G(/H) on all nodes and hand the front/middle/back columns to the nodes."""
        pm = self._pm
        cols = {"f": slice(0, 1), "m": pm.m, "b": slice(self.L_m - 1, self.L_m)}
        live = {"f": pm.f, "m": True, "b": pm.b}
        for k, fn in enumerate(funcs):
            if which is not None and not which[k]:
                continue
            g = fn.G(vb, self.L_m)
            h = fn.H(vb, self.L_m) if hess else None
            f = fn.F(vb, self.L_m) if "scaled" in group else None
            for w in "fmb":
                if not live[w]:
                    continue
                raw = group["raw"][w][k]
                raw.lg = g[:, cols[w]]
                if hess:
                    raw.lh = h[:, cols[w]]
                if f is not None:
                    fv = f[cols[w]]
                    group["scaled"][w][k].lg = np.array([np.full_like(fv, dt), fv])
        eval_gradient(group["all"])
        if hess:
            eval_hessian(group["all"], "phase")

    # integrals -----------------------------------------------------------------------------
    def value_integral(self, which, x, s):
        _, vb, dt = self.mstage(x, s)
        return np.array([self.F_I[k].F(vb, self.L_m).dot(self.w_m) * dt if flag else 0.0
                         for k, flag in enumerate(which)], dtype=np.float64)

    def deriv_integral(self, which, x, s, hess):
        self._load_boundary_locals(s, hess)
        _, vb, dt = self.mstage(x, s)
        self._load_function_locals(self._int, self.F_I, which, vb, dt, hess)

    # defects -------------------------------------------------------------------------------
    def value_dynamic(self, x, s):
        xs, vb, dt = self.mstage(x, s)
        rows = [self.d.T_v.dot(xs[self.l_v[i]: self.r_v[i]]) - self.d.I_m.dot(self.F_d[i].F(vb, self.L_m)) * dt
                for i in range(self.n_x)]
        return np.concatenate(rows)

    def jac_dynamic(self, x, s):
        d, pm = self.d, self._pm
        T, I = d.T_coo, d.I_coo
        out = []
        for i in range(self.n_x):
            if self._n_xf[i].Gv:
                out.append(np.kron(T.f.data, np.concatenate(self._n_xf[i].Gv)))
            out.append(T.m.data)
            if self._n_xb[i].Gv:
                out.append(np.kron(T.b.data, np.concatenate(self._n_xb[i].Gv)))
        _, vb, dt = self.mstage(x, s)
        self._load_function_locals(self._dyn, self.F_d, None, vb, dt, hess=False)
        sc = self._dyn["scaled"]
        for i in range(self.n_x):
            if pm.f and sc["f"][i].Gv:
                out.append(-np.kron(I.f.data, np.concatenate(sc["f"][i].Gv)))
            for gv in sc["m"][i].Gv:
                out.append(-I.m.data * gv[I.m.col - pm.l_m])
            if pm.b and sc["b"][i].Gv:
                out.append(-np.kron(I.b.data, np.concatenate(sc["b"][i].Gv)))
        return _cat(out, np.float64)

    def hess_dynamic(self, x, s, lam):
        d, pm = self.d, self._pm
        T, I = d.T_coo, d.I_coo
        out = []
        for i in range(self.n_x):
            for nd, part in ((self._n_xf[i], T.f), (self._n_xb[i], T.b)):
                if nd.Hv:
                    out.append(np.kron(part.data * lam[self.l_d[i] + part.row], np.concatenate(nd.Hv)))
        _, vb, dt = self.mstage(x, s)
        self._load_function_locals(self._dyn, self.F_d, None, vb, dt, hess=True)
        sc = self._dyn["scaled"]
        for i in range(self.n_x):
            if pm.f and sc["f"][i].Hv:
                out.append(-np.kron(I.f.data * lam[self.l_d[i] + I.f.row], np.concatenate(sc["f"][i].Hv)))
            for hv in sc["m"][i].Hv:
                out.append(-I.m.data * lam[self.l_d[i] + I.m.row] * hv[I.m.col - pm.l_m])
            if pm.b and sc["b"][i].Hv:
                out.append(-np.kron(I.b.data * lam[self.l_d[i] + I.b.row], np.concatenate(sc["b"][i].Hv)))
        return _cat(out, np.float64)

    # path constraints ----------------------------------------------------------------------
    def value_path(self, x, s):
        _, vb, _ = self.mstage(x, s)
        return _cat([fn.F(vb, self.L_m) for fn in self.F_c], np.float64)

    def jac_path(self, x, s):
        pm = self._pm
        _, vb, dt = self.mstage(x, s)
        self._load_function_locals(self._con, self.F_c, None, vb, dt, hess=False)
        raw, out = self._con["raw"], []
        for j in range(self.n_c):
            for w, on in (("f", pm.f), ("m", True), ("b", pm.b)):
                if on:
                    out.extend(raw[w][j].Gv)
        return _cat(out, np.float64)

    def hess_path(self, x, s, lam):
        pm = self._pm
        _, vb, dt = self.mstage(x, s)
        self._load_function_locals(self._con, self.F_c, None, vb, dt, hess=True)
        raw, out = self._con["raw"], []
        base = 0
        for j in range(self.n_c):
            if pm.f:
                out.extend(hv * lam[base] for hv in raw["f"][j].Hv)
            out.extend(hv * lam[base + pm.l_m: base + pm.r_m] for hv in raw["m"][j].Hv)
            if pm.b:
                out.extend(hv * lam[base + self.L_m - 1] for hv in raw["b"][j].Hv)
            base += self.L_m
        return _cat(out, np.float64)
