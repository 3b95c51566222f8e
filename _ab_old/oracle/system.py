"""NLP assembly and the cyipopt ``problem_obj`` callbacks (oracle, NumPy).

Restates the callback half of /root/reference/pockit/base/systembase.py:
  * layout l_p/r_p/l_s/r_s, symbols [I of every phase | s] ........... systembase.py:257-289
  * FUNC-boundary bounds promoted to system constraints, bare-symbol constraints -> bounds
    ................................................................... systembase.py:291-364
  * system-level nodes (integrals weighted by w_m, objective, constraints) :366-453,625-644,695-724
  * triplet layout ................................................... systembase.py:455-551
  * bounds ............................................................ systembase.py:553-590
  * objective / gradient / constraints / jacobian / hessian(_o,_c) .... systembase.py:592-835
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .chain import DNode, eval_gradient, eval_hessian, link_gradient, link_hessian
from .phase import FUNC, _cat
from .symfunc import SymFunc


def _globalize(index, l_p, r_s):
    """Phase-local index -> NLP index; negative (static parameter) indices count from r_s."""
    index = np.asarray(index)
    return np.where(index >= 0, index + l_p, index + r_s).astype(np.int64)


class System:
    Phase = None  # bound by the radau / lobatto namespaces

    def __init__(self, static_parameter, simplify=False, fastmath=False):
        if isinstance(static_parameter, int):
            names = [f"s_{i}" for i in range(static_parameter)]
        elif isinstance(static_parameter, list):
            names = static_parameter
        else:
            raise ValueError("static_parameter must be int or list of str")
        self.s = [sp.Symbol(nm) for nm in names]
        self.n_s = len(self.s)
        self._compile = (simplify, fastmath)
        self._next_id = 0
        self.p = []
        self._user_con = ([], [], [])
        self._objective_set = False
        self._ready = False

    # ------------------------------------------------------------------ modeling API
    def new_phase(self, state, control):
        self._next_id += 1
        return self.Phase(self._next_id - 1, state, control, self.s, *self._compile)

    def set_phase(self, phase):
        for i, p in enumerate(phase):
            if not p.ok:
                raise ValueError(f"Dynamics, boundary conditions, or discretization scheme of phase {i} "
                                 f"are not fully set")
        self.p = list(phase)
        self._ready = False
        return self

    def set_objective(self, objective, *, cache=None):
        self._expr_objective = sp.sympify(objective)
        self._objective_set = True
        self._ready = False
        return self

    def set_system_constraint(self, system_constraint, lower_bound, upper_bound, *, cache=None):
        lo, hi = list(lower_bound), list(upper_bound)
        if not len(system_constraint) == len(lo) == len(hi):
            raise ValueError("system_constraint, lower_bound and upper_bound must have the same length")
        self._user_con = (list(system_constraint), lo, hi)
        self._ready = False
        return self

    def update(self):
        self._ready = False
        self.prepare()

    @property
    def ok(self):
        return self._objective_set

    _LAZY = ("L", "l_p", "r_p", "l_i", "r_i", "l_s", "r_s", "v_lb", "v_ub", "c_lb", "c_ub", "n_c", "F_c", "F_o")

    def __getattr__(self, name):
        # layout attributes are built on first use (the reference rebuilds them eagerly in its setters)
        if name in System._LAZY and not self.__dict__.get("_ready", False):
            self.prepare()
            return self.__dict__[name]
        raise AttributeError(name)

    @property
    def n_p(self):
        return len(self.p)

    # ------------------------------------------------------------------ structure (setup)
    def prepare(self):
        if self._ready and all(p._ready for p in self.p):
            return
        for p in self.p:
            p.prepare()
        # layout
        sizes = [p.L for p in self.p]
        self.r_p = np.cumsum(sizes).astype(np.int64) if sizes else np.array([], np.int64)
        self.l_p = (self.r_p - np.array(sizes, dtype=np.int64)) if sizes else np.array([], np.int64)
        nI = [p.n_I for p in self.p]
        self.r_i = np.cumsum(nI).astype(np.int64) if nI else np.array([], np.int64)
        self.l_i = (self.r_i - np.array(nI, dtype=np.int64)) if nI else np.array([], np.int64)
        self.l_s = int(self.r_p[-1]) if sizes else 0
        self.r_s = self.l_s + self.n_s
        self.L = self.r_s
        self._symbols = [sym for p in self.p for sym in p.I] + list(self.s)

        # constraints: user + FUNC boundaries of bounded states/times (systembase.py:291-317)
        cons, lo, hi = (list(a) for a in self._user_con)
        for p in self.p:
            for k, lb, ub in p._var_bounds:
                if k < p.n_x and p.info_bc_0[k].kind == FUNC:
                    cons.append(p.bc_0[k]); lo.append(lb); hi.append(ub)
                if k < p.n_x and p.info_bc_f[k].kind == FUNC:
                    cons.append(p.bc_f[k]); lo.append(lb); hi.append(ub)
            for lb, ub in p._time_bounds:
                if p.info_t_0.kind == FUNC:
                    cons.append(p.t_0); lo.append(lb); hi.append(ub)
                if p.info_t_f.kind == FUNC:
                    cons.append(p.t_f); lo.append(lb); hi.append(ub)
        s_bounds, exprs, elo, ehi = [], [], [], []
        for c, lb, ub in zip(cons, lo, hi):
            if getattr(c, "is_symbol", False) and c in self.s:
                s_bounds.append((self.s.index(c), lb, ub))
            else:
                exprs.append(sp.sympify(c)); elo.append(lb); ehi.append(ub)
        self.F_c = [SymFunc(e, self._symbols, *self._compile) for e in exprs]
        self.n_c = len(self.F_c)
        self.F_o = SymFunc(self._expr_objective, self._symbols, *self._compile)

        # which integrals each consumer needs
        def uses(free):
            return [np.array([sym in free for sym in p.I], dtype=bool) for p in self.p]

        self._which_o = uses(self._expr_objective.free_symbols)
        self._which_c = uses(set().union(*[e.free_symbols for e in exprs]) if exprs else set())

        # system-level nodes
        self._n_static = [DNode().leaf(self.l_s + i) for i in range(self.n_s)]
        self._n_int = []
        for k, p in enumerate(self.p):
            pm = p._pm
            for i in range(p.n_I):
                nd = DNode()
                for w, on in (("f", pm.f), ("m", True), ("b", pm.b)):
                    if on:
                        src = p._int["scaled"][w][i]
                        nd.Gi += [_globalize(ix, self.l_p[k], self.r_s) for ix in src.Gi]
                        nd.Hr += [_globalize(ix, self.l_p[k], self.r_s) for ix in src.Hr]
                        nd.Hc += [_globalize(ix, self.l_p[k], self.r_s) for ix in src.Hc]
                self._n_int.append(nd)
        basic = self._n_int + self._n_static
        self._n_obj = DNode().local(self.F_o)
        self._n_obj.args = basic
        self._n_con = []
        for fn in self.F_c:
            nd = DNode().local(fn)
            nd.args = basic
            self._n_con.append(nd)
        link_gradient([self._n_obj] + self._n_con)
        link_hessian([self._n_obj] + self._n_con, "system")

        # triplet layout
        self.grad_col = _cat(self._n_obj.Gi, np.int64)
        self.hess_o_row, self.hess_o_col = _cat(self._n_obj.Hr, np.int64), _cat(self._n_obj.Hc, np.int64)
        jr = [np.full(len(gi), c, dtype=np.int64) for c, nd in enumerate(self._n_con) for gi in nd.Gi]
        jc = [gi for nd in self._n_con for gi in nd.Gi]
        hr = [ix for nd in self._n_con for ix in nd.Hr]
        hc = [ix for nd in self._n_con for ix in nd.Hc]
        row0 = self.n_c
        for k, p in enumerate(self.p):
            jr.append(row0 + p.jac_dyn_row)
            jc.append(_globalize(p.jac_dyn_col, self.l_p[k], self.r_s))
            row0 += int(p.r_d[-1])
            jr.append(row0 + p.jac_path_row)
            jc.append(_globalize(p.jac_path_col, self.l_p[k], self.r_s))
            row0 += p.n_c * p.L_m
            hr.append(_globalize(p.hess_dyn_row, self.l_p[k], self.r_s))
            hc.append(_globalize(p.hess_dyn_col, self.l_p[k], self.r_s))
            hr.append(_globalize(p.hess_path_row, self.l_p[k], self.r_s))
            hc.append(_globalize(p.hess_path_col, self.l_p[k], self.r_s))
        self.jac_row, self.jac_col = _cat(jr, np.int64), _cat(jc, np.int64)
        self.hess_c_row, self.hess_c_col = _cat(hr, np.int64), _cat(hc, np.int64)
        self.hess_row = np.concatenate([self.hess_o_row, self.hess_c_row])
        self.hess_col = np.concatenate([self.hess_o_col, self.hess_c_col])

        # bounds
        slo = np.full(self.n_s, -np.inf)
        shi = np.full(self.n_s, np.inf)
        for i, lb, ub in [b for p in self.p for b in p.s_b] + s_bounds:
            slo[i] = max(slo[i], lb)
            shi[i] = min(shi[i], ub)
        self.v_lb = np.concatenate([p.v_lb for p in self.p] + [slo])
        self.v_ub = np.concatenate([p.v_ub for p in self.p] + [shi])
        clo, chi = [np.array(elo, dtype=np.float64)], [np.array(ehi, dtype=np.float64)]
        for p in self.p:
            clo += [np.zeros(int(p.r_d[-1])), np.repeat(p.c_lb, p.L_m)]
            chi += [np.zeros(int(p.r_d[-1])), np.repeat(p.c_ub, p.L_m)]
        self.c_lb, self.c_ub = np.concatenate(clo), np.concatenate(chi)
        self._ready = True

    # ------------------------------------------------------------------ values
    def _split(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x[self.l_s: self.r_s], [x[l:r] for l, r in zip(self.l_p, self.r_p)]

    def _sys_args(self, which, x):
        s, xs = self._split(x)
        v = np.empty(len(self._symbols))
        for k, p in enumerate(self.p):
            v[self.l_i[k]: self.r_i[k]] = p.value_integral(which[k], xs[k], s)
        v[len(v) - self.n_s:] = s
        return v

    def _load_integral_nodes(self, which, x, hess):
        """Quadrature-weighted derivative entries of every needed integral (systembase.py:625-644,695-724)."""
        s, xs = self._split(x)
        n0 = 0
        for k, p in enumerate(self.p):
            p.deriv_integral(which[k], xs[k], s, hess)
            pm = p._pm
            for i in range(p.n_I):
                if not which[k][i]:
                    continue
                nd = self._n_int[n0 + i]
                nd.Gv, nd.Hv = [], []
                for w, on, wt in (("f", pm.f, p.w_m[0]), ("m", True, p.w_m[pm.l_m: pm.r_m]), ("b", pm.b, p.w_m[-1])):
                    if on:
                        src = p._int["scaled"][w][i]
                        nd.Gv += [v * wt for v in src.Gv]
                        if hess:
                            nd.Hv += [v * wt for v in src.Hv]
            n0 += p.n_I

    def objective(self, x):
        self.prepare()
        return self.F_o.F(self._sys_args(self._which_o, x), 1)[0]

    def gradient(self, x):
        self.prepare()
        self._load_integral_nodes(self._which_o, x, hess=False)
        self._n_obj.lg = self.F_o.G(self._sys_args(self._which_o, x), 1)
        eval_gradient([self._n_obj])
        grad = np.zeros(self.L)
        for ix, v in zip(self._n_obj.Gi, self._n_obj.Gv):
            np.add.at(grad, ix, v)
        return grad

    def constraints(self, x):
        self.prepare()
        vb = self._sys_args(self._which_c, x)
        out = [np.array([fn.F(vb, 1)[0] for fn in self.F_c], dtype=np.float64)]
        s, xs = self._split(x)
        for k, p in enumerate(self.p):
            out += [p.value_dynamic(xs[k], s), p.value_path(xs[k], s)]
        return np.concatenate(out)

    def jacobianstructure(self):
        self.prepare()
        return self.jac_row, self.jac_col

    def jacobian(self, x):
        self.prepare()
        self._load_integral_nodes(self._which_c, x, hess=False)
        vb = self._sys_args(self._which_c, x)
        for nd, fn in zip(self._n_con, self.F_c):
            nd.lg = fn.G(vb, 1)
        eval_gradient(self._n_con)
        out = [v for nd in self._n_con for v in nd.Gv]
        s, xs = self._split(x)
        for k, p in enumerate(self.p):
            out += [p.jac_dynamic(xs[k], s), p.jac_path(xs[k], s)]
        return _cat(out, np.float64)

    def hessianstructure_o(self):
        self.prepare()
        return self.hess_o_row, self.hess_o_col

    def hessian_o(self, x):
        self.prepare()
        self._load_integral_nodes(self._which_o, x, hess=True)
        vb = self._sys_args(self._which_o, x)
        self._n_obj.lg = self.F_o.G(vb, 1)
        self._n_obj.lh = self.F_o.H(vb, 1)
        eval_gradient([self._n_obj])
        eval_hessian([self._n_obj], "system")
        return _cat(self._n_obj.Hv, np.float64)

    def hessianstructure_c(self):
        self.prepare()
        return self.hess_c_row, self.hess_c_col

    def hessian_c(self, x, lam):
        self.prepare()
        lam = np.asarray(lam, dtype=np.float64)
        self._load_integral_nodes(self._which_c, x, hess=True)
        vb = self._sys_args(self._which_c, x)
        for nd, fn in zip(self._n_con, self.F_c):
            nd.lg = fn.G(vb, 1)
            nd.lh = fn.H(vb, 1)
        eval_gradient(self._n_con)
        eval_hessian(self._n_con, "system")
        out = [_cat(nd.Hv, np.float64) * lam[c] for c, nd in enumerate(self._n_con)]
        s, xs = self._split(x)
        r0 = self.n_c
        for k, p in enumerate(self.p):
            nd_rows = int(p.r_d[-1])
            out.append(p.hess_dynamic(xs[k], s, lam[r0: r0 + nd_rows]))
            r0 += nd_rows
            out.append(p.hess_path(xs[k], s, lam[r0: r0 + p.n_c * p.L_m]))
            r0 += p.n_c * p.L_m
        return _cat(out, np.float64)

    def hessianstructure(self):
        self.prepare()
        return self.hess_row, self.hess_col

    def hessian(self, x, lagrange, obj_factor):
        return np.concatenate([self.hessian_o(x) * obj_factor, self.hessian_c(x, lagrange)])
