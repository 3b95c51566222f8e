"""Transcription compiler: optimal-control model -> NLP layout + symbolic evaluation plan.

Runs once per (model, mesh) on the host.  It does, *symbolically and ahead of time*, what the
reference does numerically in every callback with its Node graph
(/root/reference/pockit/base/easyderiv.py:97-304, phasebase.py:661-825,1023-1337): the sparse
forward chain rule from per-function local derivatives to NLP-variable derivatives, including
boundary kinds (FREE / FIXED / FUNC of static parameters), free or parametrised phase times and
the ``f*dt`` scaling.  The result is

  * the NLP layout and the triplet structure in the reference's exact order
    (phasebase.py:854-995, systembase.py:455-551), and
  * for every callback an *evaluation plan*: per phase a list of per-node output expressions
    ("segments") and per boundary node a list of scalar expressions ("edge entries"), plus small
    integer tables telling the HIP kernels where each of them goes in the output arrays.

Output pieces (contiguous runs of the J / H value arrays):
  T   constant +-1 translation entries of one state            (phasebase.py:865-866,1077)
  I   one derivative entry of one dynamics function expanded over every nonzero of the
      integration matrix: -I[r,c] * value(c) (* lambda[r])     (phasebase.py:885-887,1120-1124,1280-1285)
  N   one derivative entry on every middle node                (phasebase.py:954-962,1325-1328)
  S   explicit scalar items (front/back nodes, system level)

Symbols the generated code must provide: the phase's own x/u/t/s symbols at the node,
``pk_dt`` (t_f - t_0), ``pk_tau`` (node position in [0,1]), ``pk_w`` (quadrature weight of the
node), ``pk_sigma``, ``pk_lams<c>`` (multiplier of system constraint c), ``pk_lp<j>`` (multiplier
of path constraint j at the node), the integral symbols I_k (only if a system-level function is
nonlinear in them) and placeholders pkF/pkG/pkH for the local derivatives of the model functions.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import sympy as sp

from .model import FIXED, FREE, FUNC
from .symbolic import SparseFunc

DT = sp.Symbol("pk_dt")
TAU = sp.Symbol("pk_tau")
WQ = sp.Symbol("pk_w")
SIG = sp.Symbol("pk_sigma")


def lam_sys(c):
    return sp.Symbol(f"pk_lams{c}")


def lam_path(j):
    return sp.Symbol(f"pk_lp{j}")


# compact Hessian mode: multipliers already contracted with the integration / translation blocks
def mu_sym(i):
    """mu_i(node) = sum_r (I_hat[r, c] d/2) * lambda[defect row r of state i]  over the interval(s) holding the node"""
    return sp.Symbol(f"pk_mu{i}")


def ltf_sym(i):
    """sum over the front translation entries of state i: T_f * lambda   (= lambda of its first defect row)"""
    return sp.Symbol(f"pk_ltf{i}")


def ltb_sym(i):
    """sum over the back translation entries of state i: T_b * lambda   (= -sum of lambda over the last interval)"""
    return sp.Symbol(f"pk_ltb{i}")


# ---------------------------------------------------------------------------------------------
# symbolic derivative nodes
# ---------------------------------------------------------------------------------------------
class SNode:
    """Gradient entries (idx, expr) and lower-triangular Hessian entries (ridx, cidx, expr).

    idx is ('v', a): the slot of variable a at *this* node (a per-node index), or ('c', k): one
    fixed phase-local slot k (negative k = static parameter -n_s+i, translated late)."""

    __slots__ = ("G", "H")

    def __init__(self, G=None):
        self.G = list(G) if G else []
        self.H = []


def _before(a, b):
    """Index order for lower-triangular placement: negative (static) indices sort last
    (reference: easyderiv.py:8-19)."""
    return (a < 0, a) < (b < 0, b)


class _Composer:
    def __init__(self, layout):
        self.l_v = layout.l_v

    def first(self, idx):
        return int(self.l_v[idx[1]]) + 1 if idx[0] == "v" else int(idx[1])

    def __call__(self, args, lg, lh=()):
        """Node with local gradient ``lg`` = [(arg position, expr)] and local lower-triangular
        Hessian ``lh`` = [(row arg, col arg, expr)]   (reference: easyderiv.py:120-131,231-275)."""
        nd = SNode()
        for a, ge in lg:
            nd.G += [(idx, v * ge) for idx, v in args[a].G]
        for a, ge in lg:
            nd.H += [(r, c, v * ge) for r, c, v in args[a].H]
        for pr, pc, he in lh:
            for ri, rv in args[pr].G:
                for ci, cv in args[pc].G:
                    fr, fc = self.first(ri), self.first(ci)
                    val = rv * cv * he
                    if pr == pc:
                        if not _before(fr, fc):
                            nd.H.append((ri, ci, val))
                    elif fr == fc:
                        nd.H.append((ri, ci, 2 * val))
                    elif _before(fr, fc):
                        nd.H.append((ci, ri, val))
                    else:
                        nd.H.append((ri, ci, val))
        return nd


@dataclass
class FuncRef:
    """One model function of a phase with its derivative placeholders."""
    tag: str                      # 'd<i>' dynamics, 'i<k>' integrand, 'c<j>' path constraint
    fn: SparseFunc
    F: sp.Symbol = None
    G: list = field(default_factory=list)
    H: list = field(default_factory=list)

    def __post_init__(self):
        self.F = sp.Symbol(f"pkF_{self.tag}")
        self.G = [sp.Symbol(f"pkG_{self.tag}_{r}") for r in range(len(self.fn.G_index))]
        self.H = [sp.Symbol(f"pkH_{self.tag}_{r}") for r in range(len(self.fn.H_index_row))]

    def base(self):
        """placeholder -> SymPy expression in the phase's argument symbols."""
        out = {self.F: self.fn.expr}
        out.update(zip(self.G, self.fn.grad))
        out.update(zip(self.H, self.fn.hess))
        return out


class PhasePlan:
    """Symbolic derivative structure of one phase (mesh-independent apart from slot numbers)."""

    def __init__(self, phase, index):
        self.phase, self.index = phase, index
        lay = self.layout = phase.layout
        nx, nu, ns = phase.n_x, phase.n_u, phase.n_s
        self.nx, self.nu, self.ns = nx, nu, ns
        comp = self.compose = _Composer(lay)
        L = lay.L

        static = [SNode([(("c", -ns + i), sp.Integer(1))]) for i in range(ns)]

        def boundary(info, slot):
            if info.t == FREE:
                return SNode([(("c", int(slot)), sp.Integer(1))])
            if info.t == FIXED:
                return SNode()
            f = info.v
            return comp(static, list(zip(f.G_index.tolist(), f.grad)),
                        list(zip(f.H_index_row.tolist(), f.H_index_col.tolist(), f.hess)))

        self.x_f = [boundary(phase.info_bc_0[i], lay.l_v[i]) for i in range(nx)]
        self.x_b = [boundary(phase.info_bc_f[i], lay.r_v[i] - 1) for i in range(nx)]
        x_m = [SNode([(("v", i), sp.Integer(1))]) for i in range(nx)]
        u_m = [SNode([(("v", nx + i), sp.Integer(1))]) for i in range(nu)]
        u_f = [SNode([(("c", int(lay.l_v[nx + i])), sp.Integer(1))]) for i in range(nu)]
        u_b = [SNode([(("c", int(lay.r_v[nx + i]) - 1), sp.Integer(1))]) if lay.has_back else SNode()
               for i in range(nu)]
        self.t_f = boundary(phase.info_t_0, L - 2)
        self.t_b = boundary(phase.info_t_f, L - 1)
        t_m = comp([self.t_f, self.t_b], [(0, 1 - TAU), (1, TAU)])
        self.dt = comp([self.t_f, self.t_b], [(0, sp.Integer(-1)), (1, sp.Integer(1))])
        s_m = [comp([static[i]], [(0, sp.Integer(1))]) for i in range(ns)]
        self.args = {
            "f": self.x_f + u_f + [self.t_f] + static,
            "m": x_m + u_m + [t_m] + s_m,
            "b": self.x_b + u_b + [self.t_b] + static,
        }
        self.where = ("f", "m", "b") if lay.has_back else ("f", "m")

        self.dyn = [FuncRef(f"d{i}", fn) for i, fn in enumerate(phase.F_d)]
        self.integ = [FuncRef(f"i{k}", fn) for k, fn in enumerate(phase.F_I)]
        self.path = [FuncRef(f"c{j}", fn) for j, fn in enumerate(phase.F_c)]
        self.dyn_nodes = [self._function_nodes(fr, scaled=True) for fr in self.dyn]
        self.int_nodes = [self._function_nodes(fr, scaled=True) for fr in self.integ]
        self.path_nodes = [self._function_nodes(fr, scaled=False) for fr in self.path]

    def _function_nodes(self, fr: FuncRef, scaled):
        """{'f'|'m'|'b': SNode} of f (path constraints) or f*dt (dynamics, integrands)."""
        fn, out = fr.fn, {}
        for w in self.where:
            raw = self.compose(self.args[w], list(zip(fn.G_index.tolist(), fr.G)),
                               list(zip(fn.H_index_row.tolist(), fn.H_index_col.tolist(), fr.H)))
            out[w] = self.compose([raw, self.dt], [(0, DT), (1, fr.F)], [(1, 0, sp.Integer(1))]) if scaled else raw
        return out

    def base(self):
        out = {}
        for fr in self.dyn + self.integ + self.path:
            out.update(fr.base())
        return out


# ---------------------------------------------------------------------------------------------
# evaluation plan containers
# ---------------------------------------------------------------------------------------------
@dataclass
class Seg:
    """A per-node output of a phase in one callback."""
    expr: sp.Expr
    kind: str          # 'I' expanded over the integration matrix, 'N' one value per middle node, 'D' (compact Jacobian)
                       # contracted with the integration matrix: one value per defect row of the state
    base: int          # offset of the piece in the output array
    state: int = -1    # kind 'I' / 'D': the state whose defect rows (and multipliers) the piece belongs to
    # kind 'D' only: the entry's expression at the front node / the back node (LGL), and what the translation block adds
    # to the first row / to every row of the last interval (FUNC boundary values; functions of the static parameters)
    front: sp.Expr = None
    back: sp.Expr = None
    tfront: sp.Expr = None
    tback: sp.Expr = None


@dataclass
class Item:
    """out[pos] = coef * E[list][eid] (* lambda[lam])  for the boundary nodes / system level."""
    pos: int
    coef: float
    lst: tuple         # ('f', phase) | ('b', phase) | ('s',)
    eid: int
    lam: int = -1


@dataclass
class OuterBlock:
    """One outer-product block of a system-level Hessian (objective / system constraint nonlinear in
    the integrals; reference: easyderiv.py:323-355,393-430).  A and B are gradient-entry value runs in
    the auxiliary buffer; m is the location of the scalar multiplier factor * d2F/da db."""
    pos: int
    offA: int
    lenA: int
    offB: int
    lenB: int
    offM: int
    tril: bool = False        # False: kron(A, B); True: lower-triangular products of (collapsed) A and B
    collapseA: bool = False   # replace the run by its sum (a dense column met the same column)
    collapseB: bool = False
    second: bool = False      # tril only: also emit the transposed products (off-diagonal local pair)
    count: int = 0            # number of output entries


class CallbackPlan:
    def __init__(self, nphase):
        self.segs = [[] for _ in range(nphase)]        # per phase: list[Seg]
        self.tconst = [[] for _ in range(nphase)]      # per phase: base offset of the T piece per state
        self.lists = {}                                # list key -> list[expr]
        self.items: list[Item] = []
        self.needs_I = False

    def entry(self, key, expr):
        lst = self.lists.setdefault(key, [])
        lst.append(expr)
        return len(lst) - 1


class SystemPlan:
    """Everything the code generator, the runtime tables and the structure queries need."""

    def __init__(self, system):
        self.system = system
        P = self.phase_plans = [PhasePlan(p, k) for k, p in enumerate(system.p)]
        nP = len(P)
        sizes = [p.L for p in system.p]
        self.r_p = np.cumsum(sizes).astype(np.int64) if sizes else np.zeros(0, np.int64)
        self.l_p = self.r_p - np.array(sizes, dtype=np.int64) if sizes else np.zeros(0, np.int64)
        self.n_s = system.n_s
        self.l_s = int(self.r_p[-1]) if sizes else 0
        self.r_s = self.l_s + self.n_s
        self.n = self.r_s
        self.s_syms = list(system.s)
        self.I_syms = [sym for p in system.p for sym in p.I]
        self.I_owner = [(k, i) for k, p in enumerate(system.p) for i in range(p.n_I)]
        self.sys_syms = self.I_syms + self.s_syms
        self._system_functions(system)

        # constraint row layout: [system | phase 0 defects | phase 0 path | phase 1 ...]
        self.n_sys = len(self.F_c)
        self.g_off, self.path_off = [], []
        row = self.n_sys
        for p in system.p:
            self.g_off.append(row)
            row += p.n_x * p.layout.L_d
            self.path_off.append(row)
            row += p.n_c * p.layout.L_m
        self.m = row
        self._bounds(system)

        self.jac = CallbackPlan(nP)
        self.hess = CallbackPlan(nP)
        self.aux = CallbackPlan(nP)       # auxiliary buffer: integral gradient entries (x w) and multipliers
        self.outer: list[OuterBlock] = []
        self.n_aux = 0
        self._aux_entries = {}
        self._plan_jacobian()
        self._plan_hessian()
        self._plan_values_and_gradient()

    # ------------------------------------------------------------------ system-level functions
    def _system_functions(self, system):
        """User system constraints + FUNC boundaries of bounded states/times; bare static symbols
        become bounds (reference: systembase.py:291-364)."""
        cons = list(system._system_constraint_user)
        lo = list(system._system_constraint_user_lower_bound)
        hi = list(system._system_constraint_user_upper_bound)
        for p in system.p:
            for i, lb, ub in p._variable_bounds_phase:
                if i < p.n_x and p.info_bc_0[i].t == FUNC:
                    cons.append(p.bc_0[i]); lo.append(lb); hi.append(ub)
                if i < p.n_x and p.info_bc_f[i].t == FUNC:
                    cons.append(p.bc_f[i]); lo.append(lb); hi.append(ub)
            for lb, ub in p._time_bounds_phase:
                if p.info_t_0.t == FUNC:
                    cons.append(p.t_0); lo.append(lb); hi.append(ub)
                if p.info_t_f.t == FUNC:
                    cons.append(p.t_f); lo.append(lb); hi.append(ub)
        self.static_bounds, exprs, elo, ehi = [], [], [], []
        for c, lb, ub in zip(cons, lo, hi):
            if getattr(c, "is_symbol", False) and c in self.s_syms:
                self.static_bounds.append((self.s_syms.index(c), lb, ub))
            else:
                exprs.append(sp.sympify(c)); elo.append(lb); ehi.append(ub)
        simp = system._simplify
        self.F_c = [SparseFunc(e, self.sys_syms, simp) for e in exprs]
        self.F_o = SparseFunc(system._expr_objective, self.sys_syms, simp)
        self._sys_lb, self._sys_ub = np.array(elo, dtype=np.float64), np.array(ehi, dtype=np.float64)
        nI = len(self.I_syms)
        self.which_o = sorted(a for a in self.F_o.free_args if a < nI)
        self.which_c = sorted({a for f in self.F_c for a in f.free_args if a < nI})

    def _bounds(self, system):
        slo = np.full(self.n_s, -np.inf)
        shi = np.full(self.n_s, np.inf)
        for i, lb, ub in [b for p in system.p for b in p.s_b] + self.static_bounds:
            slo[i] = max(slo[i], lb)
            shi[i] = min(shi[i], ub)
        self.v_lb = np.concatenate([p.v_lb for p in system.p] + [slo])
        self.v_ub = np.concatenate([p.v_ub for p in system.p] + [shi])
        clo, chi = [self._sys_lb], [self._sys_ub]
        for p in system.p:
            z = np.zeros(p.n_x * p.layout.L_d)
            clo += [z, np.repeat(p.c_lb, p.layout.L_m)]
            chi += [z, np.repeat(p.c_ub, p.layout.L_m)]
        self.c_lb, self.c_ub = np.concatenate(clo), np.concatenate(chi)

    # ------------------------------------------------------------------ index helpers
    def col(self, k, idx, q=None):
        """NLP column(s) of idx in phase k; q = node array for per-node indices."""
        lay = self.phase_plans[k].layout
        if idx[0] == "v":
            return self.l_p[k] + lay.l_v[idx[1]] + np.asarray(q, dtype=np.int64)
        c = int(idx[1])
        return np.int64(self.l_p[k] + c if c >= 0 else self.r_s + c)

    # ------------------------------------------------------------------ auxiliary buffer
    def _aux_scalar(self, key, expr):
        off = self.n_aux
        self.aux.items.append(Item(off, 1.0, key, self.aux.entry(key, expr)))
        self.n_aux += 1
        return off

    def _arg_entries(self, a):
        """[(NLP index array, (offset, length) in the auxiliary buffer)] of system argument a: the
        quadrature-weighted gradient entries of an integral (front | middle | back, as
        systembase.py:376-408) or the single unit entry of a static parameter."""
        if a in self._aux_entries:
            return self._aux_entries[a]
        nI = len(self.I_syms)
        out = []
        if a >= nI:
            out.append((np.array([self.l_s + (a - nI)], dtype=np.int64), (self._aux_scalar(("s",), sp.Integer(1)), 1)))
        else:
            k, i = self.I_owner[a]
            pp = self.phase_plans[k]
            lay = pp.layout
            nodes = pp.int_nodes[i]
            q = np.arange(lay.mid_lo, lay.mid_hi)
            for idx, e in nodes["f"].G:
                out.append((np.array([self.col(k, idx, 0)], dtype=np.int64), (self._aux_scalar(("f", k), WQ * e), 1)))
            for idx, e in nodes["m"].G:
                if lay.L_mid == 0:
                    continue
                self.aux.segs[k].append(Seg(WQ * e, "N", self.n_aux))
                out.append((np.broadcast_to(self.col(k, idx, q), q.shape).astype(np.int64), (self.n_aux, lay.L_mid)))
                self.n_aux += lay.L_mid
            if lay.has_back:
                for idx, e in nodes["b"].G:
                    out.append((np.array([self.col(k, idx, lay.L_m - 1)], dtype=np.int64),
                                (self._aux_scalar(("b", k), WQ * e), 1)))
        self._aux_entries[a] = out
        return out

    def _uses_I(self, expr):
        return bool(sp.sympify(expr).free_symbols & set(self.I_syms))

    # ------------------------------------------------------------------ Jacobian
    def _plan_jacobian(self):
        cb = self.jac
        rows, cols, pos = [], [], 0
        nI = len(self.I_syms)

        def scalar(row, colv, coef, key, expr, lam=-1):
            nonlocal pos
            rows.append(np.array([row], dtype=np.int64))
            cols.append(np.array([colv], dtype=np.int64))
            cb.items.append(Item(pos, coef, key, cb.entry(key, expr), lam))
            pos += 1

        # 1. system constraints (systembase.py:472-479,659-669)
        for c, fc in enumerate(self.F_c):
            for a, m in zip(fc.G_index.tolist(), fc.grad):
                cb.needs_I |= self._uses_I(m)
                if a >= nI:
                    scalar(c, self.l_s + (a - nI), 1.0, ("s",), m)
                    continue
                k, i = self.I_owner[a]
                pp = self.phase_plans[k]
                lay = pp.layout
                nodes = pp.int_nodes[i]
                for idx, e in nodes["f"].G:
                    scalar(c, self.col(k, idx, 0), 1.0, ("f", k), m * WQ * e)
                q = np.arange(lay.mid_lo, lay.mid_hi)
                for idx, e in nodes["m"].G:
                    cb.segs[k].append(Seg(m * WQ * e, "N", pos))
                    rows.append(np.full(lay.L_mid, c, dtype=np.int64))
                    cols.append(np.broadcast_to(self.col(k, idx, q), (lay.L_mid,)).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for idx, e in nodes["b"].G:
                        scalar(c, self.col(k, idx, lay.L_m - 1), 1.0, ("b", k), m * WQ * e)

        # 2. phases
        for k, pp in enumerate(self.phase_plans):
            lay, g0 = pp.layout, self.g_off[k]
            Ir, Ic = lay.I_mid_structure()
            Tr, Tc, _ = lay.T_mid_structure()
            for i in range(pp.nx):                                   # translation part
                r0 = g0 + lay.l_d[i]
                for t_row, t_val in zip(lay.Tf_row, lay.Tf_val):
                    for idx, e in pp.x_f[i].G:
                        scalar(r0 + t_row, self.col(k, idx, 0), float(t_val), ("f", k), e)
                cb.tconst[k].append(pos)
                rows.append(r0 + Tr)
                cols.append(self.l_p[k] + lay.l_v[i] + Tc)
                pos += lay.nnzT_mid
                for t_row, t_val in zip(lay.Tb_row, lay.Tb_val):
                    for idx, e in pp.x_b[i].G:
                        scalar(r0 + t_row, self.col(k, idx, lay.L_m - 1), float(t_val), ("b", k), e)
            for i in range(pp.nx):                                   # integration part
                r0 = g0 + lay.l_d[i]
                nodes = pp.dyn_nodes[i]
                for i_row, i_val in zip(lay.If_row, lay.If_val):
                    for idx, e in nodes["f"].G:
                        scalar(r0 + i_row, self.col(k, idx, 0), -float(i_val), ("f", k), e)
                for idx, e in nodes["m"].G:
                    cb.segs[k].append(Seg(e, "I", pos, i))
                    rows.append(r0 + Ir)
                    cols.append(np.broadcast_to(self.col(k, idx, Ic), Ic.shape).astype(np.int64))
                    pos += lay.nnzI_mid
                if lay.has_back:
                    for i_row, i_val in zip(lay.Ib_row, lay.Ib_val):
                        for idx, e in nodes["b"].G:
                            scalar(r0 + i_row, self.col(k, idx, lay.L_m - 1), -float(i_val), ("b", k), e)
            q = np.arange(lay.mid_lo, lay.mid_hi)
            for j in range(len(pp.path)):                            # path constraints
                r0 = self.path_off[k] + j * lay.L_m
                nodes = pp.path_nodes[j]
                for idx, e in nodes["f"].G:
                    scalar(r0, self.col(k, idx, 0), 1.0, ("f", k), e)
                for idx, e in nodes["m"].G:
                    cb.segs[k].append(Seg(e, "N", pos))
                    rows.append(r0 + q)
                    cols.append(np.broadcast_to(self.col(k, idx, q), q.shape).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for idx, e in nodes["b"].G:
                        scalar(r0 + lay.L_m - 1, self.col(k, idx, lay.L_m - 1), 1.0, ("b", k), e)
        self.nnz_J = pos
        self.jac_row = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        self.jac_col = np.concatenate(cols) if cols else np.zeros(0, np.int64)

    def jac_constant_runs(self, compact=False):
        """Runs ``[(start, stop)]`` of Jacobian positions (reference layout, or the compact one) whose value does not
        depend on x: the +-1 translation entries (the reference recomputes them in every call, phasebase.py:1071-1081) and
        boundary / system items whose expression is a number (a FREE boundary slot contributes ``coef * 1``).  A host
        shim fills them into its landing arrays once and leaves them out of the per-iterate copy
        (``pk_set_jac_constant_runs``)."""
        cb, nnz = (self.jacc, self.nnz_Jc) if compact else (self.jac, self.nnz_J)
        const = np.zeros(nnz + 1, dtype=np.int8)
        for k, pp in enumerate(self.phase_plans):
            for base in cb.tconst[k]:
                const[base: base + pp.layout.nnzT_mid] = 1
        for it in cb.items:
            if it.lam < 0 and not sp.sympify(cb.lists[it.lst][it.eid]).free_symbols:
                const[it.pos] = 1
        edges = np.flatnonzero(np.diff(np.concatenate(([0], const[:-1], [0]))))
        return [(int(a), int(b)) for a, b in zip(edges[0::2], edges[1::2])]

    # ------------------------------------------------------------------ compact Jacobian
    @property
    def jacc(self):
        """Compact (coalesced) Jacobian plan, built on first use.  The reference emits one triplet per nonzero of the
        integration matrix for EVERY derivative entry of a dynamics function (phasebase.py:885-887,1120-1124); for an
        entry whose column is the same on every node -- t_0, t_f, a static parameter -- that is K triplets per defect
        row on one (row, column).  Here such an entry is contracted with the integration block first,
        ``value(row r) = -sum_c (I_hat[r, c] d/2) e(c)`` (the I.F product of the constraints, applied to a derivative
        column), the front / back node's share and the translation block's FUNC-boundary share included, and scalar
        items that meet on one position are summed symbolically: one value per distinct (row, column) of the defect
        rows' dense columns.  Entries with a per-node column are unique already and keep the reference's form.  (What
        stays repeated: a state's own translation entry where d f_i / d x_i is not zero, and the rows of system
        constraints that depend on integrals.)  Scatter-added, the triplets give the reference's matrix."""
        if getattr(self, "_jacc", None) is None:
            self._plan_jacobian_compact()
        return self._jacc

    def _plan_jacobian_compact(self):
        nP = len(self.phase_plans)
        cb = CallbackPlan(nP)
        rows, cols, pos = [], [], 0
        nI = len(self.I_syms)

        def scalar(row, colv, coef, key, expr):
            nonlocal pos
            rows.append(np.array([row], dtype=np.int64))
            cols.append(np.array([colv], dtype=np.int64))
            cb.items.append(Item(pos, coef, key, cb.entry(key, expr)))
            pos += 1

        # 1. system constraints: the reference's entries (systembase.py:472-479,659-669)
        for c, fc in enumerate(self.F_c):
            for a, m in zip(fc.G_index.tolist(), fc.grad):
                cb.needs_I |= self._uses_I(m)
                if a >= nI:
                    scalar(c, self.l_s + (a - nI), 1.0, ("s",), m)
                    continue
                k, i = self.I_owner[a]
                pp = self.phase_plans[k]
                lay = pp.layout
                nodes = pp.int_nodes[i]
                for idx, e in nodes["f"].G:
                    scalar(c, self.col(k, idx, 0), 1.0, ("f", k), m * WQ * e)
                q = np.arange(lay.mid_lo, lay.mid_hi)
                for idx, e in nodes["m"].G:
                    cb.segs[k].append(Seg(m * WQ * e, "N", pos))
                    rows.append(np.full(lay.L_mid, c, dtype=np.int64))
                    cols.append(np.broadcast_to(self.col(k, idx, q), (lay.L_mid,)).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for idx, e in nodes["b"].G:
                        scalar(c, self.col(k, idx, lay.L_m - 1), 1.0, ("b", k), m * WQ * e)

        # 2. phases
        self.jacc_dense = []        # per phase: number of contracted (dense-column) entries
        for k, pp in enumerate(self.phase_plans):
            lay, g0 = pp.layout, self.g_off[k]
            Ir, Ic = lay.I_mid_structure()
            Tr, Tc, _ = lay.T_mid_structure()
            dense = lambda idx, lay=lay: idx[0] == "c" and (idx[1] < 0 or idx[1] >= lay.L - 2)  # noqa: E731
            merged = {}             # (row, col, list key) -> summed expression of the scalar items that meet there

            def item(row, colv, key, expr, merged=merged):
                tag = (int(row), int(colv), key)
                merged[tag] = merged.get(tag, sp.Integer(0)) + expr

            tdf = [dict() for _ in range(pp.nx)]      # state -> {dense idx: translation share of the first row}
            tdb = [dict() for _ in range(pp.nx)]      # ... of every row of the last interval
            for i in range(pp.nx):                                   # translation part
                r0 = g0 + lay.l_d[i]
                for t_row, t_val in zip(lay.Tf_row, lay.Tf_val):
                    for idx, e in pp.x_f[i].G:
                        if dense(idx):
                            tdf[i][idx] = tdf[i].get(idx, sp.Integer(0)) + float(t_val) * e
                        else:
                            item(r0 + t_row, self.col(k, idx, 0), ("f", k), float(t_val) * e)
                cb.tconst[k].append(pos)
                rows.append(r0 + Tr)
                cols.append(self.l_p[k] + lay.l_v[i] + Tc)
                pos += lay.nnzT_mid
                for idx, e in pp.x_b[i].G:
                    if dense(idx):                                    # (T_b: -1 on every row of the last interval)
                        tdb[i][idx] = tdb[i].get(idx, sp.Integer(0)) + float(lay.Tb_val[0]) * e
                    else:
                        for t_row, t_val in zip(lay.Tb_row, lay.Tb_val):
                            item(r0 + t_row, self.col(k, idx, lay.L_m - 1), ("b", k), float(t_val) * e)
            n_dense = 0
            for i in range(pp.nx):                                   # integration part
                r0 = g0 + lay.l_d[i]
                nodes = pp.dyn_nodes[i]
                where = {"f": nodes["f"].G, "m": nodes["m"].G, "b": nodes["b"].G if lay.has_back else []}
                for idx, e in where["m"]:                            # per-node columns: the reference's expanded entries
                    if dense(idx):
                        continue
                    cb.segs[k].append(Seg(e, "I", pos, i))
                    rows.append(r0 + Ir)
                    cols.append(np.broadcast_to(self.col(k, idx, Ic), Ic.shape).astype(np.int64))
                    pos += lay.nnzI_mid
                for i_row, i_val in zip(lay.If_row, lay.If_val):
                    for idx, e in where["f"]:
                        if not dense(idx):
                            item(r0 + i_row, self.col(k, idx, 0), ("f", k), -float(i_val) * e)
                for i_row, i_val in zip(lay.Ib_row, lay.Ib_val):
                    for idx, e in where["b"]:
                        if not dense(idx):
                            item(r0 + i_row, self.col(k, idx, lay.L_m - 1), ("b", k), -float(i_val) * e)
                order = []                                           # dense columns, in order of first appearance
                for lst in (where["f"], where["m"], where["b"], [(d, None) for d in tdf[i]], [(d, None) for d in tdb[i]]):
                    for idx, _ in lst:
                        if dense(idx) and idx not in order:
                            order.append(idx)
                total = lambda lst, d: sum((e for idx, e in lst if idx == d), sp.Integer(0))  # noqa: E731
                for d in order:
                    if total(where["m"], d) == 0:
                        # a column only the boundary nodes know (a FUNC boundary value's static parameters): the rows of
                        # the first / last interval, as scalar items
                        ef, eb = total(where["f"], d), total(where["b"], d)
                        for i_row, i_val in zip(lay.If_row, lay.If_val):
                            if ef != 0:
                                item(r0 + i_row, self.col(k, d), ("f", k), -float(i_val) * ef)
                        if tdf[i].get(d, 0) != 0:
                            item(r0 + int(lay.Tf_row[0]), self.col(k, d), ("f", k), tdf[i][d])
                        for i_row, i_val in zip(lay.Ib_row, lay.Ib_val):
                            if eb != 0:
                                item(r0 + i_row, self.col(k, d), ("b", k), -float(i_val) * eb)
                        if tdb[i].get(d, 0) != 0:
                            for t_row in lay.Tb_row:
                                item(r0 + t_row, self.col(k, d), ("b", k), tdb[i][d])
                        continue
                    cb.segs[k].append(Seg(total(where["m"], d), "D", pos, i, front=total(where["f"], d),
                                          back=total(where["b"], d), tfront=tdf[i].get(d, sp.Integer(0)),
                                          tback=tdb[i].get(d, sp.Integer(0))))
                    rows.append(r0 + np.arange(lay.L_d, dtype=np.int64))
                    cols.append(np.full(lay.L_d, self.col(k, d), dtype=np.int64))
                    pos += lay.L_d
                    n_dense += 1
            self.jacc_dense.append(n_dense)
            q = np.arange(lay.mid_lo, lay.mid_hi)
            for j in range(len(pp.path)):                            # path constraints: unique as they are
                r0 = self.path_off[k] + j * lay.L_m
                nodes = pp.path_nodes[j]
                for idx, e in nodes["f"].G:
                    item(r0, self.col(k, idx, 0), ("f", k), e)
                for idx, e in nodes["m"].G:
                    cb.segs[k].append(Seg(e, "N", pos))
                    rows.append(r0 + q)
                    cols.append(np.broadcast_to(self.col(k, idx, q), q.shape).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for idx, e in nodes["b"].G:
                        item(r0 + lay.L_m - 1, self.col(k, idx, lay.L_m - 1), ("b", k), e)
            for (row, colv, key), expr in merged.items():
                scalar(row, colv, 1.0, key, expr)
        self._jacc = cb
        self.nnz_Jc = pos
        self.jacc_row = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        self.jacc_col = np.concatenate(cols) if cols else np.zeros(0, np.int64)

    # ------------------------------------------------------------------ Hessian of the Lagrangian
    def _plan_hessian(self):
        cb = self.hess
        rows, cols, pos = [], [], 0
        nI = len(self.I_syms)

        def scalar(r, c, coef, key, expr, lam=-1):
            nonlocal pos
            rows.append(np.array([r], dtype=np.int64))
            cols.append(np.array([c], dtype=np.int64))
            cb.items.append(Item(pos, coef, key, cb.entry(key, expr), lam))
            pos += 1

        def system_function(fn, factor):
            """factor * Hessian of fn(I, s)   (systembase.py:413-453,735-784; easyderiv.py:358-459)."""
            nonlocal pos
            for a, m in zip(fn.G_index.tolist(), fn.grad):           # g-H part
                if a >= nI:
                    continue                                          # static leaves have no Hessian
                cb.needs_I |= self._uses_I(m)
                k, i = self.I_owner[a]
                pp = self.phase_plans[k]
                lay = pp.layout
                nodes = pp.int_nodes[i]
                for r, c, e in nodes["f"].H:
                    scalar(self.col(k, r, 0), self.col(k, c, 0), 1.0, ("f", k), factor * m * WQ * e)
                q = np.arange(lay.mid_lo, lay.mid_hi)
                for r, c, e in nodes["m"].H:
                    cb.segs[k].append(Seg(factor * m * WQ * e, "N", pos))
                    rows.append(np.broadcast_to(self.col(k, r, q), q.shape).astype(np.int64))
                    cols.append(np.broadcast_to(self.col(k, c, q), q.shape).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for r, c, e in nodes["b"].H:
                        scalar(self.col(k, r, lay.L_m - 1), self.col(k, c, lay.L_m - 1), 1.0, ("b", k),
                               factor * m * WQ * e)
            for pr, pc, h in zip(fn.H_index_row.tolist(), fn.H_index_col.tolist(), fn.hess):   # h-G part
                cb.needs_I |= self._uses_I(h)
                if pr >= nI and pc >= nI:
                    scalar(self.l_s + (pr - nI), self.l_s + (pc - nI), 1.0, ("s",), factor * h)
                    continue
                # outer products of whole gradient-entry runs (reference: easyderiv.py:323-355,393-430)
                cb.needs_I = True
                offM = self._aux_scalar(("s",), factor * h)
                diag = pr == pc
                for ri, (offA, lenA) in self._arg_entries(pr):
                    for ci, (offB, lenB) in self._arg_entries(pc):
                        a_i, a_o, a_l, b_i, b_o, b_l = ri, offA, lenA, ci, offB, lenB
                        if a_i[0] < b_i[0]:
                            if diag:
                                continue
                            a_i, a_o, a_l, b_i, b_o, b_l = b_i, b_o, b_l, a_i, a_o, a_l
                        if a_i[0] > b_i[0]:
                            rows.append(np.repeat(a_i, len(b_i)))
                            cols.append(np.tile(b_i, len(a_i)))
                            blk = OuterBlock(pos, a_o, a_l, b_o, b_l, offM, count=a_l * b_l)
                        else:
                            ca = len(a_i) > 1 and a_i[0] == a_i[-1]
                            cbb = len(b_i) > 1 and b_i[0] == b_i[-1]
                            idx = a_i[:1] if ca else a_i
                            tr, tc = np.tril_indices(len(idx))
                            rows.append(idx[tr]); cols.append(idx[tc])
                            if not diag:
                                rows.append(idx[tr]); cols.append(idx[tc])
                            blk = OuterBlock(pos, a_o, a_l, b_o, b_l, offM, tril=True, collapseA=ca, collapseB=cbb,
                                             second=not diag, count=len(tr) * (1 if diag else 2))
                        self.outer.append(blk)
                        pos += blk.count

        system_function(self.F_o, SIG)
        self.nnz_H_obj = pos
        for c, fc in enumerate(self.F_c):
            system_function(fc, lam_sys(c))

        for k, pp in enumerate(self.phase_plans):
            lay, g0 = pp.layout, self.g_off[k]
            Ir, Ic = lay.I_mid_structure()
            for i in range(pp.nx):                                   # FUNC boundary values (rare)
                lam0 = g0 + lay.l_d[i]
                for nd, t_rows, t_vals, key, qn in ((pp.x_f[i], lay.Tf_row, lay.Tf_val, ("f", k), 0),
                                                    (pp.x_b[i], lay.Tb_row, lay.Tb_val, ("b", k), lay.L_m - 1)):
                    if not nd.H:
                        continue
                    for t_row, t_val in zip(t_rows, t_vals):
                        for r, c, e in nd.H:
                            scalar(self.col(k, r, qn), self.col(k, c, qn), float(t_val), key, e, lam0 + t_row)
            for i in range(pp.nx):
                lam0 = g0 + lay.l_d[i]
                nodes = pp.dyn_nodes[i]
                for i_row, i_val in zip(lay.If_row, lay.If_val):
                    for r, c, e in nodes["f"].H:
                        scalar(self.col(k, r, 0), self.col(k, c, 0), -float(i_val), ("f", k), e, lam0 + i_row)
                for r, c, e in nodes["m"].H:
                    cb.segs[k].append(Seg(e, "I", pos, i))
                    rows.append(np.broadcast_to(self.col(k, r, Ic), Ic.shape).astype(np.int64))
                    cols.append(np.broadcast_to(self.col(k, c, Ic), Ic.shape).astype(np.int64))
                    pos += lay.nnzI_mid
                if lay.has_back:
                    for i_row, i_val in zip(lay.Ib_row, lay.Ib_val):
                        for r, c, e in nodes["b"].H:
                            scalar(self.col(k, r, lay.L_m - 1), self.col(k, c, lay.L_m - 1), -float(i_val),
                                   ("b", k), e, lam0 + i_row)
            q = np.arange(lay.mid_lo, lay.mid_hi)
            for j in range(len(pp.path)):
                lam0 = self.path_off[k] + j * lay.L_m
                nodes = pp.path_nodes[j]
                for r, c, e in nodes["f"].H:
                    scalar(self.col(k, r, 0), self.col(k, c, 0), 1.0, ("f", k), e, lam0)
                for r, c, e in nodes["m"].H:
                    cb.segs[k].append(Seg(e * lam_path(j), "N", pos))
                    rows.append(np.broadcast_to(self.col(k, r, q), q.shape).astype(np.int64))
                    cols.append(np.broadcast_to(self.col(k, c, q), q.shape).astype(np.int64))
                    pos += lay.L_mid
                if lay.has_back:
                    for r, c, e in nodes["b"].H:
                        scalar(self.col(k, r, lay.L_m - 1), self.col(k, c, lay.L_m - 1), 1.0, ("b", k), e,
                               lam0 + lay.L_m - 1)
        self.nnz_H = pos
        self.hess_row = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        self.hess_col = np.concatenate(cols) if cols else np.zeros(0, np.int64)

    # ------------------------------------------------------------------ compact Hessian
    @property
    def hessc(self):
        """Compact (coalesced) Hessian plan, built on first use: the K-fold duplication of every dynamics
        entry over the rows of the integration block (phasebase.py:923-928,1280-1285) is contracted into
        mu = I^T lambda per node, and entries of one node that hit the same (row, col) are summed
        symbolically.  Structure/values differ from the reference's triplet list but scatter-add to the
        same matrix."""
        if getattr(self, "_hessc", None) is None:
            self._plan_hessian_compact()
        return self._hessc

    def _plan_hessian_compact(self):
        if self.outer:
            raise NotImplementedError("compact Hessian layout is not available for objectives / system "
                                      "constraints that are nonlinear in the integrals")
        nP = len(self.phase_plans)
        cb = CallbackPlan(nP)
        rows, cols, pos = [], [], 0
        nI = len(self.I_syms)
        sys_funcs = [(self.F_o, SIG)] + [(fc, lam_sys(c)) for c, fc in enumerate(self.F_c)]

        def emit_scalars(key, table):
            nonlocal pos
            for (r, c), expr in table.items():
                rows.append(np.array([r], dtype=np.int64))
                cols.append(np.array([c], dtype=np.int64))
                cb.items.append(Item(pos, 1.0, key, cb.entry(key, expr)))
                pos += 1

        # system level: static-static pairs
        sys_tab = {}
        for fn, factor in sys_funcs:
            for pr, pc, h in zip(fn.H_index_row.tolist(), fn.H_index_col.tolist(), fn.hess):
                key = (self.l_s + (pr - nI), self.l_s + (pc - nI))
                sys_tab[key] = sys_tab.get(key, sp.Integer(0)) + factor * h
                cb.needs_I |= self._uses_I(h)
        emit_scalars(("s",), sys_tab)

        for k, pp in enumerate(self.phase_plans):
            lay = pp.layout
            mid, edge = {}, {"f": {}, "b": {}}

            def add(w, r, c, expr, k=k, lay=lay, mid=mid, edge=edge):
                if w == "m":
                    mid[(r, c)] = mid.get((r, c), sp.Integer(0)) + expr
                else:
                    qn = 0 if w == "f" else lay.L_m - 1
                    key = (int(self.col(k, r, qn)), int(self.col(k, c, qn)))
                    edge[w][key] = edge[w].get(key, sp.Integer(0)) + expr

            for fn, factor in sys_funcs:                              # integrand Hessians
                for a, m in zip(fn.G_index.tolist(), fn.grad):
                    if a >= nI or self.I_owner[a][0] != k:
                        continue
                    cb.needs_I |= self._uses_I(m)
                    nodes = pp.int_nodes[self.I_owner[a][1]]
                    for w in pp.where:
                        for r, c, e in nodes[w].H:
                            add(w, r, c, factor * m * WQ * e)
            for i in range(pp.nx):                                    # FUNC boundary values
                for r, c, e in pp.x_f[i].H:
                    add("f", r, c, ltf_sym(i) * e)
                for r, c, e in pp.x_b[i].H:
                    add("b", r, c, ltb_sym(i) * e)
            for i in range(pp.nx):                                    # dynamics, contracted with mu
                for w in pp.where:
                    for r, c, e in pp.dyn_nodes[i][w].H:
                        add(w, r, c, -mu_sym(i) * e)
            for j in range(len(pp.path)):                             # path constraints
                for w in pp.where:
                    for r, c, e in pp.path_nodes[j][w].H:
                        add(w, r, c, lam_path(j) * e)
            emit_scalars(("f", k), edge["f"])
            q = np.arange(lay.mid_lo, lay.mid_hi)
            for (r, c), expr in mid.items():
                cb.segs[k].append(Seg(expr, "N", pos))
                rows.append(np.broadcast_to(self.col(k, r, q), q.shape).astype(np.int64))
                cols.append(np.broadcast_to(self.col(k, c, q), q.shape).astype(np.int64))
                pos += lay.L_mid
            emit_scalars(("b", k), edge["b"])
        self._hessc = cb
        self.nnz_Hc = pos
        self.hessc_row = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        self.hessc_col = np.concatenate(cols) if cols else np.zeros(0, np.int64)

    # ------------------------------------------------------------------ f, grad f, g
    def _plan_values_and_gradient(self):
        nI = len(self.I_syms)
        self.needs_I_grad = any(self._uses_I(m) for m in self.F_o.grad)
        # dense gradient (systembase.py:646-657): per phase, per node the sum over referenced
        # integrals of (dF_o/dI_k) * w * d(phi_k dt)/dv_a ; entries on fixed slots are reductions.
        self.grad_var = []      # per phase: {'f'|'m'|'b': [expr per variable a]}
        self.grad_red = []      # per phase: {'f'|'m'|'b': {slot: expr}}  (slot: phase-local, <0 static)
        for k, pp in enumerate(self.phase_plans):
            lay = pp.layout
            nv = pp.nx + pp.nu
            var = {w: [sp.Integer(0)] * nv for w in pp.where}
            red = {w: {} for w in pp.where}
            node_slot = {"f": lambda a: int(lay.l_v[a]), "b": lambda a: int(lay.l_v[a]) + lay.L_m - 1}
            for a, m in zip(self.F_o.G_index.tolist(), self.F_o.grad):
                if a >= nI or self.I_owner[a][0] != k:
                    continue
                i = self.I_owner[a][1]
                for w in pp.where:
                    for idx, e in pp.int_nodes[i][w].G:
                        val = m * WQ * e
                        if idx[0] == "v":
                            var[w][idx[1]] = var[w][idx[1]] + val
                            continue
                        slot = int(idx[1])
                        hit = [v for v in range(nv) if w != "m" and node_slot[w](v) == slot]
                        if hit:
                            var[w][hit[0]] = var[w][hit[0]] + val
                        else:
                            red[w][slot] = red[w].get(slot, sp.Integer(0)) + val
            self.grad_var.append(var)
            self.grad_red.append(red)
        # direct dependence of the objective on static parameters
        self.grad_static = {a - nI: m for a, m in zip(self.F_o.G_index.tolist(), self.F_o.grad) if a >= nI}
        self.needs_I_grad |= any(self._uses_I(m) for m in self.grad_static.values())
        # reduction slots per phase, in a fixed order: t_0, t_f, then static parameters
        self.grad_red_slots = []
        for k, pp in enumerate(self.phase_plans):
            slots = set()
            for w in pp.where:
                slots |= set(self.grad_red[k][w])
            self.grad_red_slots.append(sorted(slots, key=lambda s: (s < 0, s)))
        self.needs_I_con = any(a < nI for f in self.F_c for a in f.free_args)
