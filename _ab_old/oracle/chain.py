"""Sparse forward-mode chain rule on a DAG of derivative nodes (oracle, NumPy).

Restates /root/reference/pockit/base/easyderiv.py in one structure:
  * ``DNode`` ....................... easyderiv.py:22-94   (Node)
  * ``sorts_before`` ................ easyderiv.py:8-19    (_less: negative = static parameter = last)
  * ``link_gradient`` / ``eval_gradient`` ........ easyderiv.py:97-140
  * ``link_hessian`` / ``eval_hessian`` with level="phase"  (per-node, elementwise) ... :143-304
  * ... with level="system" (outer products over whole trajectories) .................. :307-459
"link_*" computes index structure (setup time), "eval_*" computes values (every callback).

Every entry of a node is a pair (index array, value array) of length 1 or ``node.n``; length-1
entries are broadcast when the consuming node is longer.
"""
from __future__ import annotations

import numpy as np

_I32 = np.int32


class DNode:
    def __init__(self, n: int = 1):
        self.n = n                      # number of trajectory points the node spans
        self.args: list["DNode"] = []
        # local (w.r.t. args) derivatives
        self.lg_arg = np.empty(0, _I32)             # arg position of each local gradient row
        self.lg = np.empty((0, 1))                  # local gradient rows
        self.lh_r = np.empty(0, _I32)               # local Hessian (row arg, col arg), lower tri
        self.lh_c = np.empty(0, _I32)
        self.lh = np.empty((0, 1))
        # global (w.r.t. NLP variables) derivatives as parallel lists
        self.Gi: list[np.ndarray] = []
        self.Gv: list[np.ndarray] = []
        self.Hr: list[np.ndarray] = []
        self.Hc: list[np.ndarray] = []
        self.Hv: list[np.ndarray] = []

    def leaf(self, index, value=1.0):
        """Make this node an independent variable (or arange of variables)."""
        index = np.atleast_1d(np.asarray(index, dtype=_I32))
        self.Gi = [index]
        self.Gv = [np.full(len(index), value, dtype=np.float64)]
        return self

    def local(self, func):
        """Attach the sparsity of a SymFunc as this node's local derivative pattern."""
        self.lg_arg = func.G_index
        self.lh_r = func.H_index_row
        self.lh_c = func.H_index_col
        return self


def sorts_before(a: int, b: int) -> bool:
    """Ordering used for lower-triangular placement: non-negative ascending, then negatives."""
    return (a < 0, a) < (b < 0, b)


def _stretch(arr, n):
    return np.full(n, arr[0], dtype=arr.dtype) if (len(arr) == 1 and n > 1) else arr


# ----------------------------------------------------------------------------- gradient
def link_gradient(nodes):
    for nd in nodes:
        if len(nd.lg_arg) == 0:
            continue
        nd.Gi = [_stretch(ix, nd.n) for a in nd.lg_arg for ix in nd.args[a].Gi]


def eval_gradient(nodes):
    for nd in nodes:
        if len(nd.lg_arg) == 0:
            continue
        # zip: a node whose local values were not loaded this call yields no entries
        nd.Gv = [_stretch(v, nd.n) * row for a, row in zip(nd.lg_arg, nd.lg) for v in nd.args[a].Gv]


# ----------------------------------------------------------------------------- Hessian
def _pairs(nd):
    return [(int(r), int(c)) for r, c in zip(nd.lh_r, nd.lh_c)]


def link_hessian(nodes, level="phase"):
    for nd in nodes:
        if not nd.args:
            continue
        nd.Hr, nd.Hc = [], []
        for a in nd.lg_arg:                                   # (dF/da) * Hessian(a)
            for r, c in zip(nd.args[a].Hr, nd.args[a].Hc):
                nd.Hr.append(_stretch(r, nd.n))
                nd.Hc.append(_stretch(c, nd.n))
        for pr, pc in _pairs(nd):                             # (d2F/da db) * grad(a) x grad(b)
            for ri in nd.args[pr].Gi:
                for ci in nd.args[pc].Gi:
                    if level == "phase":
                        _phase_cross_index(nd, ri, ci, pr == pc)
                    else:
                        _system_cross_index(nd, ri, ci, pr == pc)


def eval_hessian(nodes, level="phase"):
    for nd in nodes:
        if not nd.args:
            continue
        nd.Hv = []
        for a, row in zip(nd.lg_arg, nd.lg):
            for v in nd.args[a].Hv:
                nd.Hv.append(_stretch(v, nd.n) * row)
        for (pr, pc), h in zip(_pairs(nd), nd.lh):
            A, B = nd.args[pr], nd.args[pc]
            for ri, rv in zip(A.Gi, A.Gv):
                for ci, cv in zip(B.Gi, B.Gv):
                    if level == "phase":
                        _phase_cross_value(nd, ri, ci, rv, cv, pr == pc, h)
                    else:
                        _system_cross_value(nd, ri, ci, rv, cv, pr == pc, h)


# phase level: one product per trajectory point ------------------------------------------
def _phase_cross_index(nd, ri, ci, diag):
    r2, c2 = _stretch(ri, nd.n), _stretch(ci, nd.n)
    if len(ri) == 1 and len(ci) > 1:
        r2 = np.full(len(ci), ri[0], dtype=_I32)
    if len(ci) == 1 and len(ri) > 1:
        c2 = np.full(len(ri), ci[0], dtype=_I32)
    upper = sorts_before(ri[0], ci[0])
    if diag:
        if not upper:
            nd.Hr.append(r2)
            nd.Hc.append(c2)
    elif upper:
        nd.Hr.append(c2)
        nd.Hc.append(r2)
    else:
        nd.Hr.append(r2)
        nd.Hc.append(c2)


def _phase_cross_value(nd, ri, ci, rv, cv, diag, h):
    prod = _stretch(rv, nd.n) * _stretch(cv, nd.n) * h
    if diag:
        if not sorts_before(ri[0], ci[0]):
            nd.Hv.append(prod)
    elif ri[0] == ci[0]:
        nd.Hv.append(prod * 2)       # both orders of an off-diagonal local pair hit one entry
    else:
        nd.Hv.append(prod)


# system level: outer products between whole index arrays ----------------------------------
def _collapse_index(ix):
    return np.array([ix[0]], dtype=ix.dtype) if (len(ix) > 1 and ix[0] == ix[-1]) else ix


def _collapse_value(ix, v):
    return np.array([np.sum(v)]) if (len(ix) > 1 and ix[0] == ix[-1]) else v


def _system_cross_index(nd, ri, ci, diag):
    if ri[0] < ci[0]:
        if diag:
            return
        ri, ci = ci, ri
    if ri[0] > ci[0]:
        nd.Hr.append(np.repeat(ri, len(ci)))
        nd.Hc.append(np.tile(ci, len(ri)))
        return
    a = _collapse_index(ri)
    tr, tc = np.tril_indices(len(a))
    nd.Hr.append(a[tr])
    nd.Hc.append(a[tc])
    if not diag:
        nd.Hr.append(a[tr])
        nd.Hc.append(a[tc])


def _system_cross_value(nd, ri, ci, rv, cv, diag, h):
    if ri[0] < ci[0]:
        if diag:
            return
        ri, ci, rv, cv = ci, ri, cv, rv
    if ri[0] > ci[0]:
        nd.Hv.append(np.kron(rv, cv) * h)
        return
    rv2, cv2 = _collapse_value(ri, rv), _collapse_value(ci, cv)
    tr, tc = np.tril_indices(len(rv2))
    nd.Hv.append(rv2[tr] * cv2[tc] * h)
    if not diag:
        nd.Hv.append(cv2[tr] * rv2[tc] * h)
