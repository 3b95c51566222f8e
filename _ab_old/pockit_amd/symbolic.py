"""Sparse symbolic derivatives of one scalar model expression.

This is the function-compiler front end of the MI355X evaluator: it fixes, per expression, the
*order* of the nonzero gradient and lower-triangular Hessian entries -- the order everything
downstream (triplet layout, generated HIP code) is keyed on.  The rule is the reference's
(/root/reference/pockit/base/fastfunc.py:237-269): gradient rows by ascending argument index;
for each gradient row j, Hessian entries (j, k) for ascending k <= j over the arguments the
gradient expression still depends on; symbolic zeros are dropped.

Unlike the reference no Python/NumPy callable is produced here: values are computed on the GPU
by code generated from ``expr`` / ``grad`` / ``hess`` (pockit_amd/codegen.py).
"""
from __future__ import annotations

import numpy as np
import sympy as sp


class SparseFunc:
    __slots__ = ("expr", "args", "G_index", "H_index_row", "H_index_col", "grad", "hess")

    def __init__(self, function, args, simplify: bool = False):
        self.args = list(args)
        expr = sp.sympify(function)
        if simplify:
            expr = sp.simplify(expr)
        self.expr = expr
        where = {sym: k for k, sym in enumerate(self.args)}
        unknown = [s for s in expr.free_symbols if s not in where]
        if unknown:
            raise ValueError(f"expression uses symbols that are not arguments of this function: {unknown}")

        def touched(e):
            return sorted(where[s] for s in e.free_symbols)

        self.grad, gi = [], []
        self.hess, hr, hc = [], [], []
        for j in touched(expr):
            gj = sp.diff(expr, self.args[j])
            if simplify:
                gj = sp.simplify(gj)
            if gj == 0:
                continue
            gi.append(j)
            self.grad.append(gj)
            for k in touched(gj):
                if k > j:
                    break
                hjk = sp.diff(gj, self.args[k])
                if simplify:
                    hjk = sp.simplify(hjk)
                if hjk == 0:
                    continue
                hr.append(j)
                hc.append(k)
                self.hess.append(hjk)
        self.G_index = np.array(gi, dtype=np.int32)
        self.H_index_row = np.array(hr, dtype=np.int32)
        self.H_index_col = np.array(hc, dtype=np.int32)

    @property
    def free_args(self):
        return {self.args.index(s) for s in self.expr.free_symbols}
