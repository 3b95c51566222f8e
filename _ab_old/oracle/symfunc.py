"""SymPy expression -> vectorized value / sparse gradient / sparse lower-triangular Hessian (oracle).

Restates /root/reference/pockit/base/fastfunc.py:
  * sparsity + ordering rule (gradient rows by argument index; Hessian rows (j asc, k<=j asc),
    symbolic zeros dropped) ............................................. fastfunc.py:237-269
  * joint CSE per output group (F / G / H separately), integer powers <= 3 expanded to products
    ...................................................................... fastfunc.py:41-43,180,271-296
  * SoA calling convention ``fn(x, n)`` with x = [a_0(n) | a_1(n) | ...] ... fastfunc.py:298-308
The reference compiles the generated NumPy source with Numba; here the same array expressions run
in NumPy through ``sympy.lambdify`` (no JIT).
"""
from __future__ import annotations

import numpy as np
import sympy as sp
from sympy.codegen.rewriting import create_expand_pow_optimization

_expand_pow = create_expand_pow_optimization(3)


def _cse_basic(exprs):
    repl, red = sp.cse(exprs, optimizations="basic")
    repl = [(s, _expand_pow(e)) for s, e in repl]
    red = [_expand_pow(e) for e in red]
    return repl, red


def _vector_fn(args, exprs):
    if not exprs:
        return None
    return sp.lambdify(args, list(exprs), modules="numpy", cse=_cse_basic)


class SymFunc:
    """F/G/H of one scalar expression; attributes mirror the reference's FastFunc."""

    def __init__(self, function, args, simplify=False, fastmath=False, *, cache=None):
        expr = sp.sympify(function)
        self.args = list(args)
        if simplify:
            expr = sp.simplify(expr)
        self.expr = expr
        pos = {a: i for i, a in enumerate(self.args)}

        def free(e):
            return sorted(pos[s] for s in e.free_symbols)

        g_idx, g_expr, h_idx, h_expr = [], [], [], []
        for j in free(expr):
            d1 = sp.diff(expr, self.args[j])
            if simplify:
                d1 = sp.simplify(d1)
            if d1 == 0:
                continue
            g_idx.append(j)
            g_expr.append(d1)
            for k in free(d1):
                if k > j:
                    break
                d2 = sp.diff(d1, self.args[k])
                if simplify:
                    d2 = sp.simplify(d2)
                if d2 == 0:
                    continue
                h_idx.append((j, k))
                h_expr.append(d2)
        self.G_index = np.array(g_idx, dtype=np.int32)
        self.H_index_row = np.array([r for r, _ in h_idx], dtype=np.int32)
        self.H_index_col = np.array([c for _, c in h_idx], dtype=np.int32)
        self.expr_grad, self.expr_hess = g_expr, h_expr
        self._f = _vector_fn(self.args, [expr])
        self._g = _vector_fn(self.args, g_expr)
        self._h = _vector_fn(self.args, h_expr)
        self._nv = len(self.args)

    def _split(self, x, n):
        return [x[i * n: (i + 1) * n] for i in range(self._nv)]

    def F(self, x, n):
        out = np.empty(n, dtype=np.float64)
        out[:] = self._f(*self._split(x, n))[0]
        return out

    def _rows(self, fn, k, x, n):
        out = np.empty((k, n), dtype=np.float64)
        if fn is not None:
            for r, v in enumerate(fn(*self._split(x, n))):
                out[r] = v
        return out

    def G(self, x, n):
        return self._rows(self._g, len(self.G_index), x, n)

    def H(self, x, n):
        return self._rows(self._h, len(self.H_index_row), x, n)
