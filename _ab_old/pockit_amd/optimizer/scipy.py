"""SciPy ``trust-constr`` adapter, as ``pockit.optimizer.scipy.solve``
(/root/reference/pockit/optimizer/scipy.py:13-100): lower-triangular Hessians are mirrored to full
symmetric matrices; objective and constraint Hessians come from ``hessian_o`` / ``hessian_c``."""
from __future__ import annotations

import numpy as np

from ._common import postprocess, preprocess


def _reflection(func, row, col, n):
    from scipy.sparse import coo_array

    row, col = np.asarray(row), np.asarray(col)
    diag = np.nonzero(row == col)[0]

    def full(*args):
        data = func(*args)
        half = coo_array((data, (row, col)), shape=(n, n))
        dg = coo_array((data[diag], (row[diag], col[diag])), shape=(n, n))
        return half + half.T - dg

    return full


def solve(system, guess, optimizer_options=None):
    from scipy.optimize import Bounds, NonlinearConstraint, minimize
    from scipy.sparse import coo_array

    x_0, guess_is_variable, optimizer_options = preprocess(system, guess, optimizer_options)
    m = len(system.c_lb)
    hess_o = _reflection(system.hessian_o, *system.hessianstructure_o(), system.L)
    hess_c = _reflection(system.hessian_c, *system.hessianstructure_c(), system.L)
    jac = lambda x: coo_array((system.jacobian(x), system.jacobianstructure()), shape=(m, system.L))  # noqa: E731
    cons = NonlinearConstraint(system.constraints, system.c_lb, system.c_ub, jac=jac, hess=hess_c)
    res = minimize(system.objective, x_0, method="trust-constr", jac=system.gradient, hess=hess_o,
                   constraints=cons, bounds=Bounds(system.v_lb, system.v_ub), options=optimizer_options)
    return postprocess(system, res.x, guess_is_variable), res
