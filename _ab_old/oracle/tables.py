"""Per-K constant tables of the LGR / LGL schemes (oracle, NumPy + scipy.special).

Restates, following the reference's numerical recipe so tables agree to rounding:
  * xw_lgr .... /root/reference/pockit/radau/discretization.py:89-114
  * xw_lgl .... /root/reference/pockit/lobatto/discretization.py:80-110
  * integral_matrix (from +1 down to the output node, barycentric basis, Gauss-Legendre with
    max(30, 3n) points) ............ /root/reference/pockit/base/discretizationbase.py:41-180
  * I_lgr / I_lgl ... radau/discretization.py:185-196, lobatto/discretization.py:155-166
"""
from __future__ import annotations

import functools

import numpy as np
import scipy.special

_TOL = 1e-13


@functools.lru_cache(maxsize=None)
def lgr(K: int):
    """Legendre-Gauss-Radau nodes (including -1) and weights on [-1, 1]."""
    if K <= 0:
        raise ValueError("Number of interpolation points must be at least 1.")
    inner = np.roots(scipy.special.jacobi(K - 1, 0, 1))
    x = np.array(sorted([-1.0] + [r.real for r in inner]), dtype=np.float64)
    PK = scipy.special.legendre(K)
    w = (1.0 - x) / (K * np.polyval(PK, x)) ** 2
    return x, np.asarray(w, dtype=np.float64)


@functools.lru_cache(maxsize=None)
def lgl(K: int):
    """Legendre-Gauss-Lobatto nodes and weights on [-1, 1]."""
    if K <= 0:
        raise ValueError("Number of interpolation points must be at least 1.")
    if K == 1:
        return np.array([0.0]), np.array([2.0])
    n = K - 1
    Pn = scipy.special.legendre(n)
    inner = np.roots(np.polyder(Pn))
    x = sorted([-1.0] + [r.real for r in inner] + [1.0])
    c = 2.0 / n / (n + 1)
    w = [c] + [c / np.polyval(Pn, xi) ** 2 for xi in x[1:-1]] + [c]
    return np.array(x, dtype=np.float64), np.array(w, dtype=np.float64)


def _lagrange_basis_at(points, nodes, bw):
    """L[k, j] = L_j(points[k]) through the barycentric formula, exact at coincident nodes."""
    n = len(nodes)
    if n == 1:
        return np.ones((len(points), 1))
    d = points[:, None] - nodes[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        terms = bw[None, :] / d
        den = terms.sum(axis=1)
        L = terms / den[:, None]
    L[np.isclose(den, 0.0) | ~np.isfinite(den), :] = 0.0
    for k, pk in enumerate(points):
        hit = np.nonzero(np.isclose(pk, nodes, rtol=_TOL, atol=_TOL))[0]
        if len(hit):
            L[k, :] = 0.0
            L[k, hit[0]] = 1.0
    return L


def integration_matrix(nodes_in, nodes_out):
    """A[i, j] = integral of L_j from +1 to nodes_out[i]  (zero row when nodes_out[i] == 1)."""
    nodes_in = np.asarray(nodes_in, dtype=np.float64)
    nodes_out = np.asarray(nodes_out, dtype=np.float64)
    n, m = len(nodes_in), len(nodes_out)
    A = np.zeros((m, n))
    if n == 0 or m == 0:
        return A
    bw = np.ones(n)
    for j in range(n):
        for k in range(n):
            if k != j:
                bw[j] /= nodes_in[j] - nodes_in[k]
    gx, gw = np.polynomial.legendre.leggauss(max(30, 3 * n))
    for i, b in enumerate(nodes_out):
        if np.isclose(b, 1.0, rtol=_TOL, atol=_TOL):
            continue
        half = 0.5 * (b - 1.0)
        mid = 0.5 * (b + 1.0)
        A[i, :] = np.dot(half * gw, _lagrange_basis_at(half * gx + mid, nodes_in, bw))
    return A


@functools.lru_cache(maxsize=None)
def I_lgr(K: int):
    x, _ = lgr(K)
    return integration_matrix(x, x)


@functools.lru_cache(maxsize=None)
def I_lgl(K: int):
    x, _ = lgl(K)
    return integration_matrix(x, x[:-1])
