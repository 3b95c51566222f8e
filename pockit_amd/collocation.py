"""Collocation tables and per-phase mesh layout (host side, NumPy only).

What the reference computes with scipy.special / np.roots / a 30-point quadrature
(/root/reference/pockit/radau/discretization.py:89-114,185-196,
 /root/reference/pockit/lobatto/discretization.py:80-110,155-166,
 /root/reference/pockit/base/discretizationbase.py:98-180) is computed here from the Legendre
three-term recurrence: Golub-Welsch eigenvalues as starting points, Newton polishing, and an
exactly-integrating Gauss rule for the integration matrices.  Results agree with the reference
tables to a few ulp (tests/test_collocation.py against tests/golden/tables.npz).

``MeshLayout`` is the index bookkeeping of one phase on one mesh: node ranges, variable/defect
offsets, the per-interval "kinds" (distinct integration-block patterns, with the front/back
columns removed for the first/last interval) and the wave tiles the kernels iterate over.
"""
from __future__ import annotations

import functools
import os

import numpy as np

# Which numerical recipe produces the nodes / weights / integration matrices.
#   "accurate" (default)  three-term recurrence + Newton polishing + an exact Gauss rule: correct to ~1e-15 at every order
#                         (pinned by the multiprecision tables of tests/golden/make_hiprec.py up to K = 128);
#   "reference"           the reference's recipe -- interior nodes as np.roots of the monomial-basis Jacobi / Legendre
#                         polynomial (radau/discretization.py:89-114, lobatto/discretization.py:80-110), integration matrix
#                         by Gauss-Legendre with max(30, 3K) points on the barycentric basis (discretizationbase.py:98-180).
#                         Its tables lose digits with K (quadrature weights: 1e-10 at K = 16, 1e-8 at K = 20); selecting it
#                         makes results at K > 12 agree with the reference's to rounding instead of to that table error.
_RECIPE = os.environ.get("POCKIT_AMD_TABLES", "accurate")


def use_reference_recipe(flag: bool = True) -> None:
    """Switch the table recipe (see ``_RECIPE``) for layouts built from now on."""
    global _RECIPE
    _RECIPE = "reference" if flag else "accurate"
    for fn in (lgr_nodes_weights, lgl_nodes_weights, lgr_integration_matrix, lgl_integration_matrix, lgr_error_tables,
               lgl_error_tables):
        fn.cache_clear()


def _reference_nodes(scheme, K):
    import scipy.special

    if scheme == "lgr":
        inner = np.roots(scipy.special.jacobi(K - 1, 0, 1))
        x = np.array(sorted([-1.0] + [r.real for r in inner]), dtype=np.float64)
        w = (1.0 - x) / (K * np.polyval(scipy.special.legendre(K), x)) ** 2
        return x, np.asarray(w, dtype=np.float64)
    n = K - 1
    Pn = scipy.special.legendre(n)
    x = np.array(sorted([-1.0] + [r.real for r in np.roots(np.polyder(Pn))] + [1.0]), dtype=np.float64)
    c = 2.0 / n / (n + 1)
    w = np.array([c] + [c / np.polyval(Pn, xi) ** 2 for xi in x[1:-1]] + [c], dtype=np.float64)
    return x, w


def _reference_integration_matrix(nodes, out_nodes):
    """The reference's quadrature of the Lagrange basis (a fixed Gauss-Legendre rule of max(30, 3K) points per output
    node, barycentric evaluation), vectorised over the quadrature points."""
    n = len(nodes)
    gap = nodes[:, None] - nodes[None, :]
    np.fill_diagonal(gap, 1.0)
    bw = 1.0 / np.prod(gap, axis=1)
    gx, gw = np.polynomial.legendre.leggauss(max(30, 3 * n))
    A = np.zeros((len(out_nodes), n))
    for i, b in enumerate(out_nodes):
        if abs(b - 1.0) <= 1e-13:
            continue                                   # (the row of the end point: an empty integral)
        half = 0.5 * (b - 1.0)
        t = half * gx + 0.5 * (b + 1.0)
        if n == 1:
            A[i, 0] = half * gw.sum()
            continue
        d = t[:, None] - nodes[None, :]
        on_node = np.abs(d) <= 1e-13 * (1.0 + np.abs(nodes))[None, :]
        d[on_node] = 1.0
        terms = bw[None, :] / d
        L = terms / terms.sum(axis=1, keepdims=True)
        hit = on_node.any(axis=1)
        L[hit] = on_node[hit].astype(np.float64)
        A[i] = (half * gw) @ L
    return A


# ------------------------------------------------------------------------------ Legendre helpers
def _legendre(n, x):
    """P_n(x) and P_{n-1}(x) by recurrence (vectorized)."""
    x = np.asarray(x, dtype=np.float64)
    p_prev, p = np.ones_like(x), x.copy()
    if n == 0:
        return p_prev, np.zeros_like(x)
    for k in range(2, n + 1):
        p_prev, p = p, ((2 * k - 1) * x * p - (k - 1) * p_prev) / k
    return p, p_prev


def _legendre_deriv(n, x):
    """P'_n(x) for |x| < 1."""
    p, pm = _legendre(n, x)
    return n * (pm - x * p) / (1.0 - x * x)


def _gauss_legendre(m):
    k = np.arange(1, m)
    off = k / np.sqrt(4.0 * k * k - 1.0)
    x = np.linalg.eigvalsh(np.diag(off, 1) + np.diag(off, -1))
    for _ in range(3):
        p, pm = _legendre(m, x)
        x = x - p * (1.0 - x * x) / (m * (pm - x * p))
    p, pm = _legendre(m, x)
    dp = m * (pm - x * p) / (1.0 - x * x)
    return x, 2.0 / ((1.0 - x * x) * dp * dp)


@functools.lru_cache(maxsize=None)
def lgr_nodes_weights(K: int):
    """Legendre-Gauss-Radau nodes on [-1, 1) including -1, and quadrature weights."""
    if K < 1:
        raise ValueError("Number of interpolation points must be at least 1.")
    if _RECIPE == "reference":
        return _reference_nodes("lgr", K)
    m = K - 1
    if m == 0:
        x = np.array([-1.0])
    else:
        k = np.arange(m)
        diag = 1.0 / ((2 * k + 1.0) * (2 * k + 3.0))          # Jacobi(0,1) three-term recurrence
        k = np.arange(1, m)
        off = np.sqrt(k * (k + 1.0)) / (2 * k + 1.0)
        xi = np.linalg.eigvalsh(np.diag(diag) + np.diag(off, 1) + np.diag(off, -1))
        for _ in range(4):                                     # polish on P_{K-1} + P_K
            pK, pKm = _legendre(K, xi)
            dK = K * (pKm - xi * pK) / (1.0 - xi * xi)
            pm2 = _legendre(K - 1, xi)
            dKm = (K - 1) * (pm2[1] - xi * pm2[0]) / (1.0 - xi * xi)
            xi = xi - (pK + pKm) / (dK + dKm)
        x = np.concatenate(([-1.0], np.sort(xi)))
    pK, _ = _legendre(K, x)
    w = (1.0 - x) / (K * pK) ** 2
    return x, w


@functools.lru_cache(maxsize=None)
def lgl_nodes_weights(K: int):
    """Legendre-Gauss-Lobatto nodes on [-1, 1] and quadrature weights."""
    if K < 1:
        raise ValueError("Number of interpolation points must be at least 1.")
    if K == 1:
        return np.array([0.0]), np.array([2.0])
    if _RECIPE == "reference":
        return _reference_nodes("lgl", K)
    n = K - 1
    if n == 1:
        x = np.array([-1.0, 1.0])
    else:
        m = n - 1                                              # interior: roots of P'_n ~ Jacobi(1,1)
        k = np.arange(1, m)
        off = np.sqrt(k * (k + 2.0) / ((2 * k + 1.0) * (2 * k + 3.0)))
        xi = np.linalg.eigvalsh(np.diag(off, 1) + np.diag(off, -1)) if m > 1 else np.array([0.0])
        for _ in range(4):                                     # polish: (1-x^2)P'' = 2xP' - n(n+1)P
            p, _ = _legendre(n, xi)
            dp = _legendre_deriv(n, xi)
            d2p = (2.0 * xi * dp - n * (n + 1.0) * p) / (1.0 - xi * xi)
            xi = xi - dp / d2p
        x = np.concatenate(([-1.0], np.sort(xi), [1.0]))
    p, _ = _legendre(n, x)
    w = 2.0 / (n * (n + 1.0) * p * p)
    return x, w


def _integration_matrix(nodes, out_nodes):
    """A[i, j] = integral from +1 to out_nodes[i] of the j-th Lagrange basis on ``nodes``."""
    if _RECIPE == "reference":
        return _reference_integration_matrix(np.asarray(nodes, dtype=np.float64), np.asarray(out_nodes, dtype=np.float64))
    n = len(nodes)
    bw = np.array([1.0 / np.prod(nodes[j] - np.delete(nodes, j)) for j in range(n)])
    gx, gw = _gauss_legendre(n // 2 + 2)                       # exact for degree n-1
    A = np.zeros((len(out_nodes), n))
    for i, b in enumerate(out_nodes):
        if abs(b - 1.0) < 1e-14:
            continue
        half, mid = 0.5 * (b - 1.0), 0.5 * (b + 1.0)
        t = half * gx + mid
        if n == 1:
            L = np.ones((len(t), 1))
        else:
            d = t[:, None] - nodes[None, :]
            hit = np.abs(d) < 1e-15
            d[hit] = 1.0
            terms = bw[None, :] / d
            L = terms / terms.sum(axis=1, keepdims=True)
            rows = hit.any(axis=1)
            L[rows] = hit[rows].astype(np.float64)
        A[i] = (half * gw) @ L
    return A


@functools.lru_cache(maxsize=None)
def lgr_integration_matrix(K: int):
    x, _ = lgr_nodes_weights(K)
    return _integration_matrix(x, x)                           # K x K


@functools.lru_cache(maxsize=None)
def lgl_integration_matrix(K: int):
    x, _ = lgl_nodes_weights(K)
    return _integration_matrix(x, x[:-1])                      # (K-1) x K


# ------------------------------------------------------------------------------ mesh error estimation tables
def _lagrange_matrix(nodes, points):
    """M[r, c] = L_c(points[r]) for the Lagrange basis on ``nodes`` (barycentric form, exact at the nodes)."""
    nodes, points = np.asarray(nodes, dtype=np.float64), np.asarray(points, dtype=np.float64)
    n = len(nodes)
    bw = np.array([1.0 / np.prod(nodes[j] - np.delete(nodes, j)) for j in range(n)])
    d = points[:, None] - nodes[None, :]
    hit = np.abs(d) < 1e-14
    d[hit] = 1.0
    terms = bw[None, :] / d
    M = terms / terms.sum(axis=1, keepdims=True)
    rows = hit.any(axis=1)
    M[rows] = hit[rows].astype(np.float64)
    return M


@functools.lru_cache(maxsize=None)
def lgr_error_tables(K: int):
    """One LGR interval with one more point (reference: radau/discretization.py:285-360): V_x (K+1)x(K+1) from
    the K nodes + end point to the K+1 augmented nodes, V_u (K+1)xK, T = values - value at +1, I = I_lgr(K+1)."""
    x, _ = lgr_nodes_weights(K)
    x1 = np.concatenate((x, [1.0]))
    xa, _ = lgr_nodes_weights(K + 1)
    Vx = _lagrange_matrix(x1, xa)
    Vu = _lagrange_matrix(x, xa)
    T = Vx.copy()
    T[:, K] -= 1.0                                            # L_c(+1) = delta_{c,K}
    return Vx, Vu, T, lgr_integration_matrix(K + 1), xa


@functools.lru_cache(maxsize=None)
def lgl_error_tables(K: int):
    """One LGL interval with one more point (reference: lobatto/discretization.py:255-305): V (K+1)xK for states
    and controls, T KxK = values at the first K augmented nodes - value at +1, I = I_lgl(K+1) (K x (K+1))."""
    x, _ = lgl_nodes_weights(K)
    xa, _ = lgl_nodes_weights(K + 1)
    V = _lagrange_matrix(x, xa)
    T = V[:-1].copy()
    T[:, K - 1] -= 1.0
    return V, V, T, lgl_integration_matrix(K + 1), xa
