#!/usr/bin/env python3
"""Collocation tables in 200-digit arithmetic (mpmath): an INDEPENDENT pin for pockit_amd/collocation.py at the orders where
the reference's own tables (np.roots of a monomial-basis polynomial, radau/discretization.py:89-114) run out of digits.

Run from the repo root:   python tests/golden/make_hiprec.py        ->  tests/golden/hiprec_tables.npz

Nothing of the product or of the reference is used: nodes are the roots of P_{K-1} + P_K (LGR, plus -1) / of P'_{K-1}
(LGL, plus +-1) found by Newton's method on the three-term recurrence in multiprecision from Chebyshev-like starting points;
weights and the integration matrix ``A[i, j] = int_{+1}^{x_i} L_j`` follow from the exactness conditions on monomials
(``sum_j w_j x_j^k = int x^k``, ``sum_j A[i, j] x_j^k = (x_i^{k+1} - 1) / (k + 1)``), solved as Vandermonde systems with
enough digits to absorb their conditioning.  Stored rounded to float64.
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORDERS = (2, 3, 5, 8, 12, 13, 14, 15, 16, 17, 18, 19, 20, 24, 32, 48, 64, 96, 128)


def legendre(n, x):
    """(P_n(x), P_{n-1}(x)) by the three-term recurrence."""
    p0, p1 = mp.mpf(1), x
    if n == 0:
        return p0, mp.mpf(0)
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1, p0


def dlegendre(n, x):
    p, pm = legendre(n, x)
    return n * (pm - x * p) / (1 - x * x)


def newton(f, x0):
    x = mp.mpf(x0)
    for _ in range(200):
        h = mp.mpf(10) ** (-mp.mp.dps // 3)
        fx = f(x)
        d = (f(x + h) - f(x - h)) / (2 * h)
        step = fx / d
        x -= step
        if abs(step) < mp.mpf(10) ** (-(mp.mp.dps - 10)):
            break
    return x


def roots_in(f, n_roots, lo=-1, hi=1):
    """All n_roots simple roots of f in (lo, hi): bracket by sign changes on a fine Chebyshev-spaced grid, then Newton."""
    m = 40 * n_roots + 50
    grid = [mp.cos(mp.pi * (m - i) / m) for i in range(m + 1)]
    grid = [lo + (g + 1) / 2 * (hi - lo) for g in grid]
    vals = [f(g) for g in grid[1:-1]]
    pts = grid[1:-1]
    out = []
    for a, b, fa, fb in zip(pts[:-1], pts[1:], vals[:-1], vals[1:]):
        if fa == 0:
            out.append(a)
        elif fa * fb < 0:
            out.append(mp.findroot(f, (a, b), solver="anderson", tol=mp.mpf(10) ** (-(mp.mp.dps - 15)), maxsteps=500))
    assert len(out) == n_roots, (len(out), n_roots)
    return sorted(out)


def lgr_nodes(K):
    if K == 1:
        return [mp.mpf(-1)]
    # (P_{K-1} + P_K) / (1 + x) has the K - 1 interior Radau nodes as its roots
    f = lambda x: (legendre(K, x)[0] + legendre(K, x)[1]) / (1 + x)  # noqa: E731
    return [mp.mpf(-1)] + roots_in(f, K - 1)


def lgl_nodes(K):
    if K == 2:
        return [mp.mpf(-1), mp.mpf(1)]
    n = K - 1
    return [mp.mpf(-1)] + roots_in(lambda x: dlegendre(n, x), n - 1) + [mp.mpf(1)]


def tables(nodes, rows):
    """weights (exact for degree < K) and A[i, j] = int_{+1}^{nodes[i]} L_j for i < rows."""
    K = len(nodes)
    V = mp.matrix(K, K)                 # V[k, j] = x_j^k
    for j, x in enumerate(nodes):
        p = mp.mpf(1)
        for k in range(K):
            V[k, j] = p
            p *= x
    rhs = mp.matrix(K, 1 + rows)
    for k in range(K):
        rhs[k, 0] = (1 - mp.mpf(-1) ** (k + 1)) / (k + 1)
        for i in range(rows):
            rhs[k, 1 + i] = (nodes[i] ** (k + 1) - 1) / (k + 1)
    sol = mp.inverse(V) * rhs
    w = [sol[j, 0] for j in range(K)]
    A = [[sol[j, 1 + i] for j in range(K)] for i in range(rows)]
    return w, A


def main():
    out = {}
    for K in ORDERS:
        mp.mp.dps = 60 + 3 * K          # (Vandermonde conditioning ~ 10^(0.4 K))
        x = lgr_nodes(K)
        w, A = tables(x, K)
        out[f"lgr_x_{K}"] = np.array([float(v) for v in x])
        out[f"lgr_w_{K}"] = np.array([float(v) for v in w])
        out[f"lgr_I_{K}"] = np.array([[float(v) for v in row] for row in A])
        if K >= 2:
            x = lgl_nodes(K)
            w, A = tables(x, K - 1)
            out[f"lgl_x_{K}"] = np.array([float(v) for v in x])
            out[f"lgl_w_{K}"] = np.array([float(v) for v in w])
            out[f"lgl_I_{K}"] = np.array([[float(v) for v in row] for row in A])
        print("K =", K, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "hiprec_tables.npz"), **out)


if __name__ == "__main__":
    sys.exit(main())
