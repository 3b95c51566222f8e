"""Guess / solution container for one phase (host-side API plumbing either side of the hot path).

Mirrors the accessors of the reference's ``Variable`` (/root/reference/pockit/base/variablebase.py:92-140,
319-363), its interpolation / differentiation matrices and ``adapt`` (:137-391; SURVEY.md section 8(f) rank 3,
host-side: once per mesh refinement, not per NLP iteration) and its two guess helpers (:393-470).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

from .model import FIXED


def _near(a, b):
    return abs(a - b) <= 1e-8 + 1e-8 * abs(b)


def lagrange_values(nodes, points):
    """V[r, c] = value at points[r] of the Lagrange basis polynomial of nodes[c] (reference: variablebase.py:11-40;
    here in barycentric form, exact rows of the identity where a point coincides with a node)."""
    nodes, points = np.asarray(nodes, dtype=np.float64), np.asarray(points, dtype=np.float64)
    n = len(nodes)
    if len(points) == 0:
        return np.zeros((0, n))
    if n == 1:
        return np.ones((len(points), 1))
    span = nodes[-1] - nodes[0]
    z, p = (nodes - nodes[0]) / span, (points - nodes[0]) / span
    diff = z[:, None] - z[None, :]
    np.fill_diagonal(diff, 1.0)
    bw = 1.0 / np.prod(diff, axis=1)
    d = p[:, None] - z[None, :]
    hit = d == 0.0
    d[hit] = 1.0
    terms = bw[None, :] / d
    with np.errstate(divide="ignore", invalid="ignore"):      # (rows of points that coincide with a node are replaced below)
        V = terms / np.sum(terms, axis=1, keepdims=True)
    rows = np.any(hit, axis=1)
    V[rows] = hit[rows]
    return V


def lagrange_derivatives(nodes, points):
    """D[r, c] = derivative at points[r] of the Lagrange basis polynomial of nodes[c] (variablebase.py:43-62):
    L_c'(x) = sum_{k != c} 1 / (x_c - x_k) prod_{m != c, k} (x - x_m) / (x_c - x_m)."""
    nodes, points = np.asarray(nodes, dtype=np.float64), np.asarray(points, dtype=np.float64)
    n = len(nodes)
    if len(points) == 0:
        return np.zeros((0, n))
    if n == 1:
        return np.zeros((len(points), 1))
    span = nodes[-1] - nodes[0]
    z, p = (nodes - nodes[0]) / span, (points - nodes[0]) / span
    D = np.zeros((len(p), n))
    for c in range(n):
        for k in range(n):
            if k == c:
                continue
            term = np.full(len(p), 1.0 / (z[c] - z[k]))
            for m in range(n):
                if m != c and m != k:
                    term *= (p - z[m]) / (z[c] - z[m])
            D[:, c] += term
    return D / span


class _BatchView:
    def __init__(self, data, left, right):
        self._data, self._left, self._right = data, left, right

    def __getitem__(self, i):
        return self._data[self._left[i]: self._right[i]]

    def __setitem__(self, i, value):
        self._data[self._left[i]: self._right[i]] = value

    def __len__(self):
        return len(self._left)


class Variable:
    def __init__(self, phase, data):
        self._data = data
        nx = phase.n_x
        self._x = _BatchView(data, phase.l_v[:nx], phase.r_v[:nx])
        self._u = _BatchView(data, phase.l_v[nx:], phase.r_v[nx:])
        self._tx01, self._tu01 = phase.t_x, phase.t_u
        lay = phase.layout
        self._mesh, self._N = lay.mesh, lay.N
        lgr = lay.scheme == "lgr"
        # node window [lo, hi) of every interval on the state / control time axes, and whether consecutive
        # windows share their end node (column layout of the assembled matrices)
        self._win_x = (lay.lm, lay.lm + lay.K + 1) if lgr else (lay.lm, lay.rm)
        self._win_u = (lay.lm, lay.rm)
        self._ncol_x, self._ncol_u = lay.state_len, lay.L_m

    # ------------------------------------------------------------------ interpolation (variablebase.py:137-317)
    def _scaled(self, t):
        """Validate output times and map them to [0, 1] (variablebase.py:155-169)."""
        t = np.array(t, dtype=np.float64)
        for a, b in zip(t[:-1], t[1:]):
            if a > b and not np.isclose(a, b):
                raise ValueError("t is not in ascending order")
        if t[0] < self.t_0:
            if not np.isclose(t[0], self.t_0, rtol=0, atol=1e-8):
                raise ValueError("t[0] must be greater than or equal to t_0")
            t[0] = self.t_0
        if t[-1] > self.t_f:
            if not np.isclose(t[-1], self.t_f, rtol=0, atol=1e-8):
                raise ValueError("t[-1] must be less than or equal to t_f")
            t[-1] = self.t_f
        return (t - self.t_0) / (self.t_f - self.t_0)

    def _by_interval(self, t):
        """Output points grouped by mesh interval (variablebase.py:137-153): a point on an interior mesh point
        belongs to the interval on its left, a repeated one to the interval on its right."""
        groups = [[] for _ in range(self._N)]
        j = 0
        for i, ti in enumerate(t):
            while self._mesh[j + 1] < ti and not _near(self._mesh[j + 1], ti):
                j += 1
            if j + 1 < self._N and i > 0 and _near(self._mesh[j + 1], ti) and _near(t[i - 1], ti):
                j += 1
            groups[j].append(ti)
        return groups

    def _assemble(self, t, nodes01, window, ncol, basis):
        groups = self._by_interval(self._scaled(t)) if len(t) else [[] for _ in range(self._N)]
        lo, hi = window
        rows, cols, vals = [], [], []
        r0 = 0
        for j, pts in enumerate(groups):
            if not pts:
                continue
            B = basis(nodes01[lo[j]: hi[j]], np.array(pts))
            nr, nc = B.shape
            rows.append(np.repeat(r0 + np.arange(nr), nc))
            cols.append(np.tile(lo[j] + np.arange(nc), nr))
            vals.append(B.ravel())
            r0 += nr
        if not rows:
            return scipy.sparse.csr_array((r0, ncol))
        M = scipy.sparse.coo_array((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                   shape=(r0, ncol))
        M.sum_duplicates()
        M.eliminate_zeros()
        return M.tocsr()

    def V_x(self, t):
        """Interpolation matrix of the states to the times ``t`` (CSR): ``V_x(t) @ v.x[i]``."""
        return self._assemble(t, self._tx01, self._win_x, self._ncol_x, lagrange_values)

    def V_u(self, t):
        return self._assemble(t, self._tu01, self._win_u, self._ncol_u, lagrange_values)

    def D_x(self, t):
        """Differentiation matrix of the states at the times ``t`` (CSR): ``D_x(t) @ v.x[i]`` = dx_i/dt."""
        return self._assemble(t, self._tx01, self._win_x, self._ncol_x, lagrange_derivatives) / (self.t_f - self.t_0)

    def D_u(self, t):
        return self._assemble(t, self._tu01, self._win_u, self._ncol_u, lagrange_derivatives) / (self.t_f - self.t_0)

    def adapt(self, phase):
        """A new ``Variable`` on the discretization of ``phase``, interpolated from this one (variablebase.py:365-391)."""
        span = self.t_f - self.t_0
        Vx, Vu = self.V_x(phase.t_x * span + self.t_0), self.V_u(phase.t_u * span + self.t_0)
        data = np.empty(phase.L)
        nx = phase.n_x
        for i in range(nx):
            data[phase.l_v[i]: phase.r_v[i]] = Vx @ self.x[i]
        for i in range(phase.n_u):
            data[phase.l_v[nx + i]: phase.r_v[nx + i]] = Vu @ self.u[i]
        data[-2:] = self._data[-2:]
        return type(self)(phase, data)

    x = property(lambda self: self._x)
    u = property(lambda self: self._u)
    data = property(lambda self: self._data)
    t_x = property(lambda self: self._tx01 * (self.t_f - self.t_0) + self.t_0)
    t_u = property(lambda self: self._tu01 * (self.t_f - self.t_0) + self.t_0)

    @property
    def t_0(self):
        return self._data[-2]

    @t_0.setter
    def t_0(self, value):
        self._data[-2] = value

    @property
    def t_f(self):
        return self._data[-1]

    @t_f.setter
    def t_f(self, value):
        self._data[-1] = value


def _guess_times(v, phase):
    if phase.info_t_0.t == FIXED:
        v.t_0 = phase.t_0
    else:
        v.t_0 -= 0.5
    if phase.info_t_f.t == FIXED:
        v.t_f = phase.t_f
    else:
        v.t_f += 0.5
    return v


def constant_guess(phase, value: float = 1.0) -> Variable:
    """All variables ``value`` except FIXED boundary values/times."""
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(value), dtype=np.float64))
    for i in range(phase.n_x):
        if phase.info_bc_0[i].t == FIXED:
            v.x[i][0] = phase.bc_0[i]
        if phase.info_bc_f[i].t == FIXED:
            v.x[i][-1] = phase.bc_f[i]
    return _guess_times(v, phase)


def linear_guess(phase, default: float = 1.0) -> Variable:
    """States interpolated linearly between FIXED boundary values, ``default`` elsewhere."""
    if not phase.ok:
        raise ValueError("phase is not fully configured")
    v = Variable(phase, np.full(phase.L, float(default), dtype=np.float64))
    for i in range(phase.n_x):
        fixed0 = phase.info_bc_0[i].t == FIXED
        fixedf = phase.info_bc_f[i].t == FIXED
        if fixed0 and fixedf:
            v.x[i] = v._tx01 * (phase.bc_f[i] - phase.bc_0[i]) + phase.bc_0[i]
        elif fixed0:
            v.x[i] = phase.bc_0[i]
        elif fixedf:
            v.x[i] = phase.bc_f[i]
    return _guess_times(v, phase)
