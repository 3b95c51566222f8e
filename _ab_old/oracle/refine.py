"""Mesh error estimation and continuous hp-refinement of one phase (oracle, NumPy).

Restates, interval by interval instead of through global sparse matrices:
  * augmented tables (one more point per interval) ........ /root/reference/pockit/radau/discretization.py:285-360,
                                                            /root/reference/pockit/lobatto/discretization.py:255-305
  * T_x_aug x  vs  dt * I_m_aug f(V_xu_aug x) .............. /root/reference/pockit/base/phasebase.py:1339-1369
  * per-interval verdict (np.allclose on the reference's slices) ... phasebase.py:1378-1390
  * check_continuous / refine_continuous ................... phasebase.py:1401-1437,1522-1617
"""
from __future__ import annotations

import functools

import numpy as np
import scipy.interpolate

from . import tables


def _lagrange_matrix(nodes, points):
    """M[r, c] = L_c(points[r]) for the Lagrange basis on ``nodes`` (reference recipe: scipy.interpolate.lagrange)."""
    cols = []
    for c in range(len(nodes)):
        y = np.zeros(len(nodes))
        y[c] = 1
        cols.append(np.polyval(scipy.interpolate.lagrange(nodes, y), points))
    return np.array(cols, dtype=np.float64).T


@functools.lru_cache(maxsize=None)
def aug_lgr(K):
    """(V_x (K+1)x(K+1), V_u (K+1)xK, T (K+1)x(K+1), I (K+1)x(K+1), aug nodes) of one LGR interval."""
    x, _ = tables.lgr(K)
    x1 = np.concatenate((x, [1.0]))
    xa, _ = tables.lgr(K + 1)
    Vx = _lagrange_matrix(x1, xa)
    Vu = _lagrange_matrix(x, xa)
    full = _lagrange_matrix(x1, np.concatenate((xa, [1.0])))
    T = full[:-1] - full[-1]
    return Vx, Vu, T, tables.I_lgr(K + 1), xa


@functools.lru_cache(maxsize=None)
def aug_lgl(K):
    """(V (K+1)xK, T KxK, I Kx(K+1), aug nodes) of one LGL interval."""
    x, _ = tables.lgl(K)
    xa, _ = tables.lgl(K + 1)
    V = _lagrange_matrix(x, xa)
    T = V[:-1] - V[-1]
    return V, T, tables.I_lgl(K + 1), xa


def error_data(phase, x, s):
    """(T_x_aug, I_f_aug), each (n_x, rows): rows = sum(K_j + 1) for LGR, sum(K_j) for LGL."""
    d = phase.d
    xs, _, dt = phase.mstage(x, s)
    tm = (xs[-1] + xs[-2]) / 2
    lgr = d.scheme == "lgr"
    K = np.asarray(d.num_point)
    width, mid = np.diff(d.mesh), (d.mesh[1:] + d.mesh[:-1]) / 2
    Tout, Iout = [], []
    for j in range(len(K)):
        k = int(K[j])
        lo = int(d.l_m[j])
        if lgr:
            Vx, Vu, T, I, xa = aug_lgr(k)
            xv = [xs[d.l_v[i] + lo: d.l_v[i] + lo + k + 1] for i in range(phase.n_x)]
        else:
            Vx, T, I, xa = aug_lgl(k)
            Vu = Vx
            xv = [xs[d.l_v[i] + lo: d.l_v[i] + lo + k] for i in range(phase.n_x)]
        uv = [xs[d.l_v[phase.n_x + i] + lo: d.l_v[phase.n_x + i] + lo + k] for i in range(phase.n_u)]
        n = len(xa)
        t_aug = ((xa * width[j] / 2 + mid[j]) - 0.5) * dt + tm
        vb = np.concatenate([Vx @ v for v in xv] + [Vu @ v for v in uv] + [t_aug, np.repeat(s, n)])
        Tout.append(np.array([T @ v for v in xv]))
        Iout.append(np.array([(I * width[j] / 2) @ f.F(vb, n) for f in phase.F_d]) * dt)
    return np.concatenate(Tout, axis=1), np.concatenate(Iout, axis=1)


def interval_slices(phase):
    """The reference's [l_m_aug, r_m_aug) of every interval (LGL slices overlap by one, as there)."""
    K = np.asarray(phase.d.num_point)
    if phase.d.scheme == "lgr":
        r = np.cumsum(K + 1)
        return r - (K + 1), r
    l = np.concatenate(([0], np.cumsum(K[:-1])))
    return l, l + K + 1


def check_intervals(phase, T, I, atol, rtol, mtol):
    l, r = interval_slices(phase)
    ok = np.ones(phase.N, dtype=bool)
    for j in range(phase.N):
        if phase._mesh[j + 1] - phase._mesh[j] < mtol:
            continue
        ok[j] = np.allclose(T[:, l[j]: r[j]], I[:, l[j]: r[j]], atol=atol, rtol=rtol)
    return ok


def check_continuous(phase, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
                     relative_tolerance_continuous=1e-8, tolerance_mesh=1e-4):
    if phase.n_s and static_parameter is None:
        raise ValueError("phase has static parameters, but the value of static parameters is not given")
    s = np.array([] if static_parameter is None else static_parameter, dtype=np.float64)
    T, I = error_data(phase, variable.data, s)
    return bool(np.all(check_intervals(phase, T, I, absolute_tolerance_continuous, relative_tolerance_continuous,
                                       tolerance_mesh)))


def refined_mesh(phase, T, I, ok, rtol, num_point_min, num_point_max, mesh_length_min, mesh_length_max):
    """New (mesh, num_point) from the per-interval error: raise the order while it fits, split otherwise."""
    l, r = interval_slices(phase)
    mesh_new, K_new = [], []
    for j in range(phase.N):
        kj = int(phase._num_point[j])
        if ok[j]:
            mesh_new.append(phase._mesh[j])
            K_new.append(kj)
            continue
        Tj, Ij = T[:, l[j]: r[j]], I[:, l[j]: r[j]]
        rel = np.abs(Tj - Ij) / (1.0 + np.max(np.abs(Ij), axis=1).reshape(-1, 1))
        extra = max(int(np.ceil(np.log(np.max(rel) / rtol) / np.log(kj))), 1)
        if kj + extra <= num_point_max:
            mesh_new.append(phase._mesh[j])
            K_new.append(kj + extra)
            continue
        width = phase._mesh[j + 1] - phase._mesh[j]
        n_min = int(np.ceil(width / mesh_length_max))
        n_max = max(int(np.floor(width / mesh_length_min)), 1)
        n_int = max(int(np.ceil((kj + extra) / num_point_min)), 2)
        n_int = max(min(n_int, n_max), n_min)
        for m in np.linspace(phase._mesh[j], phase._mesh[j + 1], n_int, endpoint=False):
            mesh_new.append(m)
            K_new.append(num_point_min)
    mesh_new.append(1.0)
    return mesh_new, K_new


def refine_continuous(phase, variable, static_parameter=None, absolute_tolerance_continuous=1e-8,
                      relative_tolerance_continuous=1e-8, num_point_min=6, num_point_max=12, mesh_length_min=1e-3,
                      mesh_length_max=1.0):
    s = np.array([] if static_parameter is None else static_parameter, dtype=np.float64)
    T, I = error_data(phase, variable.data, s)
    ok = check_intervals(phase, T, I, absolute_tolerance_continuous, relative_tolerance_continuous, mesh_length_min)
    if np.all(ok):
        return
    mesh_new, K_new = refined_mesh(phase, T, I, ok, relative_tolerance_continuous, num_point_min, num_point_max,
                                   mesh_length_min, mesh_length_max)
    phase.set_discretization(mesh_new, K_new)
