"""Host side of mesh error estimation and continuous hp-refinement.

The device kernel ``pk_err`` (csrc/pk_kernels.hip.h) re-collocates every mesh interval with one more point
and returns both sides of the integral-form collocation equation on the augmented rule; this module builds
the tables that kernel indexes and turns its output into the per-interval verdicts and the new mesh.

Behaviour restated (not copied) from the reference:
  phasebase.py:1339-1372   _error_estimation_data_continuous   (device: pk_err; tables: error_tables below)
  phasebase.py:1374-1437   _error_check_interval_continuous / check_continuous     -> interval_ok
  phasebase.py:1522-1617   refine_continuous (raise the order while it fits, split the interval otherwise)
                                                                                   -> refined_discretization
"""
from __future__ import annotations

import math

import numpy as np

from . import collocation, runtime


def error_tables(plan):
    """(records, tables, n_out, views, groups) for ``pk_set_mesh_error_tables``.

    records: one PkErrIv per mesh interval; tables: float64 blob; views[k] = (offset, n_x, rows) of phase k in the
    outputs; groups: int32 [n, 2] = (first record, count) -- ONE wavefront of pk_err handles a run of consecutive
    intervals of one phase and one K, as many as fit its 64 lanes with K + 1 lanes per interval; every phase is padded
    to a multiple of 4 groups (count 0: a workgroup of 4 waves never mixes phases).  An interval with K + 1 > 64 is the
    first group of a block of its own (count 1, the block's other three groups carry count -1): all 256 threads of the
    workgroup walk its augmented nodes."""
    tables, blocks = [], {}
    size = 0

    def put(arr):
        nonlocal size
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        off = size
        tables.append(arr)
        size += len(arr)
        return off

    records, views, groups = [], [], []
    out_off = 0
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        lgr = lay.scheme == "lgr"
        rows_iv = lay.K + 1 if lgr else lay.K
        row0 = np.concatenate(([0], np.cumsum(rows_iv[:-1])))
        rows = int(np.sum(rows_iv))
        for j in range(lay.N):
            K = int(lay.K[j])
            key = (lgr, K)
            if key not in blocks:
                Vx, Vu, T, I, xa = (collocation.lgr_error_tables if lgr else collocation.lgl_error_tables)(K)
                blocks[key] = (put(np.concatenate([Vx.ravel(), Vu.ravel(), T.ravel(), I.ravel()])), xa)
            tab_off, xa = blocks[key]
            rec = np.zeros((), dtype=runtime.ERRIV_DTYPE)
            rec["phase"], rec["K"], rec["lm"], rec["row0"] = k, K, int(lay.lm[j]), int(row0[j])
            rec["tab_off"] = tab_off
            rec["tau_off"] = put(lay.mesh[j] + (xa + 1.0) * 0.5 * lay.width[j])
            rec["rows"], rec["out_off"], rec["width"] = rows, out_off, lay.width[j]
            records.append(rec)
        first = len(records) - lay.N
        j = 0
        while j < lay.N:
            K = int(lay.K[j])
            if K + 1 > runtime.WAVE:        # more augmented nodes than a wave has lanes: the interval takes a whole
                while len(groups) % runtime.WAVES_PER_BLOCK:      # workgroup (first group of a block, the others idle)
                    groups.append((first, 0))
                groups.append((first + j, 1))
                groups.extend([(first + j, -1)] * (runtime.WAVES_PER_BLOCK - 1))
                j += 1
                continue
            cap = max(1, runtime.WAVE // (K + 1))
            cnt = 1
            while j + cnt < lay.N and cnt < cap and int(lay.K[j + cnt]) == K:
                cnt += 1
            groups.append((first + j, cnt))
            j += cnt
        while len(groups) % runtime.WAVES_PER_BLOCK:
            groups.append((first, 0))
        views.append((out_off, pp.nx, rows))
        out_off += pp.nx * rows
    return (np.array(records, dtype=runtime.ERRIV_DTYPE), np.concatenate(tables), out_off, views,
            np.array(groups, dtype=np.int32).reshape(-1, 2))


def interval_rows(layout):
    """[lo, hi) of every interval in the row axis of the error data.  LGL windows are one row longer than the
    interval's own K rows (they reach into the next interval; the last one is clipped), as in the reference."""
    K = np.asarray(layout.K, dtype=np.int64)
    if layout.scheme == "lgr":
        hi = np.cumsum(K + 1)
        return hi - (K + 1), hi
    lo = np.concatenate(([0], np.cumsum(K[:-1])))
    return lo, lo + K + 1


def interval_ok(layout, T, I, atol, rtol, mtol):
    """Per-interval verdicts: |T - I| <= atol + rtol |I| on every row of the interval's window (NaN/inf fail);
    intervals narrower than ``mtol`` are accepted unchecked."""
    lo, hi = interval_rows(layout)
    with np.errstate(invalid="ignore"):
        good = np.abs(T - I) <= atol + rtol * np.abs(I)
    good &= np.isfinite(T) & np.isfinite(I)
    col_ok = np.all(good, axis=0)
    bad_before = np.concatenate(([0], np.cumsum(~col_ok)))
    hi = np.minimum(hi, len(col_ok))
    ok = bad_before[hi] == bad_before[lo]
    return ok | (layout.width < mtol)


def refined_discretization(layout, T, I, ok, rtol, num_point_min, num_point_max, mesh_length_min, mesh_length_max):
    """(mesh, num_point) after one refinement sweep.  For an interval that failed: the estimated number of extra
    points is ceil(log(e / rtol) / log K) (at least 1) with e the largest error relative to 1 + max|I| of the
    state; if K + extra fits below ``num_point_max`` the order is raised, otherwise the interval is split evenly
    into max(ceil((K + extra) / num_point_min), 2) pieces (clamped by the length limits) of ``num_point_min``."""
    lo, hi = interval_rows(layout)
    mesh, num_point = [], []
    for j in range(layout.N):
        K = int(layout.K[j])
        a, b = layout.mesh[j], layout.mesh[j + 1]
        if ok[j]:
            mesh.append(a)
            num_point.append(K)
            continue
        Tj, Ij = T[:, lo[j]: hi[j]], I[:, lo[j]: hi[j]]
        scale = 1.0 + np.max(np.abs(Ij), axis=1, keepdims=True)
        worst = float(np.max(np.abs(Tj - Ij) / scale))
        extra = max(int(np.ceil(np.log(worst / rtol) / np.log(K))), 1)
        if K + extra <= num_point_max:
            mesh.append(a)
            num_point.append(K + extra)
            continue
        pieces = max(math.ceil((K + extra) / num_point_min), 2)
        most = max(math.floor((b - a) / mesh_length_min), 1)
        least = math.ceil((b - a) / mesh_length_max)
        pieces = max(min(pieces, most), least)
        mesh.extend(np.linspace(a, b, pieces, endpoint=False))
        num_point.extend([num_point_min] * pieces)
    mesh.append(1.0)
    return mesh, num_point


# ------------------------------------------------------------------------------ bang-bang (switch-point) refinement
# Behaviour restated from the reference (phasebase.py:1619-1868 refine_discontinuous, :2280-2344 helpers); Radau
# only, as there.  Input: f_bb[b, node] = bang-bang constraint b scaled to [0, 1] at the collocation nodes.
SHOCK_THRESHOLD = 0.1        # a mesh point separates two intervals whose mean levels differ by more than this


def _switch_points(values, K):
    """Zeros in (-1, 1) of the degree K-1 interpolant of ``values`` (given at the K LGR nodes), ascending."""
    nodes, _ = collocation.lgr_nodes_weights(K)
    coef = np.linalg.solve(np.vander(nodes), values)          # highest power first
    zeros = np.roots(coef)
    real = zeros[np.isreal(zeros)].real
    return np.sort(real[(real > -1.0) & (real < 1.0)])


def _level(v, tol):
    """0: at the lower bound, 1: at the upper bound, 10: in between, -100: no neighbour."""
    if v is None:
        return -100
    return 0 if v < tol else (1 if v > 1 - tol else 10)


def _merge_mesh(candidates, old_interior, lmin, lmax):
    """Sorted candidates -> mesh: drop points closer than ``lmin`` to the ends, merge points closer than ``lmin``
    to their predecessor (an old mesh point yields to a new one, two new ones meet in the middle), split gaps wider
    than ``lmax`` evenly."""
    pts = [0.0] + [v for v in sorted(candidates) if lmin < v < 1 - lmin] + [1.0]
    old = set(float(v) for v in old_interior)
    mesh = [0.0]
    for v in pts[1:]:
        gap = v - mesh[-1]
        if gap < lmin:
            if float(mesh[-1]) in old:
                mesh[-1] = v
            elif float(v) not in old:
                mesh[-1] = (v + mesh[-1]) / 2
        elif gap > lmax:
            start, pieces = mesh[-1], int(np.ceil(gap / lmax))
            mesh.extend(start + (v - start) * (k + 1) / pieces for k in range(pieces))
        else:
            mesh.append(v)
    return np.array(mesh, dtype=np.float64)


def switch_point_discretization(layout, f_bb, tol, num_point_min, num_point_max, mesh_length_min, mesh_length_max):
    """(mesh, num_point) that puts mesh points on the switching times of the bang-bang constraints.

    Pass 1: an interval in which a constraint crosses 1/2 gets a mesh point at every zero of its interpolant minus
    1/2; a zero within ``mesh_length_min`` of an end of the interval replaces that mesh point instead (once).
    Pass 2: an interval that is neither switched nor saturated moves its offending end(s) inwards by the distance
    of its mean level from the nearest bound (or, if that end is already taken, the next mesh point across which the
    mean levels jump).  Mesh points across which nothing jumps are dropped.  Both passes sweep the left half
    left-to-right and the right half right-to-left, constraint by constraint, as the reference does."""
    N, mesh, lm, rm = layout.N, layout.mesh, layout.lm, layout.rm
    n_b = f_bb.shape[0]
    mean = np.empty((n_b, N))
    for j in range(N):
        _, w = collocation.lgr_nodes_weights(int(layout.K[j]))
        mean[:, j] = f_bb[:, lm[j]: rm[j]] @ w / 2
    calm = {p for p in range(1, N) if np.all(np.abs(mean[:, p - 1] - mean[:, p]) <= SHOCK_THRESHOLD)}
    placed, retired = [], set()
    settled = np.zeros((n_b, N), dtype=bool)
    half = N // 2
    sweeps = [(range(half), False), (range(N - 1, half - 1, -1), True)]

    def retire(p, position):
        """Replace old mesh point p by ``position`` unless it was replaced before."""
        if p in retired:
            return False
        retired.add(p)
        placed.append(position)
        return True

    for b in range(n_b):                                                          # pass 1
        for order, from_right in sweeps:
            for j in order:
                vals = f_bb[b, lm[j]: rm[j]]
                a, c = mesh[j], mesh[j + 1]
                if np.any(vals < 0.5) and np.any(vals > 0.5):
                    zeros = _switch_points(vals - 0.5, int(layout.K[j])) * (c - a) / 2 + (a + c) / 2
                    for z in (zeros[::-1] if from_right else zeros):
                        near_left, near_right = z < a + mesh_length_min, z > c - mesh_length_min
                        if from_right and near_right or not from_right and not near_left and near_right:
                            settled[b, j] |= retire(j + 1, z)
                        elif near_left:
                            settled[b, j] |= retire(j, z)
                        else:
                            placed.append(z)
                            settled[b, j] = True
                elif np.all(vals < tol) or np.all(vals > 1 - tol):
                    settled[b, j] = True

    def across(start, step):
        """First mesh point from ``start`` in direction ``step`` across which the mean levels jump."""
        p = start
        while p in calm:
            p += step
        return p

    for b in range(n_b):                                                          # pass 2
        for order, from_right in sweeps:
            for j in order:
                if settled[b, j]:
                    continue
                a, c = mesh[j], mesh[j + 1]
                shift = abs(mean[b, j] - round(mean[b, j])) * (c - a)
                vals = f_bb[b]
                left_ok = _level(vals[lm[j] - 1] if j > 0 else None, tol) + _level(vals[lm[j]], tol) <= 2
                right_ok = _level(vals[rm[j] - 1], tol) + _level(vals[rm[j]] if j < N - 1 else None, tol) <= 2

                def fix_left():
                    if not retire(j, a + shift):
                        p = across(j + 1, +1)
                        if p < N:
                            retire(p, mesh[p] - shift)

                def fix_right():
                    if not retire(j + 1, c - shift):
                        p = across(j, -1)
                        if p > 0:
                            retire(p, mesh[p] + shift)

                steps = [(fix_left, left_ok), (fix_right, right_ok)]
                for fix, fine in (steps[::-1] if from_right else steps):
                    if not fine:
                        fix()
    kept = [mesh[p] for p in range(1, N) if p not in retired and p not in calm]
    new_mesh = _merge_mesh(placed + kept, mesh[1:-1], mesh_length_min, mesh_length_max)
    short = min(1e-2, mesh_length_min * 10)
    num_point = [num_point_min if w < short else num_point_max for w in np.diff(new_mesh)]
    return new_mesh, num_point
