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
    """(records, tables, n_out, views) for ``pk_set_mesh_error_tables``.

    records: PkErrIv array, every phase padded to a multiple of 4 records with K = 0 (a workgroup of 4 waves
    never mixes phases); tables: float64 blob; views[k] = (offset, n_x, rows) of phase k in the outputs."""
    tables, blocks = [], {}
    size = 0

    def put(arr):
        nonlocal size
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        off = size
        tables.append(arr)
        size += len(arr)
        return off

    records, views = [], []
    out_off = 0
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        lgr = lay.scheme == "lgr"
        rows_iv = lay.K + 1 if lgr else lay.K
        row0 = np.concatenate(([0], np.cumsum(rows_iv[:-1])))
        rows = int(np.sum(rows_iv))
        if int(np.max(lay.K)) + 1 > runtime.WAVE:
            raise NotImplementedError("mesh error estimation maps the K + 1 augmented nodes of an interval to the "
                                      "64 lanes of a wavefront: num_point <= 63")
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
        while len(records) % runtime.WAVES_PER_BLOCK:
            rec = np.zeros((), dtype=runtime.ERRIV_DTYPE)
            rec["phase"] = k
            records.append(rec)
        views.append((out_off, pp.nx, rows))
        out_off += pp.nx * rows
    return np.array(records, dtype=runtime.ERRIV_DTYPE), np.concatenate(tables), out_off, views


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
