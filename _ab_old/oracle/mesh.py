"""Per-phase discretization tables: index partitions, node positions/weights, T and I (oracle).

Restates the ``Discretization`` objects of the reference (only the half consumed by the NLP
callbacks; the ``*_aug`` error-estimation tables are out of scope):
  * LGR: /root/reference/pockit/radau/discretization.py:14-87,117-166,199-257,488-521
  * LGL: /root/reference/pockit/lobatto/discretization.py:14-77,113-152,169-227,414-441
  * front/middle/back partitions and COO split by column:
    /root/reference/pockit/base/discretizationbase.py:10-38,199-329

Conventions (both schemes): a phase has ``L_m`` middle-stage nodes; node 0 is the "front"; LGL's
last node is the "back"; LGR states carry one extra slot (index ``L_m``) for the final point.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import scipy.sparse

from . import tables


def _ranges_disjoint(counts):
    r = np.cumsum(counts)
    return np.concatenate(([0], r[:-1])), r


def _ranges_shared(counts):
    l = np.concatenate(([0], np.cumsum(counts[:-1] - 1)))
    return l, l + counts


class Part:
    """front index | [l_m, r_m) middle range | back index   (IndexNode of the reference)."""

    def __init__(self, front, middle, back):
        self.front, self.back = front, back
        self.l_m, self.r_m = middle
        self.L_m = middle[1] - middle[0]
        self.f = front is not None
        self.b = back is not None
        self.m = slice(*middle)


def _split_by_column(csr, part):
    """COO entries (row-major, as CSR->COO yields them) split into front/middle/back by column."""
    coo = csr.tocoo()
    coo.eliminate_zeros()
    row, col, dat = coo.row.astype(np.int32), coo.col.astype(np.int32), coo.data.astype(np.float64)
    is_f = (col == part.front) if part.f else np.zeros(len(col), bool)
    is_b = ((col == part.back) if part.b else np.zeros(len(col), bool)) & ~is_f
    is_m = ~(is_f | is_b)

    def pick(mask):
        return SimpleNamespace(row=row[mask], col=col[mask], data=dat[mask], nnz=int(mask.sum()))

    return SimpleNamespace(f=pick(is_f), m=pick(is_m), b=pick(is_b))


def _block_matrix(blocks, row0, col0, shape):
    data, row, col = [], [], []
    for B, r0, c0 in zip(blocks, row0, col0):
        nr, nc = B.shape
        data.extend(B.flatten())
        row.extend(r0 + np.repeat(np.arange(nr), nc))
        col.extend(c0 + np.tile(np.arange(nc), nr))
    M = scipy.sparse.coo_array((data, (row, col)), shape=shape)
    M.eliminate_zeros()
    return M.tocsr()


class Mesh:
    def __init__(self, scheme, mesh, num_point, n_x, n_u):
        assert scheme in ("lgr", "lgl")
        self.scheme, self.mesh, self.num_point = scheme, mesh, num_point
        self.n_x, self.n_u = n_x, n_u
        K = np.asarray(num_point)
        width = np.diff(mesh)
        mid = (mesh[1:] + mesh[:-1]) / 2
        lgr = scheme == "lgr"

        # node ranges of each interval in the middle stage
        self.l_m, self.r_m = _ranges_disjoint(K) if lgr else _ranges_shared(K)
        self.L_m = int(self.r_m[-1])

        # variable layout [x_0 .. | u_0 ..] (t_0, t_f follow)
        if lgr:
            sizes = [self.L_m + 1] * n_x + [self.L_m] * n_u
        else:
            sizes = [self.L_m] * (n_x + n_u)
        self.l_v, self.r_v = (np.asarray(a, dtype=int) for a in _ranges_disjoint(np.array(sizes, dtype=int))) \
            if sizes else (np.array([], int), np.array([], int))

        # node positions (scaled to [0,1]) and quadrature weights
        t = np.zeros(self.L_m)
        w = np.zeros(self.L_m)
        for l, r, k, d, m in zip(self.l_m, self.r_m, K, width, mid):
            xk, wk = tables.lgr(int(k)) if lgr else tables.lgl(int(k))
            t[l:r] = xk * d / 2 + m
            if lgr:
                w[l:r] = wk * d / 2
            else:
                w[l:r] += wk * d / 2
        self.t_m, self.w_m = t, w

        # defect rows per state
        rows_per = K if lgr else K - 1
        L_d = int(np.sum(rows_per))
        self.l_d, self.r_d = _ranges_disjoint(np.full(n_x, L_d, dtype=np.int32)) if n_x else \
            (np.array([], int), np.array([0]))
        row0, _ = _ranges_disjoint(rows_per)
        if lgr:
            colT, _ = _ranges_shared(K + 1)
            Tblocks = [np.hstack((np.eye(int(k)), -np.ones((int(k), 1)))) for k in K]
            self.T_v = _block_matrix(Tblocks, row0, colT, (L_d, self.L_m + 1))
            Iblocks = [tables.I_lgr(int(k)) * d / 2 for k, d in zip(K, width)]
            self.I_m = _block_matrix(Iblocks, row0, self.l_m, (L_d, self.L_m))
            self.part_state = Part(0, (1, self.L_m), self.L_m)
            self.part_control = Part(0, (1, self.L_m), None)
            self.part_mstage = Part(0, (1, self.L_m), None)
            self.t_x = np.concatenate([self.t_m, [1.0]])
            self.l_x, self.r_x = _ranges_shared(K + 1)
            self.l_u, self.r_u = _ranges_disjoint(K)
        else:
            Tblocks = [np.hstack((np.eye(int(k) - 1), -np.ones((int(k) - 1, 1)))) for k in K]
            self.T_v = _block_matrix(Tblocks, row0, self.l_m, (L_d, self.L_m))
            Iblocks = [tables.I_lgl(int(k)) * d / 2 for k, d in zip(K, width)]
            self.I_m = _block_matrix(Iblocks, row0, self.l_m, (L_d, self.L_m))
            p = Part(0, (1, self.L_m - 1), self.L_m - 1)
            self.part_state = self.part_control = self.part_mstage = p
            self.t_x = self.t_m
            self.l_x, self.r_x = self.l_m, self.r_m
            self.l_u, self.r_u = self.l_m, self.r_m
        self.t_u = self.t_m
        self.T_coo = _split_by_column(self.T_v, self.part_state)
        self.I_coo = _split_by_column(self.I_m, self.part_mstage)

    def to_mstage(self, v):
        """[x_i(nodes) .. | u_j(nodes) ..]; LGR drops each state's final point (v2m_)."""
        if self.scheme == "lgl":
            return v
        L = self.L_m
        out = np.empty(L * (self.n_x + self.n_u))
        for i in range(self.n_x):
            out[i * L: (i + 1) * L] = v[self.l_v[i]: self.r_v[i] - 1]
        for i in range(self.n_x, self.n_x + self.n_u):
            out[i * L: (i + 1) * L] = v[self.l_v[i]: self.r_v[i]]
        return out
