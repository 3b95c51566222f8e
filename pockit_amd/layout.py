"""Index bookkeeping of one phase on one mesh, and the wave tiles the HIP kernels iterate over.

Host-side only (NumPy).  Restates the layout facts of the reference's ``Discretization`` objects
  * LGR: /root/reference/pockit/radau/discretization.py:44-87,117-140,199-257,488-521
  * LGL: /root/reference/pockit/lobatto/discretization.py:44-77,113-136,169-227,414-441
  * front / middle / back split by column: /root/reference/pockit/base/discretizationbase.py:258-314
in a form shaped for the GPU: instead of global CSR/COO matrices, each interval refers to a
*kind* = (K, front column dropped?, back column dropped?) whose small entry tables are shared by
all intervals of that kind, and consecutive intervals of one kind are grouped into wave tiles.

Conventions: a phase has ``L_m`` middle-stage nodes.  LGR: interval j owns nodes
[lm_j, lm_j+K_j), its defects use the extra end slot lm_j+K_j of the state arrays (length L_m+1).
LGL: interval j owns nodes [lm_j, lm_j+K_j) with the last one shared with interval j+1.
Node 0 is the *front* node; the *back* slot is state index L_m (LGR) or node L_m-1 (LGL).
"""
from __future__ import annotations

import numpy as np

from . import collocation

WAVE = 64
BIG_MAX = 256      # an interval with more than 64 points is handled by a whole workgroup (PK_BIG code objects); up to BIG_MAX its
                   # per-node values are staged in LDS, beyond that in a slot of device memory (no limit on num_point, as
                   # in the reference: radau/discretization.py:488-521)


class Kind:
    """Entry tables of one distinct interval pattern (unit width; kernels scale by d_j/2)."""

    def __init__(self, scheme, K, drop_front, drop_back):
        self.K, self.drop_front, self.drop_back = K, drop_front, drop_back
        lgr = scheme == "lgr"
        A = collocation.lgr_integration_matrix(K) if lgr else collocation.lgl_integration_matrix(K)
        self.R = A.shape[0]
        self.full = A                                            # R x K, used by eval_g
        ir, ic, iv = [], [], []
        for r in range(self.R):
            for c in range(K):
                if A[r, c] == 0.0:
                    continue                                      # reference eliminates exact zeros
                if (drop_front and c == 0) or (drop_back and not lgr and c == K - 1):
                    continue
                ir.append(r); ic.append(c); iv.append(A[r, c])
        self.I_r = np.array(ir, dtype=np.int32)
        self.I_c = np.array(ic, dtype=np.int32)
        self.I_v = np.array(iv, dtype=np.float64)
        # translation block rows: +1 at local column r, -1 at the end column (K for LGR, K-1 for LGL)
        end = K if lgr else K - 1
        tr, tc, tv = [], [], []
        for r in range(self.R):
            if not (drop_front and r == 0):
                tr.append(r); tc.append(r); tv.append(1.0)
            if not drop_back:
                tr.append(r); tc.append(end); tv.append(-1.0)
        self.T_r = np.array(tr, dtype=np.int32)
        self.T_c = np.array(tc, dtype=np.int32)
        self.T_v = np.array(tv, dtype=np.float64)
        self.nnzI, self.nnzT = len(iv), len(tv)


class MeshLayout:
    def __init__(self, scheme, mesh, num_point, n_x, n_u):
        assert scheme in ("lgr", "lgl")
        self.scheme, self.mesh = scheme, np.asarray(mesh, dtype=np.float64)
        self.K = np.asarray(num_point, dtype=np.int64)
        self.n_x, self.n_u = n_x, n_u
        lgr = scheme == "lgr"
        N = self.N = len(self.K)
        self.width = np.diff(self.mesh)
        mid = (self.mesh[1:] + self.mesh[:-1]) / 2
        self.stride = self.K if lgr else self.K - 1               # nodes an interval adds
        self.R = self.stride                                       # defect rows per interval
        self.lm = np.concatenate(([0], np.cumsum(self.stride[:-1]))).astype(np.int64)
        self.rm = self.lm + self.K
        self.L_m = int(self.rm[-1])
        self.ld = np.concatenate(([0], np.cumsum(self.R[:-1]))).astype(np.int64)
        self.L_d = int(np.sum(self.R))
        self.state_len = self.L_m + 1 if lgr else self.L_m
        sizes = np.array([self.state_len] * n_x + [self.L_m] * n_u, dtype=np.int64)
        self.r_v = np.cumsum(sizes)
        self.l_v = self.r_v - sizes
        self.L = int(self.r_v[-1]) + 2
        self.l_d = np.arange(n_x, dtype=np.int64) * self.L_d
        self.r_d = self.l_d + self.L_d
        # middle range of the middle-stage nodes, front/back presence
        self.has_back = not lgr
        self.mid_lo, self.mid_hi = 1, (self.L_m if lgr else self.L_m - 1)
        self.L_mid = max(self.mid_hi - self.mid_lo, 0)
        self.back_slot = self.L_m if lgr else self.L_m - 1        # state-array index of the back slot

        tau = np.zeros(self.L_m)
        w = np.zeros(self.L_m)
        for j in range(N):
            k = int(self.K[j])
            xk, wk = collocation.lgr_nodes_weights(k) if lgr else collocation.lgl_nodes_weights(k)
            sl = slice(self.lm[j], self.rm[j])
            tau[sl] = xk * self.width[j] / 2 + mid[j]
            if lgr:
                w[sl] = wk * self.width[j] / 2
            else:
                w[sl] += wk * self.width[j] / 2
        self.tau, self.w = tau, w
        self.t_x = np.concatenate([tau, [1.0]]) if lgr else tau

        # kinds
        self.kinds: list[Kind] = []
        index = {}

        def kind_id(K, df, db):
            key = (int(K), bool(df), bool(db))
            if key not in index:
                index[key] = len(self.kinds)
                self.kinds.append(Kind(scheme, *key))
            return index[key]

        self.kid = np.array([kind_id(self.K[j], j == 0, j == N - 1) for j in range(N)], dtype=np.int32)
        self.kid_full = np.array([kind_id(self.K[j], False, False) for j in range(N)], dtype=np.int32)
        nnzI = np.array([self.kinds[k].nnzI for k in self.kid], dtype=np.int64)
        nnzT = np.array([self.kinds[k].nnzT for k in self.kid], dtype=np.int64)
        self.offI = np.concatenate(([0], np.cumsum(nnzI[:-1])))
        self.offT = np.concatenate(([0], np.cumsum(nnzT[:-1])))
        self.nnzI_mid, self.nnzT_mid = int(nnzI.sum()), int(nnzT.sum())

        # front / back column entries (already scaled by d/2, exactly as the reference: (A*d)/2)
        A0 = self.kinds[self.kid_full[0]].full * self.width[0] / 2
        rows = [r for r in range(A0.shape[0]) if A0[r, 0] != 0.0]
        self.If_row = np.array(rows, dtype=np.int64)
        self.If_val = A0[rows, 0] if rows else np.zeros(0)
        self.Tf_row = np.array([0], dtype=np.int64)
        self.Tf_val = np.array([1.0])
        last = N - 1
        Rl = int(self.R[last])
        self.Tb_row = self.ld[last] + np.arange(Rl, dtype=np.int64)
        self.Tb_val = np.full(Rl, -1.0)
        if lgr:
            self.Ib_row, self.Ib_val = np.zeros(0, np.int64), np.zeros(0)
        else:
            Al = self.kinds[self.kid_full[last]].full * self.width[last] / 2
            kk = int(self.K[last]) - 1
            rows = [r for r in range(Al.shape[0]) if Al[r, kk] != 0.0]
            self.Ib_row = self.ld[last] + np.array(rows, dtype=np.int64)
            self.Ib_val = Al[rows, kk] if rows else np.zeros(0)

    # ------------------------------------------------------------------ structure helpers
    def I_mid_structure(self):
        """(defect row within a state, column node) of every middle entry of I_m, reference order."""
        rows, cols = [], []
        for j in range(self.N):
            k = self.kinds[self.kid[j]]
            rows.append(self.ld[j] + k.I_r)
            cols.append(self.lm[j] + k.I_c)
        return np.concatenate(rows), np.concatenate(cols)

    def T_mid_structure(self):
        """(defect row, state-array column, value) of every middle entry of T_v, reference order."""
        rows, cols, vals = [], [], []
        for j in range(self.N):
            k = self.kinds[self.kid[j]]
            rows.append(self.ld[j] + k.T_r)
            cols.append(self.lm[j] + k.T_c)
            vals.append(k.T_v)
        return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)

    # ------------------------------------------------------------------ tiles
    def tiles(self, intervals_per_wave=None):
        """Group consecutive intervals of one kind into wave tiles of at most 64 nodes.

        Returns an int32 array [ntile, 8]: j0, nj, kid, kid_full, q0, r0, offI, offT."""
        out = []
        j = 0
        while j < self.N:
            K = int(self.K[j])
            st = int(self.stride[j])
            cap = max(WAVE // K, 1) if self.scheme == "lgr" else max((WAVE - 1) // st, 1)     # (K > 64: one interval, a block)
            if intervals_per_wave:
                cap = max(1, min(cap, int(intervals_per_wave)))
            nj = 1
            while j + nj < self.N and nj < cap and self.kid[j + nj] == self.kid[j]:
                nj += 1
            out.append((j, nj, self.kid[j], self.kid_full[j], self.lm[j], self.ld[j], self.offI[j], self.offT[j]))
            j += nj
        return np.array(out, dtype=np.int32).reshape(-1, 8)
