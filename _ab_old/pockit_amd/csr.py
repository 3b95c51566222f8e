"""Triplet list -> CSR map (host side of the device-resident CSR hand-off).

The NLP callbacks return values in the reference's triplet order (duplicates included, systembase.py:671-693,
811-835).  A GPU linear solver wants CSR.  The (row, col) sort is a property of the mesh, so it is done once
here; per iterate the device only gathers (``pk_csr``): ``csr[p] = sum(triplets[perm[seg[p]:seg[p+1]]])``.
"""
from __future__ import annotations

import numpy as np


class CsrMap:
    """CSR structure of a triplet pattern plus the gather map that fills its values.

    ``indptr`` (n_rows + 1), ``indices`` (nnz): the CSR structure, columns ascending within a row;
    ``perm`` (n_triplets): triplet indices in (row, col) order, ties in triplet order (stable: repeated
    entries are summed in the order the reference lists them); ``seg`` (nnz + 1): runs of ``perm`` per CSR
    entry, ``None`` when no entry repeats."""

    def __init__(self, rows, cols, shape):
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        n_rows, n_cols = (int(v) for v in shape)
        if rows.shape != cols.shape or rows.ndim != 1:
            raise ValueError("rows and cols must be one-dimensional arrays of the same length")
        if len(rows) == 0:
            raise ValueError("empty pattern")
        if len(rows) > np.iinfo(np.int32).max:
            raise ValueError("pattern too large for 32-bit indices")
        if rows.min() < 0 or rows.max() >= n_rows or cols.min() < 0 or cols.max() >= n_cols:
            raise ValueError("triplet index outside the matrix")
        key = rows * n_cols + cols
        order = np.argsort(key, kind="stable")
        sorted_key = key[order]
        first = np.concatenate(([True], sorted_key[1:] != sorted_key[:-1]))
        starts = np.flatnonzero(first)
        unique_key = sorted_key[starts]
        self.shape = (n_rows, n_cols)
        self.n_triplets = len(rows)
        self.nnz = len(starts)
        self.perm = order.astype(np.int32)
        self.seg = None if self.nnz == self.n_triplets else np.concatenate((starts, [len(rows)])).astype(np.int32)
        self.indices = (unique_key % n_cols).astype(np.int32)
        self.indptr = np.searchsorted(unique_key // n_cols, np.arange(n_rows + 1)).astype(np.int32)

    def gather(self, triplet_values):
        """Host execution of the gather (what ``pk_csr`` does on the device, up to the association of the sums
        inside a run); used by the tests."""
        v = np.asarray(triplet_values, dtype=np.float64)[self.perm]
        if self.seg is None:
            return v
        out = np.zeros(self.nnz)
        return np.add.reduceat(v, self.seg[:-1], out=out) if len(v) else out

    def to_scipy(self, values):
        import scipy.sparse

        return scipy.sparse.csr_array((np.asarray(values), self.indices, self.indptr), shape=self.shape)
