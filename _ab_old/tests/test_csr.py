"""Triplet -> CSR map of the device-resident hand-off (pockit_amd/csr.py), on the reference's own golden
triplet lists: structure and summed values against scipy.sparse's COO -> CSR conversion."""
import os

import numpy as np
import pytest
import scipy.sparse

import models
from pockit_amd.csr import CsrMap

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", sorted(models.SMALL_CASES))
def test_csr_map_matches_scipy_on_golden_triplets(name):
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    n, m = int(gold["n"]), int(gold["m"])
    for rows, cols, vals, shape in ((gold["jr"], gold["jc"], gold["J"], (m, n)), (gold["hr"], gold["hc"], gold["H"], (n, n))):
        if len(rows) == 0:
            continue
        cm = CsrMap(rows, cols, shape)
        ref = scipy.sparse.coo_array((vals, (rows, cols)), shape=shape).tocsr()
        ref.sum_duplicates()
        ref.sort_indices()
        # scipy drops nothing here (explicit zeros stay), so the structures must be identical
        assert np.array_equal(cm.indptr, ref.indptr) and np.array_equal(cm.indices, ref.indices)
        got = cm.gather(vals)
        assert np.max(np.abs(got - ref.data)) <= 1e-13 * max(1.0, np.max(np.abs(ref.data)))
        assert cm.nnz == ref.nnz and cm.n_triplets == len(rows)
        assert sorted(cm.perm.tolist()) == list(range(len(rows)))
        if cm.seg is not None:
            assert cm.seg[0] == 0 and cm.seg[-1] == len(rows) and np.all(np.diff(cm.seg) > 0)
        assert (cm.to_scipy(got) != ref).nnz == 0 or np.allclose(cm.to_scipy(got).toarray(), ref.toarray(), atol=1e-13)


def test_csr_map_rejects_bad_patterns():
    with pytest.raises(ValueError):
        CsrMap([0, 1], [0], (2, 2))
    with pytest.raises(ValueError):
        CsrMap([], [], (2, 2))
    with pytest.raises(ValueError):
        CsrMap([0, 2], [0, 0], (2, 2))
    with pytest.raises(ValueError):
        CsrMap([0, 1], [0, -1], (2, 2))


def test_duplicates_are_summed_in_triplet_order():
    cm = CsrMap([1, 0, 1, 1], [0, 1, 0, 2], (2, 3))
    assert cm.indptr.tolist() == [0, 1, 3] and cm.indices.tolist() == [1, 0, 2]
    assert cm.perm.tolist() == [1, 0, 2, 3] and cm.seg.tolist() == [0, 1, 3, 4]
    assert cm.gather([1.0, 2.0, 4.0, 8.0]).tolist() == [2.0, 5.0, 8.0]
