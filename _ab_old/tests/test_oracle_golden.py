"""Pin the oracle (CPU restatement) to the golden vectors captured from the reference.

Fixtures: tests/golden/small/*.npz, tests/golden/tables.npz, tests/golden/full.json, all produced
by tests/golden/make_golden.py (imports /root/reference in the build container).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import models
import oracle.lobatto
import oracle.radau
from oracle import tables

HERE = os.path.dirname(os.path.abspath(__file__))
NS = {"radau": oracle.radau, "lobatto": oracle.lobatto}
TOL = 1e-12  # oracle vs reference run the same NumPy expressions; differences are rounding only


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    if a.size:
        assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))


@pytest.mark.parametrize("name", sorted(models.SMALL_CASES))
def test_small_case_matches_reference(name):
    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, guess = builder(NS[scheme], **kw)
    x, lam, sigma = gold["x"], gold["lam"], float(gold["sigma"])
    close(models.pack_guess(system, guess), gold["x0"])
    assert int(system.L) == int(gold["n"]) and len(system.c_lb) == int(gold["m"])
    assert np.array_equal(system.l_p, gold["l_p"]) and np.array_equal(system.r_p, gold["r_p"])
    assert (system.l_s, system.r_s) == (int(gold["l_s"]), int(gold["r_s"]))
    for k in ("v_lb", "v_ub", "c_lb", "c_ub"):
        assert np.array_equal(getattr(system, k), gold[k])
    x_before = x.copy()
    close(system.objective(x), gold["f"])
    close(system.gradient(x), gold["grad"])
    close(system.constraints(x), gold["g"])
    jr, jc = system.jacobianstructure()
    assert np.array_equal(jr, gold["jr"]) and np.array_equal(jc, gold["jc"])
    close(system.jacobian(x), gold["J"])
    hr, hc = system.hessianstructure()
    assert np.array_equal(hr, gold["hr"]) and np.array_equal(hc, gold["hc"])
    close(system.hessian(x, lam, sigma), gold["H"])
    r, c = system.hessianstructure_o()
    assert np.array_equal(r, gold["hro"]) and np.array_equal(c, gold["hco"])
    close(system.hessian_o(x), gold["Ho"])
    r, c = system.hessianstructure_c()
    assert np.array_equal(r, gold["hrc"]) and np.array_equal(c, gold["hcc"])
    close(system.hessian_c(x, lam), gold["Hc"])
    assert np.array_equal(x, x_before)  # the oracle never writes to the caller's x


def test_tables_match_reference():
    gold = np.load(os.path.join(HERE, "golden", "tables.npz"))
    for K in range(1, 13):
        x, w = tables.lgr(K)
        close(x, gold[f"lgr_x_{K}"], 1e-14)
        close(w, gold[f"lgr_w_{K}"], 1e-14)
        close(tables.I_lgr(K), gold[f"lgr_I_{K}"], 1e-14)
        x, w = tables.lgl(K)
        close(x, gold[f"lgl_x_{K}"], 1e-14)
        close(w, gold[f"lgl_w_{K}"], 1e-14)
        if K >= 2:
            close(tables.I_lgl(K), gold[f"lgl_I_{K}"], 1e-14)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, np.int64)).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(models.FULL_CASES))
def test_full_size_config_matches_reference_summary(name):
    """BASELINE.json configs at full size: structure hashes + value checksums/samples."""
    gold = json.load(open(os.path.join(HERE, "golden", "full.json")))[name]
    builder, scheme, kw = models.FULL_CASES[name]
    system, _, guess = builder(NS[scheme], **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    assert int(system.L) == gold["n"] and len(system.c_lb) == gold["m"]
    jr, jc = system.jacobianstructure()
    hr, hc = system.hessianstructure()
    assert (len(jr), len(hr)) == (gold["nnz_J"], gold["nnz_H"])
    assert (_sha(jr), _sha(jc), _sha(hr), _sha(hc)) == (gold["sha_jr"], gold["sha_jc"], gold["sha_hr"], gold["sha_hc"])

    def check(v, S):
        v = np.asarray(v)
        assert len(v) == S["len"]
        scale = max(1.0, S["max"])
        assert np.max(np.abs(v[np.array(S["idx"])] - np.array(S["samples"]))) <= 1e-11 * scale
        assert abs(v.sum() - S["sum"]) <= 1e-11 * max(1.0, S["sumabs"])

    check(x, gold["x"])
    check(lam, gold["lam"])
    assert abs(system.objective(x) - gold["f"]) <= 1e-11 * max(1.0, abs(gold["f"]))
    check(system.gradient(x), gold["grad"])
    check(system.constraints(x), gold["g"])
    check(system.jacobian(x), gold["J"])
    check(system.hessian(x, lam, sigma), gold["H"])


@pytest.mark.parametrize("name", sorted(models.ERROR_CASES))
def test_mesh_error_estimation_and_refinement_match_reference(name):
    """T_x_aug / I_f_aug, per-interval verdicts and refined meshes of phasebase.py:1339-1437,1522-1617."""
    from oracle import refine

    builder, scheme, kw = models.ERROR_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "error", name + ".npz"))
    system, phases, _ = builder(NS[scheme], **kw)
    system.prepare()
    x = gold["x"]
    s = x[system.l_s: system.r_s]
    for k, p in enumerate(phases):
        xp = x[system.l_p[k]: system.r_p[k]]
        T, I = refine.error_data(p, xp, s)
        close(T, gold[f"T_{k}"], 1e-13)
        close(I, gold[f"I_{k}"], 1e-13)
        for tag, (atol, rtol) in (("a", (1e-3, 1e-3)), ("b", (1e-7, 1e-6))):
            assert np.array_equal(refine.check_intervals(p, T, I, atol, rtol, 1e-4), gold[f"ok_{tag}_{k}"])
            mesh0, K0 = p._mesh.copy(), p._num_point.copy()
            p.refine_continuous(NS[scheme].Variable(p, xp.copy()), s if len(s) else None, atol, rtol, num_point_min=3,
                                num_point_max=7, mesh_length_min=1e-3, mesh_length_max=1.0)
            assert np.allclose(p._mesh, gold[f"mesh_{tag}_{k}"], rtol=0, atol=1e-15)
            assert np.array_equal(p._num_point, gold[f"K_{tag}_{k}"])
            p.set_discretization(mesh0, K0)
