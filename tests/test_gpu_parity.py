"""GPU parity: the HIP evaluator (through the C ABI) against golden vectors of the reference and
against the oracle, on identical inputs.  Tolerance: structures exact,
values |a-b| <= 1e-11 * max(1, max|b|) per array (fp64; libm ulps, fused multiply-add,
reduction order)."""
import os

import numpy as np
import pytest

import models

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-11


def close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    if a.size:
        err = np.max(np.abs(a - b))
        assert err <= tol * max(1.0, np.max(np.abs(b))), f"{what}: err {err:.3e}"


def _ns(scheme, pkg):
    import importlib

    return importlib.import_module(f"{pkg}.{scheme}")


def _supported(name):
    return True


@pytest.mark.parametrize("name", sorted(n for n in models.SMALL_CASES if _supported(n)))
def test_small_case_matches_reference_golden(name):
    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, _ = builder(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = gold["x"], gold["lam"], float(gold["sigma"])
    x_before = x.copy()
    jr, jc = system.jacobianstructure()
    hr, hc = system.hessianstructure()
    assert np.array_equal(jr, gold["jr"]) and np.array_equal(jc, gold["jc"])
    assert np.array_equal(hr, gold["hr"]) and np.array_equal(hc, gold["hc"])
    close(system.objective(x), gold["f"], what="f")
    close(system.gradient(x), gold["grad"], what="grad")
    close(system.constraints(x), gold["g"], what="g")
    close(system.jacobian(x), gold["J"], what="J")
    close(system.hessian(x, lam, sigma), gold["H"], what="H")
    close(system.hessian_o(x), gold["Ho"], what="Ho")
    close(system.hessian_c(x, lam), gold["Hc"], what="Hc")
    assert np.array_equal(x, x_before), "x must not be written"


@pytest.mark.parametrize("ipw", [1, 2, 3, 64])
@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=37, num_point=5)),
                                  ("brachistochrone", "lobatto", dict(mesh=23, num_point=6)),
                                  ("two_stage_rocket", "radau", dict(mesh=40, num_point=3)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=19, num_point=4)),
                                  ("humanoid_wbc", "radau", dict(mesh=9, num_point=7))])
def test_tilings_match_oracle(case, ipw, monkeypatch):
    """Every intervals-per-wave tiling (incl. ragged last tiles) gives the oracle's result."""
    monkeypatch.setenv("POCKIT_AMD_IPW", str(ipw))
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    close(system.objective(x), ref.objective(x), what="f")
    close(system.gradient(x), ref.gradient(x), what="grad")
    close(system.constraints(x), ref.constraints(x), what="g")
    close(system.jacobian(x), ref.jacobian(x), what="J")
    close(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma), what="H")
    # the fused cycle path (one node evaluation for f, grad f, g, J) must give the same five outputs
    f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
    close(f, ref.objective(x), what="cycle f")
    close(grad, ref.gradient(x), what="cycle grad")
    close(g, ref.constraints(x), what="cycle g")
    close(J, ref.jacobian(x), what="cycle J")
    close(H, ref.hessian(x, lam, sigma), what="cycle H")


def test_ragged_mesh_matches_oracle():
    """hp-style mesh: every interval its own width and polynomial order (K = 1 .. 9)."""
    rng = np.random.default_rng(5)
    K = rng.integers(1, 10, size=31).tolist()
    mesh = np.concatenate(([0.0], np.cumsum(rng.uniform(0.2, 1.0, size=31)))).tolist()
    kw = dict(mesh=mesh, num_point=K)
    system, _, guess = models.brachistochrone(_ns("radau", "pockit_amd"), **kw)
    ref, _, _ = models.brachistochrone(_ns("radau", "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    close(system.constraints(x), ref.constraints(x), what="g")
    close(system.gradient(x), ref.gradient(x), what="grad")
    close(system.jacobian(x), ref.jacobian(x), what="J")
    close(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma), what="H")


def test_full_size_configs_match_reference_summary():
    """BASELINE.json configs at full size against the reference's checksums / strided samples."""
    import json

    full = json.load(open(os.path.join(HERE, "golden", "full.json")))
    for name in ("C2_brach_lgr_200x8", "S_brach_lgr_1250x8", "C3_quad_lgr_2000x6", "C4_rocket_lgr_2x1000x4"):
        gold = full[name]
        builder, scheme, kw = models.FULL_CASES[name]
        system, _, guess = builder(_ns(scheme, "pockit_amd"), **kw)
        x, lam, sigma = models.bench_inputs(system, guess)

        def check(v, S, what):
            v = np.asarray(v)
            assert len(v) == S["len"], what
            scale = max(1.0, S["max"])
            assert np.max(np.abs(v[np.array(S["idx"])] - np.array(S["samples"]))) <= TOL * scale, what
            assert abs(v.sum() - S["sum"]) <= 1e-10 * max(1.0, S["sumabs"]), what

        assert (system.plan.nnz_J, system.plan.nnz_H) == (gold["nnz_J"], gold["nnz_H"])
        assert abs(system.objective(x) - gold["f"]) <= TOL * max(1.0, abs(gold["f"]))
        check(system.gradient(x), gold["grad"], name + " grad")
        check(system.constraints(x), gold["g"], name + " g")
        check(system.jacobian(x), gold["J"], name + " J")
        check(system.hessian(x, lam, sigma), gold["H"], name + " H")
        system._invalidate()


def test_repeated_evaluation_with_changing_x():
    """The in-launch finalize (per-tile partials handed to the last-arriving workgroup) must see the
    *current* launch's partials: alternate between two points on one evaluator."""
    system, _, guess = models.two_stage_rocket(_ns("radau", "pockit_amd"), 300, 4)
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 300, 4)
    x1, lam, sigma = models.bench_inputs(system, guess)
    x2 = x1 * (1.0 + 0.05 * np.random.default_rng(7).uniform(-1, 1, x1.shape))
    want = {id(x): (ref.objective(x), ref.gradient(x)) for x in (x1, x2)}
    for x in (x1, x2, x1, x2, x2, x1):
        for _ in range(3):
            close(system.objective(x), want[id(x)][0], what="f")
            close(system.gradient(x), want[id(x)][1], what="grad")
        f, grad, _, _, _ = system.evaluator.cycle(x, lam, sigma)
        close(f, want[id(x)][0], what="cycle f")
        close(grad, want[id(x)][1], what="cycle grad")
