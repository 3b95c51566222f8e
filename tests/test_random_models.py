"""Seeded random models (tests/random_models.py) against the REFERENCE's own callback vectors (tests/golden/random, written by
tests/golden/make_random.py in the build container): on the CPU the oracle and the NumPy execution of the product's plan, on
the GPU (``-m gpu``) the HIP kernels through the C ABI -- callbacks, stand-alone kernels, the one-launch cycle, and the same
with the derivative set forced into groups.  Structures exactly, values to 1e-11."""
import importlib
import os

import numpy as np
import pytest

import random_models as rm

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(scheme, seed) for scheme, seeds in rm.SEEDS.items() for seed in seeds]
TOL = 1e-11


def close(a, b, what, tol=TOL):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    assert a.shape == b.shape, what
    if a.size:
        err = np.max(np.abs(a - b))
        assert err <= tol * max(1.0, np.max(np.abs(b))), f"{what}: err {err:.3e} (scale {np.max(np.abs(b)):.3e})"


def gold(scheme, seed):
    return np.load(os.path.join(HERE, "golden", "random", f"{scheme}_{seed}.npz"))


def check_structure(system, g):
    jr, jc = system.jacobianstructure()
    hr, hc = system.hessianstructure()
    assert np.array_equal(jr, g["jr"]) and np.array_equal(jc, g["jc"]), "Jacobian structure"
    assert np.array_equal(hr, g["hr"]) and np.array_equal(hc, g["hc"]), "Hessian structure"
    for key in ("v_lb", "v_ub", "c_lb", "c_ub"):
        assert np.array_equal(getattr(system, key), g[key]), key


@pytest.mark.parametrize("scheme,seed", CASES)
def test_oracle_and_plan_match_the_reference_on_a_random_model(scheme, seed):
    from plan_interp import Interp

    g = gold(scheme, seed)
    x, lam, sigma = g["x"], g["lam"], float(g["sigma"])
    ref, _ = rm.random_model(importlib.import_module(f"oracle.{scheme}"), seed, scheme)
    check_structure(ref, g)
    close(ref.objective(x.copy()), g["f"], "oracle f")
    close(ref.gradient(x.copy()), g["grad"], "oracle grad")
    close(ref.constraints(x.copy()), g["g"], "oracle g")
    close(ref.jacobian(x.copy()), g["J"], "oracle J")
    close(ref.hessian(x.copy(), lam, sigma), g["H"], "oracle H")
    system, _ = rm.random_model(importlib.import_module(f"pockit_amd.{scheme}"), seed, scheme)
    check_structure(system, g)
    it = Interp(system.plan, x, lam, sigma)
    close(it.objective(), g["f"], "plan f")
    close(it.gradient(), g["grad"], "plan grad")
    close(it.constraints(), g["g"], "plan g")
    close(it.jacobian(), g["J"], "plan J")
    close(it.hessian(), g["H"], "plan H")


@pytest.mark.gpu
@pytest.mark.parametrize("cap,ipw", [(None, None), ("3", None), ("3", "1"), (None, "2")])
@pytest.mark.parametrize("scheme,seed", CASES)
def test_gpu_matches_the_reference_on_a_random_model(scheme, seed, cap, ipw, monkeypatch):
    if cap:
        monkeypatch.setenv("POCKIT_AMD_GROUP_CAP", cap)      # the derivative set in groups of three
    if ipw:
        monkeypatch.setenv("POCKIT_AMD_IPW", ipw)            # one / two intervals per wave: many ragged tiles
    g = gold(scheme, seed)
    x, lam, sigma = g["x"].copy(), g["lam"].copy(), float(g["sigma"])
    system, _ = rm.random_model(importlib.import_module(f"pockit_amd.{scheme}"), seed, scheme)
    check_structure(system, g)
    close(system.objective(x), g["f"], "f")
    close(system.gradient(x), g["grad"], "grad f")
    close(system.constraints(x), g["g"], "g")
    close(system.jacobian(x), g["J"], "J")
    close(system.hessian(x, lam, sigma), g["H"], "H")
    ev = system.evaluator
    close(ev.objective_direct(x), g["f"], "f (stand-alone)")
    close(ev.gradient_direct(x), g["grad"], "grad f (stand-alone)")
    close(ev.constraints_direct(x), g["g"], "g (pk_g)")
    close(ev.jacobian_direct(x), g["J"], "J (pk_jac)")
    close(ev.hessian_direct(x, lam, sigma), g["H"], "H (pk_hess)")
    for a, b, what in zip(ev.cycle(x, lam, sigma), (g["f"], g["grad"], g["g"], g["J"], g["H"]), ("f", "grad", "g", "J", "H")):
        close(a, b, what + " (cycle)")
    assert np.array_equal(x, g["x"]), "x must not be written"
    system._invalidate()
