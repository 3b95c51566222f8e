"""GPU parity over EVERY example model of the reference (the 33 programs of /root/reference/examples), each at its own
mesh: the model is rebuilt on the GPU box from tests/golden/examples/<name>.model.json through this package's modeling API
(the description was written in the build container by running the reference's program against this package,
tests/golden/make_examples.py) and its five callbacks -- through the C ABI, on the HIP kernels -- are compared with the
REFERENCE's own outputs on the same x, lambda, sigma (tests/golden/examples/<name>.npz): structures exactly, values to
1e-11.  Also the one-launch cycle, the compact layouts (against the scatter-added reference
triplets) and that x is never written."""
import glob
import json
import os

import numpy as np
import pytest

import model_io

HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(HERE, "golden", "examples")
NAMES = sorted(os.path.basename(p)[:-11] for p in glob.glob(os.path.join(EX, "*.model.json")))
TOL = 1e-11


def close(a, b, what, tol=TOL):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    assert a.shape == b.shape, what
    if a.size:
        err = np.max(np.abs(a - b))
        assert err <= tol * max(1.0, np.max(np.abs(b))), f"{what}: err {err:.3e} (scale {np.max(np.abs(b)):.3e})"


def dense(rows, cols, vals, shape):
    import scipy.sparse as sps

    return sps.coo_matrix((vals, (rows, cols)), shape=shape).tocsr()


def load(name):
    with open(os.path.join(EX, name + ".model.json")) as fh:
        return model_io.load_system(json.load(fh)), np.load(os.path.join(EX, name + ".npz"))


def test_all_example_models_have_fixtures():
    assert len(NAMES) == 33


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_example_model_matches_the_reference_on_the_gpu(name):
    system, gold = load(name)
    x, lam, sigma = gold["x"].copy(), gold["lam"].copy(), float(gold["sigma"])
    x_before = x.copy()
    jr, jc = system.jacobianstructure()
    hr, hc = system.hessianstructure()
    assert np.array_equal(jr, gold["jr"]) and np.array_equal(jc, gold["jc"]), "Jacobian structure"
    assert np.array_equal(hr, gold["hr"]) and np.array_equal(hc, gold["hc"]), "Hessian structure"
    for key in ("v_lb", "v_ub", "c_lb", "c_ub"):
        assert np.array_equal(getattr(system, key), gold[key]), key
    # the five callbacks as a solver calls them (fused x-kernel + Hessian kernel behind the shim)
    close(system.objective(x), gold["f"], "f")
    close(system.gradient(x), gold["grad"], "grad f")
    close(system.constraints(x), gold["g"], "g")
    close(system.jacobian(x), gold["J"], "J")
    close(system.hessian(x, lam, sigma), gold["H"], "H")
    ev = system.evaluator
    assert not ev.src.spilling_kernels, f"kernels spill vector registers: {ev.src.spilling_kernels}"
    # the stand-alone kernel of each callback
    close(ev.objective_direct(x), gold["f"], "f (pk_int)")
    close(ev.gradient_direct(x), gold["grad"], "grad f (pk_grad)")
    close(ev.constraints_direct(x), gold["g"], "g (pk_g)")
    close(ev.jacobian_direct(x), gold["J"], "J (pk_jac)")
    close(ev.hessian_direct(x, lam, sigma), gold["H"], "H (pk_hess)")
    # all five from ONE launch (pk_cycle), host arrays in and out
    f, grad, g, J, H = ev.cycle(x, lam, sigma)
    close(f, gold["f"], "f (pk_cycle)")
    close(grad, gold["grad"], "grad f (pk_cycle)")
    close(g, gold["g"], "g (pk_cycle)")
    close(J, gold["J"], "J (pk_cycle)")
    close(H, gold["H"], "H (pk_cycle)")
    assert np.array_equal(x, x_before), "x must not be written"
    # compact layouts: equal matrices after scatter-add
    n, m = x.size, lam.size
    if ev.src.compact:
        system.set_hessian_layout("compact")
        cr, cc = system.hessianstructure()
        Hc = system.hessian(x, lam, sigma)
        system.set_hessian_layout("reference")
        ref = dense(gold["hr"], gold["hc"], gold["H"], (n, n))
        got = dense(cr, cc, Hc, (n, n))
        close(got.toarray(), ref.toarray(), "compact Hessian")
    if ev.src.compact_j:
        system.set_jacobian_layout("compact")
        cr, cc = system.jacobianstructure()
        Jc = system.jacobian(x)
        system.set_jacobian_layout("reference")
        ref = dense(gold["jr"], gold["jc"], gold["J"], (m, n))
        got = dense(cr, cc, Jc, (m, n))
        close(got.toarray(), ref.toarray(), "compact Jacobian")
