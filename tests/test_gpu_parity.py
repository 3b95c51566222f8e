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
    ev = system.evaluator                      # the stand-alone kernels of each callback (no x cache, no fusion)
    close(ev.objective_direct(x), gold["f"], what="f direct")
    close(ev.gradient_direct(x), gold["grad"], what="grad direct")
    close(ev.constraints_direct(x), gold["g"], what="g direct")
    close(ev.jacobian_direct(x), gold["J"], what="J direct")
    close(ev.hessian_direct(x, lam, sigma), gold["H"], what="H direct")
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


@pytest.mark.parametrize("ipw", [1, 3, 7, 64])
@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=83, num_point=1)),      # R = nnzI = 1
                                  ("brachistochrone", "lobatto", dict(mesh=83, num_point=2)),    # R = 1
                                  ("two_stage_rocket", "radau", dict(mesh=41, num_point=1)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=50, num_point=2))])
def test_unit_divisor_tiles_match_oracle(case, ipw, monkeypatch):
    """Tiles whose per-interval divisors (defect rows, integration / translation entries) are 1 -- LGR K = 1 and
    LGL K = 2 -- holding several intervals: the quotient p // 1 has no 32-bit magic number (magic_div)."""
    monkeypatch.setenv("POCKIT_AMD_IPW", str(ipw))
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    tiles = system.evaluator.tables.tiles
    if ipw >= 3:
        assert tiles["nj"].max() >= 3 and (tiles["magicR"][tiles["nj"] > 0] == 0).all()
    x, lam, sigma = models.bench_inputs(system, guess)
    close(system.objective(x), ref.objective(x), what="f")
    close(system.gradient(x), ref.gradient(x), what="grad")
    close(system.constraints(x), ref.constraints(x), what="g")
    close(system.jacobian(x), ref.jacobian(x), what="J")
    close(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma), what="H")
    f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
    close(g, ref.constraints(x), what="cycle g")
    close(J, ref.jacobian(x), what="cycle J")
    close(H, ref.hessian(x, lam, sigma), what="cycle H")
    ev = system.evaluator
    close(ev.constraints_direct(x), ref.constraints(x), what="g direct")
    close(ev.jacobian_direct(x), ref.jacobian(x), what="J direct")
    close(ev.hessian_direct(x, lam, sigma), ref.hessian(x, lam, sigma), what="H direct")


# Tolerance at high orders: the ORACLE restates the reference's np.roots-based collocation tables, which lose digits with
# K (quadrature weights against the product's Newton-polished ones: 2e-12 at K = 12, 8e-11 at K = 16, 4e-9 at K = 20,
# 2e-7 at K = 24; the product's node residuals stay at 1e-14) -- the comparison can only be as tight as the oracle.
@pytest.mark.parametrize("cap", ["", "64"])
@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=13, num_point=12), TOL),
                                  ("brachistochrone", "lobatto", dict(mesh=9, num_point=16), 2e-9),
                                  ("planar_quadrotor", "radau", dict(mesh=21, num_point=10), TOL),
                                  ("humanoid_wbc", "radau", dict(mesh=7, num_point=9), TOL),
                                  ("two_stage_rocket", "radau", dict(mesh=5, num_point=16), 2e-9),
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.2, 0.5, 0.6, 1.0], num_point=[9, 16, 20, 7]), 1e-7),
                                  ("planar_quadrotor", "lobatto", dict(mesh=3, num_point=24), 1e-5)])
def test_high_order_tiles_match_oracle(case, cap, monkeypatch):
    """Patterns with 9 <= K <= 16 stage their (up to 256-entry) tables in LDS over four entries per lane; K > 16 (and
    POCKIT_AMD_TAB_CAP=64: every K > 8) reads them from global memory in the unstaged variant of phase B."""
    if cap:
        monkeypatch.setenv("POCKIT_AMD_TAB_CAP", cap)
    bname, scheme, kw, tol = case

    def check(a, b, what=""):              # this test's tolerance
        close(a, b, tol=tol, what=what)

    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    check(system.objective(x), ref.objective(x), what="f")
    check(system.gradient(x), ref.gradient(x), what="grad")
    check(system.constraints(x), ref.constraints(x), what="g")
    check(system.jacobian(x), ref.jacobian(x), what="J")
    check(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma), what="H")
    f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
    check(g, ref.constraints(x), what="cycle g")
    check(J, ref.jacobian(x), what="cycle J")
    check(H, ref.hessian(x, lam, sigma), what="cycle H")
    ev = system.evaluator
    check(ev.constraints_direct(x), ref.constraints(x), what="g direct")
    check(ev.jacobian_direct(x), ref.jacobian(x), what="J direct")
    check(ev.hessian_direct(x, lam, sigma), ref.hessian(x, lam, sigma), what="H direct")


@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=[0, 0.4, 1.0], num_point=[64, 33])),
                                  ("brachistochrone", "lobatto", dict(mesh=[0, 0.3, 0.7, 1.0], num_point=[64, 40, 64])),
                                  ("planar_quadrotor", "radau", dict(mesh=[0, 0.5, 1.0], num_point=[64, 64])),
                                  ("two_stage_rocket", "lobatto", dict(mesh=2, num_point=64))])
def test_maximum_supported_order_matches_the_plan_interpreter(case):
    """The documented limit -- 64 points per interval, one interval filling a wavefront (the LGR end slot then lies
    beyond the wave's lanes) -- against the NumPy execution of the same plan with the product's own tables
    (tests/plan_interp.py): the oracle's np.roots-based tables are useless at this order (weights off by more than 1e-3)."""
    from plan_interp import Interp

    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    it = Interp(system.plan, x, lam, sigma)
    want = dict(f=it.objective(), grad=it.gradient(), g=it.constraints(), J=it.jacobian(), H=it.hessian())
    close(system.objective(x), want["f"], what="f")
    close(system.gradient(x), want["grad"], what="grad")
    close(system.constraints(x), want["g"], what="g")
    close(system.jacobian(x), want["J"], what="J")
    close(system.hessian(x, lam, sigma), want["H"], what="H")
    f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
    close(g, want["g"], what="cycle g")
    close(J, want["J"], what="cycle J")
    close(H, want["H"], what="cycle H")
    ev = system.evaluator
    close(ev.constraints_direct(x), want["g"], what="g direct")
    close(ev.jacobian_direct(x), want["J"], what="J direct")
    close(ev.hessian_direct(x, lam, sigma), want["H"], what="H direct")


@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=[0, 0.2, 0.5, 0.7, 1.0], num_point=[6, 100, 8, 65])),
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.3, 0.5, 1.0], num_point=[63, 64, 7])),
                                  ("brachistochrone", "lobatto", dict(mesh=[0, 0.3, 0.6, 1.0], num_point=[66, 5, 130])),
                                  ("planar_quadrotor", "radau", dict(mesh=[0, 0.5, 0.6, 1.0], num_point=[128, 6, 200])),
                                  ("planar_quadrotor", "lobatto", dict(mesh=[0, 0.5, 1.0], num_point=[256, 7])),
                                  ("two_stage_rocket", "radau", dict(mesh=[0, 0.4, 1.0], num_point=[70, 9])),
                                  ("humanoid_wbc", "radau", dict(mesh=[0, 0.5, 1.0], num_point=[4, 80])),
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.3, 1.0], num_point=[300, 5])),      # > 256: staged in device memory
                                  ("brachistochrone", "lobatto", dict(mesh=[0, 0.2, 0.6, 1.0], num_point=[257, 70, 513])),
                                  ("planar_quadrotor", "lobatto", dict(mesh=[0, 0.5, 1.0], num_point=[7, 400])),
                                  ("planar_quadrotor", "radau", dict(mesh=[0, 0.4, 1.0], num_point=[264, 263])),   # (K + 1 = 264 / 265 augmented nodes)
                                  ("planar_quadrotor", "radau", dict(mesh=[0, 0.5, 0.6, 1.0], num_point=[128, 6, 200]), "mfma"),
                                  ("brachistochrone", "lobatto", dict(mesh=[0, 0.3, 0.6, 1.0], num_point=[66, 5, 130]), "mfma"),
                                  # derivative set evaluated in GROUPS (forced by a group size of 3) on such intervals
                                  ("planar_quadrotor", "radau", dict(mesh=[0, 0.5, 0.6, 1.0], num_point=[128, 6, 200]), "groups"),
                                  ("brachistochrone", "lobatto", dict(mesh=[0, 0.3, 0.6, 1.0], num_point=[66, 5, 130]), "groups"),
                                  ("two_stage_rocket", "radau", dict(mesh=[0, 0.4, 1.0], num_point=[70, 9]), "groups"),
                                  ("humanoid_wbc", "radau", dict(mesh=[0, 0.5, 1.0], num_point=[4, 80]), "groups"),
                                  ("planar_quadrotor", "lobatto", dict(mesh=[0, 0.5, 1.0], num_point=[7, 400]), "groups")])
def test_intervals_with_more_points_than_a_wavefront_has_lanes(case, monkeypatch):
    """num_point > 64 (the reference has no limit, radau/discretization.py:488-521): such an interval is evaluated
    by a whole workgroup (PK_BIG code objects), next to ordinary wave tiles; up to 256 points its per-node values are
    staged in LDS, beyond that in the interval's slot of a staging buffer in device memory.  Reference: the NumPy execution of the same
    plan with the product's own tables (pinned by the multiprecision tables of tests/test_tables_hiprec.py up to K = 128;
    the oracle's np.roots-based tables carry no digits at these orders).  "mfma": the same with the interval's products on
    the fp64 matrix cores (POCKIT_AMD_BIG_MFMA=1; measured slower than the VALU form, kept as a switch)."""
    from plan_interp import Interp

    if len(case) == 4 and case[3] == "mfma":
        monkeypatch.setenv("POCKIT_AMD_BIG_MFMA", "1")
    if len(case) == 4 and case[3] == "groups":
        monkeypatch.setenv("POCKIT_AMD_GROUP_CAP", "3")
    bname, scheme, kw = case[:3]
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    it = Interp(system.plan, x, lam, sigma)
    want = dict(f=it.objective(), grad=it.gradient(), g=it.constraints(), J=it.jacobian(), H=it.hessian())
    ev = system.evaluator
    assert ev.src.big == (max(kw["num_point"]) > 64)
    assert ev.src.grouped == (len(case) == 4 and case[3] == "groups")
    f, grad, g, J, H = ev.cycle(x, lam, sigma)                       # pk_cycle: the three roles of every big block
    close(f, want["f"], what="cycle f")
    close(grad, want["grad"], what="cycle grad")
    close(g, want["g"], what="cycle g")
    close(J, want["J"], what="cycle J")
    close(H, want["H"], what="cycle H")
    close(system.objective(x), want["f"], what="f")                  # the host shim: pk_xall (+ pk_fin), pk_hess
    close(system.gradient(x), want["grad"], what="grad")
    close(system.constraints(x), want["g"], what="g")
    close(system.jacobian(x), want["J"], what="J")
    close(system.hessian(x, lam, sigma), want["H"], what="H")
    close(ev.objective_direct(x), want["f"], what="f direct")        # single callbacks (served by the fused x-kernel)
    close(ev.gradient_direct(x), want["grad"], what="grad direct")
    close(ev.constraints_direct(x), want["g"], what="g direct")
    close(ev.jacobian_direct(x), want["J"], what="J direct")
    close(ev.hessian_direct(x, lam, sigma), want["H"], what="H direct")
    # two-launch cycle (pk_xall, then pk_hess with the reductions), unsplit and split x-part -- at a point the context has not
    # seen, two launches FIRST: an entry a kernel failed to write would show what an older iterate left in the landing block,
    # which at the same x looks right (tools/two_launch_probe2.py; the same-x form of this check was blind to that)
    x2 = x * (1.0 + 1.0e-3 * np.random.default_rng(11).uniform(-1.0, 1.0, x.shape))
    ev.set_cycle_mode(False)
    try:
        f2, grad2, g2, J2, H2 = [np.array(v) for v in ev.cycle(x2, lam, sigma)]
    finally:
        ev.set_cycle_mode(True)
    f1, grad1, g1, J1, H1 = [np.array(v) for v in ev.cycle(x2, lam, sigma)]
    it2 = Interp(system.plan, x2, lam, sigma)
    close(g1, it2.constraints(), what="g at the second point")
    close(J1, it2.jacobian(), what="J at the second point")
    if ev.src.cycle_subs:      # every pass a workgroup of its own in pk_cycle / pk_hess, a loop in pk_xall: the compiler
        for a, b, what in ((J2, J1, "J"), (H2, H1, "H"), (g2, g1, "g"), (grad2, grad1, "grad")):     # contracts the same expressions differently
            close(a, b, tol=1e-13, what="two launches against one: " + what)
        close(f2, f1, tol=1e-13, what="two launches against one: f")
    else:
        assert np.array_equal(J2, J1) and np.array_equal(H2, H1) and np.array_equal(g2, g1) and np.array_equal(grad2, grad1)
        assert f2 == f1
    system.set_hessian_layout("compact")                             # compact layout: pk_hessc walks such an interval 64 nodes at a time
    close(system.hessian(x, lam, sigma), it.hessian_compact(), what="compact H")
    system.set_hessian_layout("reference")
    system.set_jacobian_layout("compact")                            # compact Jacobian: pk_jacc gives such an interval a workgroup too
    close(system.jacobian(x), it.jacobian_compact(), what="compact J")
    close(system.evaluator.jacobian_compact(x), it.jacobian_compact(), what="compact J, one-shot")
    system.set_jacobian_layout("reference")
    close(system.jacobian(x), want["J"], what="J after the layout switch")
    # mesh error estimation: an interval with K + 1 > 64 augmented nodes is walked by a whole workgroup of pk_err
    import plan_interp

    data = ev.mesh_error(x)
    want_e = plan_interp.mesh_error(system.plan, x)
    for k, (T, I) in enumerate(want_e):
        close(data[k][0], T, what=f"error T phase {k}")
        close(data[k][1], I, what=f"error I phase {k}")
    assert isinstance(system.check_continuous(guess), bool)


@pytest.mark.parametrize("scheme,kw", [("radau", dict(mesh=(0, 0.2, 0.5, 1), num_point=(3, 80, 4))),
                                       ("lobatto", dict(mesh=(0, 0.2, 0.5, 1), num_point=(3, 80, 4))),
                                       ("radau", dict(mesh=(0, 0.4, 0.6, 1), num_point=(70, 5, 97))),
                                       ("lobatto", dict(mesh=(0, 0.5, 1), num_point=(258, 3)))])
def test_model_nonlinear_in_the_integrals_on_intervals_with_more_than_64_points(scheme, kw):
    """Objective and system constraints nonlinear in the integrals (outer-product Hessian blocks, easyderiv.py:323-459; the
    integrals are needed before every other kernel) on a mesh with workgroup-wide intervals: the integral prepass (pk_int)
    and the auxiliary pass (pk_aux) walk such an interval with a whole workgroup too.  Reference: the NumPy execution of
    the same plan (tests/plan_interp.py)."""
    from plan_interp import Interp

    system, _, guess = models.derivative_model(_ns(scheme, "pockit_amd"), **kw)
    assert system.plan.outer and system.evaluator.src.big
    x, lam, sigma = models.bench_inputs(system, guess)
    it = Interp(system.plan, x, lam, sigma)
    want = dict(f=it.objective(), grad=it.gradient(), g=it.constraints(), J=it.jacobian(), H=it.hessian())
    ev = system.evaluator
    f, grad, g, J, H = ev.cycle(x, lam, sigma)
    for name, got in (("f", f), ("grad", grad), ("g", g), ("J", J), ("H", H)):
        close(got, want[name], what="cycle " + name)
    close(system.objective(x), want["f"], what="f")
    close(system.gradient(x), want["grad"], what="grad")
    close(system.constraints(x), want["g"], what="g")
    close(system.jacobian(x), want["J"], what="J")
    close(system.hessian(x, lam, sigma), want["H"], what="H")
    close(ev.objective_direct(x), want["f"], what="f direct")
    close(ev.gradient_direct(x), want["grad"], what="grad direct")
    close(ev.constraints_direct(x), want["g"], what="g direct")
    close(ev.jacobian_direct(x), want["J"], what="J direct")
    close(ev.hessian_direct(x, lam, sigma), want["H"], what="H direct")
    system.set_jacobian_layout("compact")          # (dense t0 / tf / static columns: the contracted rows of a workgroup-wide interval)
    close(system.jacobian(x), it.jacobian_compact(), what="compact J")
    system.set_jacobian_layout("reference")
    # the same model on an ordinary mesh agrees with the oracle (pins the interpreter's outer-product path)
    small, _, sg = models.derivative_model(_ns(scheme, "pockit_amd"))
    ref, _, rg = models.derivative_model(_ns(scheme, "oracle"))
    xs, ls, ss = models.bench_inputs(small, sg)
    close(Interp(small.plan, xs, ls, ss).hessian(), ref.hessian(xs, ls, ss), what="interpreter vs oracle H")


def test_hessian_stored_into_the_landing_array_by_its_kernel_equals_the_copied_one():
    """Host shim, small Hessians: pk_hess stores into the pinned landing array itself (host option "hess_direct", on by default up
    to the copy-kernel threshold) instead of into device memory with a copy behind it.  The landing arrays are recycled and never
    cleared: every entry must be written by the kernel, on every call -- checked with inputs that change from call to call."""
    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), 150, 6)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    lib, h = ev.ctx.lib, ev.ctx.handle
    rng = np.random.default_rng(5)
    inputs = [(x * (1 + 1e-3 * rng.standard_normal(x.size)), lam * rng.standard_normal(lam.size), float(rng.uniform(0.5, 2))) for _ in range(4)]
    got = {}
    for direct in (1, 0, 1):
        ev.ctx.check(lib.pk_set_host_option(h, b"hess_direct", direct))
        for k, (xk, lk, sk) in enumerate(inputs):
            system.objective(xk)
            H = np.array(system.hessian(xk, lk, sk))
            assert np.all(np.isfinite(H))
            if (k, ) in got:
                assert np.array_equal(H, got[(k, )]), f"hess_direct={direct}, input {k}"
            got[(k, )] = H
    ref, _, _ = models.planar_quadrotor(_ns("radau", "oracle"), 150, 6)
    xk, lk, sk = inputs[2]
    close(got[(2, )], ref.hessian(xk, lk, sk), what="H")
    ev.ctx.check(lib.pk_set_host_option(h, b"hess_direct", 1))


@pytest.mark.parametrize("mesh", [12, 301])          # (a small system: kernels store into the landing block; a larger one: copies)
def test_line_search_trial_points_then_an_accepted_point(mesh):
    """What IPOPT's backtracking does to the callbacks: rejected trial points ask for the objective and the constraints only,
    the accepted point for everything.  Behind a rejected point grad f / J are no longer copied ahead (host option
    "adaptive_prefetch") but fetched when asked for -- every value must still be the one of ITS x."""
    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), mesh, 6)
    ref, _, _ = models.planar_quadrotor(_ns("radau", "oracle"), mesh, 6)
    x, lam, sigma = models.bench_inputs(system, guess)
    rng = np.random.default_rng(mesh)
    for it in range(6):
        for _ in range(it % 3):                                   # 0, 1 or 2 rejected trial points
            xt = x * (1 + 1e-3 * rng.standard_normal(x.size))
            close(system.objective(xt), ref.objective(xt), what="trial f")
            close(system.constraints(xt), ref.constraints(xt), what="trial g")
        xa = x * (1 + 1e-3 * rng.standard_normal(x.size))
        close(system.objective(xa), ref.objective(xa), what="f")
        g = system.constraints(xa)
        grad = system.gradient(xa)
        J = system.jacobian(xa)
        H = system.hessian(xa, lam, sigma)
        close(grad, ref.gradient(xa), what="grad")
        close(g, ref.constraints(xa), what="g")
        close(J, ref.jacobian(xa), what="J")
        close(H, ref.hessian(xa, lam, sigma), what="H")


def test_prepared_x_cache_is_dropped_by_calls_that_reuse_the_context_buffers():
    """objective / gradient / constraints / jacobian / hessian on x1 serve from ONE upload of x1; any other entry
    point that uploads a different x (mesh error, the one-launch cycle, the *_direct and CSR calls) in between must
    not leave them answering for that other x."""
    system, _, guess = models.two_stage_rocket(_ns("radau", "pockit_amd"), 60, 4)
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 60, 4)
    x1, lam, sigma = models.bench_inputs(system, guess)
    x2 = x1 * (1.0 + 0.05 * np.random.default_rng(11).uniform(-1, 1, x1.shape))
    ev = system.evaluator
    want = dict(f=ref.objective(x1), grad=ref.gradient(x1), g=ref.constraints(x1), J=ref.jacobian(x1),
                H=ref.hessian(x1, lam, sigma))
    others = [lambda: ev.mesh_error(x2), lambda: ev.cycle(x2, lam, sigma), lambda: ev.gradient_direct(x2),
              lambda: ev.jacobian_direct(x2), lambda: ev.jacobian_csr(x2), lambda: ev.hessian_csr(x2, lam, sigma),
              lambda: ev.hessian_compact(x2, lam, sigma), lambda: ev.constraints_direct(x2),
              lambda: ev.objective_direct(x2), lambda: ev.hessian_direct(x2, lam, sigma)]
    for other in others:
        close(system.gradient(x1), want["grad"], what="grad before")
        other()
        close(system.hessian(x1, lam, sigma), want["H"], what="H after another x went through the context")
        other()
        close(system.jacobian(x1), want["J"], what="J after")
        other()
        close(system.gradient(x1), want["grad"], what="grad after")
        close(system.constraints(x1), want["g"], what="g after")
        close(system.objective(x1), want["f"], what="f after")


def test_pageable_result_targets_of_a_c_abi_caller():
    """pk_set_result_targets with ordinary (pageable) host arrays, as a caller of the C ABI may pass them: the results
    are copied there (a kernel stores only into targets the device can see: pinned memory of pk_host_alloc, or the
    context's own buffers)."""
    import ctypes as C

    from pockit_amd import runtime

    system, _, guess = models.two_stage_rocket(_ns("radau", "pockit_amd"), 30, 4)
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 30, 4)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev, plan = system.evaluator, system.plan
    lib, h, dp = ev.ctx.lib, ev.ctx.handle, runtime.as_dp
    for host_direct in (False, True):
        ev.set_host_mode(True, host_direct)
        out = [np.full(k, np.nan) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
        ev.ctx.check(lib.pk_set_result_targets(h, *[dp(a) for a in out]))
        ev.ctx.check(lib.pk_prepare_x(h, dp(x)))
        for what in range(4):
            ev.ctx.check(lib.pk_fetch(h, what, None))
        ev.ctx.check(lib.pk_eval_hess_prepared(h, dp(lam), C.c_double(sigma), None))
        ev.ctx.check(lib.pk_set_result_targets(h, None, None, None, None, None))
        close(out[0][0], ref.objective(x), what="f")
        close(out[1], ref.gradient(x), what="grad")
        close(out[2], ref.constraints(x), what="g")
        close(out[3], ref.jacobian(x), what="J")
        close(out[4], ref.hessian(x, lam, sigma), what="H")
    ev.set_host_mode(True, False)
    ev._invalidate_x()
    close(system.gradient(x), ref.gradient(x), what="grad afterwards")


@pytest.mark.parametrize("scheme", ["radau", "lobatto"])
def test_elementary_functions_of_a_user_model_match_oracle(scheme):
    """A user model may use any elementary function SymPy differentiates and the reference prints (tan, asin, acos, atan,
    atan2, sinh ... atanh, exp, log, sqrt, rational / negative powers): the generated device code must agree with the
    oracle's NumPy execution in structure and values."""
    system, _, guess = models.elementary_functions_model(_ns(scheme, "pockit_amd"))
    ref, _, _ = models.elementary_functions_model(_ns(scheme, "oracle"))
    x, lam, sigma = models.bench_inputs(system, guess)
    jr, jc = ref.jacobianstructure()
    hr, hc = ref.hessianstructure()
    pjr, pjc = system.jacobianstructure()
    phr, phc = system.hessianstructure()
    assert np.array_equal(jr, pjr) and np.array_equal(jc, pjc) and np.array_equal(hr, phr) and np.array_equal(hc, phc)
    close(system.objective(x), ref.objective(x.copy()), what="f")
    close(system.gradient(x), ref.gradient(x.copy()), what="grad")
    close(system.constraints(x), ref.constraints(x.copy()), what="g")
    close(system.jacobian(x), ref.jacobian(x.copy()), what="J")
    close(system.hessian(x, lam, sigma), ref.hessian(x.copy(), lam, sigma), what="H")
    f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
    close(J, ref.jacobian(x.copy()), what="cycle J")
    close(H, ref.hessian(x.copy(), lam, sigma), what="cycle H")


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
    for name in ("C2_brach_lgr_200x8", "S_brach_lgr_1250x8", "C3_quad_lgr_2000x6", "C4_rocket_lgr_2x1000x4",
                 "C5_humanoid_lgr_5000x8"):
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
        # the benchmarked kernel itself: all five outputs of the ONE-launch cycle (pk_cycle) at the full size
        f1, grad1, g1, J1, H1 = system.evaluator.cycle(x, lam, sigma)
        assert abs(f1 - gold["f"]) <= TOL * max(1.0, abs(gold["f"])), name + " cycle f"
        check(grad1, gold["grad"], name + " cycle grad")
        check(g1, gold["g"], name + " cycle g")
        check(J1, gold["J"], name + " cycle J")
        check(H1, gold["H"], name + " cycle H")
        # ... and EVERY entry of every array, callbacks and one-launch cycle, against the oracle (itself pinned to the same
        # reference summaries at these sizes, tests/test_oracle_golden.py): a wrong run that the 64 samples and the sum miss
        # cannot pass
        ref, _, _ = builder(_ns(scheme, "oracle"), **kw)
        want = (ref.objective(x), ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma))
        jr, jc = system.jacobianstructure()
        rr, rc = ref.jacobianstructure()
        assert np.array_equal(jr, rr) and np.array_equal(jc, rc), name + " J structure"
        hr, hc = system.hessianstructure()
        rr, rc = ref.hessianstructure()
        assert np.array_equal(hr, rr) and np.array_equal(hc, rc), name + " H structure"
        got = (system.objective(x), system.gradient(x), system.constraints(x), system.jacobian(x),
               system.hessian(x, lam, sigma))
        for a, b, what in zip(got, want, ("f", "grad", "g", "J", "H")):
            close(a, b, what=f"{name} callbacks {what} (all entries)")
        for a, b, what in zip((f1, grad1, g1, J1, H1), want, ("f", "grad", "g", "J", "H")):
            close(a, b, what=f"{name} cycle {what} (all entries)")
        system._invalidate()


def test_cycle_is_bit_stable_at_forty_thousand_nodes():
    """The single-launch cycle at the humanoid's full size, 150 times on the same inputs: every launch must reproduce
    the stand-alone Hessian kernel bit for bit.  (Regression: the 16-byte streaming store is an asm statement, so the
    compiler did not insert the wait state gfx9-class hardware needs between a store of more than 64 bits and a VALU
    write of its data registers; with several waves per SIMD about 7 % of the launches had 16 entries of one segment
    replaced by the next iteration's products.  The 12k-node configurations -- one wave per SIMD -- never showed it.)"""
    builder, scheme, kw = models.FULL_CASES["C5_humanoid_lgr_5000x8"]
    system, _, guess = builder(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    H0 = np.array(ev.hessian_direct(x, lam, sigma))
    f0, grad0, g0, J0, Hc = (np.array(v) for v in ev.cycle(x, lam, sigma))
    assert np.array_equal(Hc, H0)
    bad = []
    for rep in range(150):
        f1, grad1, g1, J1, H1 = ev.cycle(x, lam, sigma)
        if not (np.array_equal(H1, H0) and np.array_equal(J1, J0) and np.array_equal(g1, g0)
                and np.array_equal(grad1, grad0) and f1 == f0):
            bad.append(rep)
    assert not bad, f"launches that differ from the first: {bad}"
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


def test_two_shards_on_one_gpu_reassemble_to_the_oracle():
    """The mesh-interval sharding data path on real kernels: two rank-local evaluators (each holding
    half of every phase's tiles) run on one GPU; summing their zero-initialised outputs -- what the
    RCCL all-reduce does across GPUs -- must reproduce the unsharded oracle result."""
    import ctypes as C

    import torch

    from pockit_amd.evaluator import Evaluator
    from pockit_amd.sharding import tile_filter

    system, _, guess = models.two_stage_rocket(_ns("radau", "pockit_amd"), 90, 4)
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 90, 4)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    world = 2
    evs, Is, outs = [], [], []
    for r in range(world):
        ev = Evaluator(plan, intervals_per_wave=3, tile_filter=tile_filter(r, world))
        I = torch.zeros(max(len(plan.I_syms), 1), dtype=torch.float64, device=dev)
        ev.ctx.check(ev.ctx.lib.pk_set_shard(ev.ctx.handle, int(r != 0), 1, C.c_void_p(I.data_ptr())))
        evs.append(ev)
        Is.append(I)
        outs.append({k: torch.zeros(n, dtype=torch.float64, device=dev) for k, n in
                     (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))})
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    torch.cuda.synchronize()             # torch's zero-fills run on its own stream; the contexts use theirs
    for ev in evs:
        ev.ctx.check(ev.ctx.lib.pk_eval_integrals_dev(ev.ctx.handle, p(dx), None))
        ev.sync()
    total = sum(Is)                      # the all-reduce of the integrals
    for I in Is:
        I.copy_(total)
    torch.cuda.synchronize()
    for ev, o in zip(evs, outs):
        lib, h = ev.ctx.lib, ev.ctx.handle
        ev.ctx.check(lib.pk_eval_f_from_integrals_dev(h, p(dx), p(o["f"]), None))
        ev.ctx.check(lib.pk_eval_grad_dev(h, p(dx), p(o["grad"]), None))
        ev.ctx.check(lib.pk_eval_g_dev(h, p(dx), p(o["g"]), None))
        ev.ctx.check(lib.pk_eval_jac_dev(h, p(dx), p(o["J"]), None))
        ev.ctx.check(lib.pk_eval_hess_dev(h, p(dx), p(dlam), float(sigma), p(o["H"]), None))
        ev.sync()
    close(outs[0]["f"].cpu().numpy()[0], ref.objective(x), what="f")
    close(outs[1]["f"].cpu().numpy()[0], ref.objective(x), what="f (rank 1)")
    for k, want in (("grad", ref.gradient(x)), ("g", ref.constraints(x)), ("J", ref.jacobian(x)),
                    ("H", ref.hessian(x, lam, sigma))):
        close(sum(o[k] for o in outs).cpu().numpy(), want, what=k)
        # disjoint support: no position is written by both shards
        both = (outs[0][k] != 0) & (outs[1][k] != 0)
        assert int(both.sum()) <= (plan.n_s + 2 * len(plan.phase_plans) if k == "grad" else 0), k
    for ev in evs:
        ev.close()


def test_scipy_trust_constr_solves_lqr_on_the_gpu_evaluator():
    """IPOPT-free end-to-end check (cyipopt is not installed): the SciPy adapter drives the GPU callbacks
    to the LQR optimum (README.md:95-118 model); objective compared with the Riccati solution."""
    from scipy.integrate import solve_ivp

    from pockit_amd.optimizer import scipy as scipy_solver

    ns = _ns("lobatto", "pockit_amd")
    system, (phase,), guess = models.lqr(ns, 6, 6)
    guess = [ns.constant_guess(phase, 0.0), [0.0]]
    (var, s), res = scipy_solver.solve(system, guess, {"maxiter": 200, "gtol": 1e-10, "xtol": 1e-12})
    a, b, q, r, sw = -1.0, 1.0, 1.0, 0.1, 1.0
    ric = solve_ivp(lambda t, P: -(2 * a * P - b * b * P * P / r + q), (1.0, 0.0), [sw / 2], rtol=1e-11, atol=1e-12)
    optimum = ric.y[0, -1] * 1.0 ** 2
    assert abs(res.fun - optimum) <= 2e-6 * max(1.0, abs(optimum))
    assert abs(var.x[0][-1] - s[0]) < 1e-9 and abs(var.x[0][0] - 1.0) < 1e-12


@pytest.mark.parametrize("scheme,mesh,num_point", [("radau", 10, 8), ("lobatto", 6, 6)])
def test_scipy_trust_constr_solves_the_brachistochrone_to_the_cycloid_time(scheme, mesh, num_point):
    """A second IPOPT-free end-to-end solve, nonlinear this time (free final time, path bounds, trigonometric dynamics): SciPy's
    trust-constr on the GPU callbacks (objective, gradient, constraints, Jacobian, Hessians of objective and constraints) must
    reach the analytic optimum of the brachistochrone (0, 0) -> (2, 2): the cycloid's descent time theta sqrt(R / g)."""
    from scipy.optimize import brentq

    from pockit_amd.optimizer import scipy as scipy_solver

    ns = _ns(scheme, "pockit_amd")
    system, (phase,), guess = models.brachistochrone(ns, mesh, num_point)
    (var,), res = scipy_solver.solve(system, guess, {"maxiter": 400, "gtol": 1e-9, "xtol": 1e-11})
    theta = brentq(lambda t: (t - np.sin(t)) / (1 - np.cos(t)) - 1.0, 0.1, 2 * np.pi - 0.1)      # x_f / y_f = 1
    radius = 2.0 / (1 - np.cos(theta))
    optimum = theta * np.sqrt(radius / 9.81)
    assert res.status in (1, 2) and res.constr_violation < 1e-10
    assert abs(res.fun - optimum) <= 1e-7, (res.fun, optimum)
    assert abs(var.t_f - optimum) <= 1e-7                                                        # (the objective is the final time)
    assert abs(var.x[0][-1] - 2.0) < 1e-12 and abs(var.x[1][-1] - 2.0) < 1e-12


@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=37, num_point=5)),
                                  ("brachistochrone", "lobatto", dict(mesh=23, num_point=6)),
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.1, 0.15, 0.5, 0.9, 1.0], num_point=[3, 7, 2, 5, 1])),
                                  ("two_stage_rocket", "radau", dict(mesh=40, num_point=3)),
                                  ("two_stage_rocket", "lobatto", dict(mesh=11, num_point=4)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=19, num_point=4)),
                                  ("planar_quadrotor", "radau", dict(mesh=300, num_point=6)),
                                  ("humanoid_wbc", "radau", dict(mesh=9, num_point=7)),
                                  ("humanoid_wbc", "lobatto", dict(mesh=6, num_point=5))])
def test_compact_hessian_equals_coalesced_oracle(case):
    """pk_hessc: one value per distinct position; scatter-added it must equal the scatter-add of the
    oracle's (= the reference's) duplicate-laden triplet list."""
    import scipy.sparse as ssp

    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    n = system.plan.n
    hr, hc = ref.hessianstructure()
    want = ssp.coo_array((ref.hessian(x, lam, sigma), (hr, hc)), shape=(n, n)).tocsr()
    system.set_hessian_layout("compact")
    cr, cc = system.hessianstructure()
    vals = system.hessian(x, lam, sigma)
    assert len(vals) == len(cr) < len(hr) or bname == "lqr"
    got = ssp.coo_array((vals, (cr, cc)), shape=(n, n)).tocsr()
    diff = abs(got - want)
    scale = max(1.0, abs(want).max())
    assert (diff.max() if diff.nnz else 0.0) <= TOL * scale
    system.set_hessian_layout("reference")
    close(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma), what="reference layout still served")


def test_compact_hessian_rides_on_the_prepared_x():
    """hessian() in the compact layout follows the prepared-x protocol (x of the iterate already on the device, multipliers
    staged, values by DMA into a pinned array of the caller's): interleaved with the other callbacks on two iterates it
    returns exactly what the one-shot entry point pk_eval_hessc computes, in arrays that stay valid."""
    import ctypes as C

    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), mesh=60, num_point=6)
    x1, lam, sigma = models.bench_inputs(system, guess)
    x2 = x1 * (1 + 1e-3)
    ev = system.evaluator
    lib, h = ev.ctx.lib, ev.ctx.handle

    def one_shot(x, lam_, sg):
        out = np.empty(system.plan.nnz_Hc)
        ev.ctx.check(lib.pk_eval_hessc(h, x.ctypes.data_as(C.POINTER(C.c_double)), lam_.ctypes.data_as(C.POINTER(C.c_double)),
                                       C.c_double(sg), out.ctypes.data_as(C.POINTER(C.c_double))))
        ev._invalidate_x()
        return out

    want = {(k, sg): one_shot(x, lam, sg) for k, x in ((1, x1), (2, x2)) for sg in (sigma, 0.5)}
    system.set_hessian_layout("compact")
    try:
        f1 = system.objective(x1)
        a = system.hessian(x1, lam, sigma)
        g1 = system.constraints(x1)
        b = system.hessian(x1, lam, 0.5)
        c = system.hessian(x2, lam, sigma)                      # a new x straight into the Hessian
        j2 = system.jacobian(x2)
        d = system.hessian(x2, lam, 0.5)
        assert np.array_equal(a, want[(1, sigma)]) and np.array_equal(b, want[(1, 0.5)])
        assert np.array_equal(c, want[(2, sigma)]) and np.array_equal(d, want[(2, 0.5)])
        assert a is not b and not np.shares_memory(a, b)
        ev.zero_copy = True
        e = system.hessian(x1, lam, sigma)
        assert np.array_equal(e, want[(1, sigma)]) and np.array_equal(a, want[(1, sigma)])
    finally:
        ev.zero_copy = False
        system.set_hessian_layout("reference")
    close(system.hessian(x1, lam, sigma), system.evaluator.hessian_direct(x1, lam, sigma), what="reference layout afterwards")
    assert np.isfinite(f1) and np.all(np.isfinite(g1)) and np.all(np.isfinite(j2))


# ---------------------------------------------------------------------------------------------------------
# mesh error estimation on device + hp-refinement
ERR_TOLS = {"a": (1e-3, 1e-3), "b": (1e-7, 1e-6)}
REFINE_KW = dict(num_point_min=3, num_point_max=7, mesh_length_min=1e-3, mesh_length_max=1.0)


def _values(system, phases, ns, x):
    plan = system.plan
    value = [ns.Variable(p, x[plan.l_p[k]: plan.r_p[k]].copy()) for k, p in enumerate(phases)]
    if system.n_s:
        value.append(x[plan.l_s: plan.r_s].copy())
    return value


@pytest.mark.parametrize("name", sorted(models.ERROR_CASES))
def test_mesh_error_kernel_matches_reference_golden(name):
    """pk_err (T_aug x, dt I_aug f per interval) against the reference's _error_estimation_data_continuous, the
    check against its verdicts, and one refine_continuous sweep (new mesh + adapted values) end to end."""
    builder, scheme, kw = models.ERROR_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "error", name + ".npz"))
    ns = _ns(scheme, "pockit_amd")
    system, phases, _ = builder(ns, **kw)
    x = gold["x"]
    data = system.evaluator.mesh_error(x)
    for k, (T, I) in enumerate(data):
        close(T, gold[f"T_{k}"], what=f"T phase {k}")
        close(I, gold[f"I_{k}"], what=f"I phase {k}")
    for tag, (atol, rtol) in ERR_TOLS.items():
        expect = all(bool(np.all(gold[f"ok_{tag}_{k}"])) for k in range(len(phases)))
        assert system.check_continuous(_values(system, phases, ns, x), atol, rtol, 1e-4) == expect
        for k, p in enumerate(phases):      # phase-level entry point
            s = x[system.plan.l_s: system.plan.r_s] if system.n_s else None
            v = ns.Variable(p, x[system.plan.l_p[k]: system.plan.r_p[k]].copy())
            assert p.check_continuous(v, s, atol, rtol, 1e-4) == bool(np.all(gold[f"ok_{tag}_{k}"]))
    for tag, (atol, rtol) in ERR_TOLS.items():
        system, phases, _ = builder(ns, **kw)
        value = _values(system, phases, ns, x)
        out = system.refine_continuous(value if len(value) > 1 else value[0], atol, rtol, **REFINE_KW)
        changed = any(not np.array_equal(p._num_point, gold[f"K_{tag}_{k}"]) or len(p._mesh) != len(gold[f"mesh_{tag}_{k}"])
                      for k, p in enumerate(phases))
        assert not changed
        if all(bool(np.all(gold[f"ok_{tag}_{k}"])) for k in range(len(phases))):
            continue            # nothing to refine at this tolerance: the input is returned as is
        out = out if isinstance(out, list) else [out]
        for k, p in enumerate(phases):
            assert np.allclose(p._mesh, gold[f"mesh_{tag}_{k}"], rtol=0, atol=1e-15)
            close(out[k].data, gold[f"adapt_{tag}_{k}"], 1e-10, what=f"adapt {tag} phase {k}")
        # the system is usable on the new discretization
        x_new = np.concatenate([v.data for v in out[: len(phases)]] + ([np.asarray(out[-1])] if system.n_s else []))
        assert len(x_new) == system.L and np.isfinite(system.objective(x_new))


def test_mesh_error_full_size_matches_oracle():
    """Quadrotor 2000 x 6 (LGR) and a ragged LGL hp mesh against the oracle's per-interval restatement."""
    from oracle import refine as oref

    rng = np.random.default_rng(11)
    K = rng.integers(2, 12, size=57).tolist()
    mesh = np.concatenate(([0.0], np.cumsum(rng.uniform(0.2, 1.0, size=57)))).tolist()
    for builder, scheme, kw in ((models.planar_quadrotor, "radau", dict(mesh=2000, num_point=6)),
                                (models.brachistochrone, "lobatto", dict(mesh=mesh, num_point=K)),
                                (models.two_stage_rocket, "radau", dict(mesh=mesh, num_point=K))):
        system, phases, guess = builder(_ns(scheme, "pockit_amd"), **kw)
        ref, rphases, _ = builder(_ns(scheme, "oracle"), **kw)
        ref.prepare()
        x, _, _ = models.bench_inputs(system, guess)
        data = system.evaluator.mesh_error(x)
        s = x[ref.l_s: ref.r_s]
        for k, rp in enumerate(rphases):
            T, I = oref.error_data(rp, x[ref.l_p[k]: ref.r_p[k]].copy(), s)
            close(data[k][0], T, what="T")
            close(data[k][1], I, what="I")
            for atol, rtol in ((1e-3, 1e-3), (1e-9, 1e-9)):
                from pockit_amd import refine as pref

                assert np.array_equal(pref.interval_ok(phases[k].layout, data[k][0], data[k][1], atol, rtol, 1e-4),
                                      oref.check_intervals(rp, T, I, atol, rtol, 1e-4))
        system._invalidate()


def test_cycle_graph_replay_gives_identical_results():
    """pk_set_cycle_graph: the cached hipGraph of the fused cycle reproduces the plain launches bit for bit, and a
    change of sigma (re-capture) is honoured."""
    import ctypes as C

    import torch

    system, _, guess = models.brachistochrone(_ns("radau", "pockit_amd"), mesh=37, num_point=5)
    plan, ev = system.plan, system.evaluator
    x, lam, _ = models.bench_inputs(system, guess)
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))
    torch.cuda.synchronize()

    def run(sigma):
        o = {k: torch.zeros(n, dtype=torch.float64, device=dev) for k, n in sizes}
        torch.cuda.synchronize()
        for _ in range(3):
            ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o[k].data_ptr() for k, _ in sizes])
        ev.sync()
        return {k: v.cpu().numpy() for k, v in o.items()}, o

    plain = {s: run(s)[0] for s in (1.0, 0.25)}
    ev.set_cycle_graph(True)
    try:
        for s in (1.0, 0.25, 1.0):
            got, keep = run(s)
            for k, _ in sizes:
                assert np.array_equal(got[k], plain[s][k]), (k, s)
    finally:
        ev.set_cycle_graph(False)


def test_batches_enqueued_by_the_library_equal_single_cycles():
    """pk_eval_cycle_dev_repeat (what bench.py's timed batches call): `count` cycles enqueued from C, as plain launches and
    -- pk_set_cycle_graph -- as one hipGraph of `count` kernel nodes replayed twice, leave exactly the outputs of a single
    cycle (the hand-off slots are put back by every launch, so any number of launches may follow each other)."""
    import ctypes as C

    import torch

    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), mesh=200, num_point=6)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    want = ev.cycle(x, lam, sigma)
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))
    lib, h = ev.ctx.lib, ev.ctx.handle

    def run(count, times):
        o = {k: torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for k, n in sizes}
        torch.cuda.synchronize()
        for _ in range(times):
            ev.ctx.check(lib.pk_eval_cycle_dev_repeat(h, C.c_void_p(dx.data_ptr()), C.c_void_p(dlam.data_ptr()), C.c_double(float(sigma)),
                                                      *[C.c_void_p(o[k].data_ptr()) for k, _ in sizes], None, count, 0, None))
        ev.sync()
        return [o[k].cpu().numpy() for k, _ in sizes]

    for got in (run(1, 1), run(7, 1), run(20, 3)):
        for a, b in zip(got, want):
            assert np.array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())
    ev.set_cycle_graph(True)
    try:
        for got in (run(20, 2), run(5, 3), run(20, 1)):              # capture, replay, re-capture for another count
            for a, b in zip(got, want):
                assert np.array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())
    finally:
        ev.set_cycle_graph(False)


@pytest.mark.parametrize("case", [("planar_quadrotor", "radau", dict(mesh=301, num_point=6)),
                                  ("two_stage_rocket", "radau", dict(mesh=150, num_point=4)),
                                  ("brachistochrone", "lobatto", dict(mesh=1100, num_point=5)),
                                  ("humanoid_wbc", "radau", dict(mesh=260, num_point=8)),
                                  ("lqr", "lobatto", dict(mesh=10, num_point=10)),       # objective depends on a static parameter
                                  ("lqr", "radau", dict(mesh=7, num_point=12))])         # tables read from global memory (K > 8)
@pytest.mark.parametrize("split", ["1", "0"])
def test_single_launch_cycle_equals_two_launch_cycle_bit_for_bit(case, split, monkeypatch):
    """pk_cycle (x-kernel, Hessian and finalize workgroups in ONE launch, partial sums handed over inside the
    launch) runs the same waves and the same fixed-shape reductions as pk_xall -> pk_hess(+reductions): all five
    outputs must be bit-identical, for split (<= 1024 tiles) and unsplit launches, and equal to the oracle (a model whose
    cycle runs pass-parallel -- the humanoid on this mesh, DESIGN.md section 3c -- agrees to 1e-13 instead)."""
    import torch

    monkeypatch.setenv("POCKIT_AMD_SPLIT", split)      # two waves per tile in the x-part (values / Jacobian) or one
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))

    def run():
        o = {k: torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for k, n in sizes}
        torch.cuda.synchronize()
        for _ in range(2):
            ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o[k].data_ptr() for k, _ in sizes])
        ev.sync()
        return {k: v.cpu().numpy() for k, v in o.items()}

    single = run()
    ev.set_cycle_mode(False)
    try:
        two = run()
    finally:
        ev.set_cycle_mode(True)
    for k, _ in sizes:
        if ev.src.cycle_subs:      # passes as workgroups of their own in pk_cycle / pk_hess, as a loop in pk_xall: the same
            close(single[k], two[k], tol=1e-13, what=k)     # expressions compiled in different surroundings (last bits)
        else:
            assert np.array_equal(single[k], two[k]), k
    close(single["f"][0], ref.objective(x), what="f")
    close(single["grad"], ref.gradient(x), what="grad")
    close(single["g"], ref.constraints(x), what="g")
    close(single["J"], ref.jacobian(x), what="J")
    close(single["H"], ref.hessian(x, lam, sigma), what="H")


def test_single_launch_cycle_on_a_mesh_with_more_workgroups_than_the_chip_holds():
    """360k nodes: ~5000 workgroups, more than are resident at once -- the finalize workgroup of pk_cycle polls
    while tile workgroups are still waiting to be dispatched.  Must equal the two-launch cycle bit for bit and the
    oracle within tolerance."""
    import torch

    kw = dict(mesh=60000, num_point=6)
    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), **kw)
    plan, ev = system.plan, system.evaluator
    assert len(ev.tables.tiles) > 4096
    x, lam, sigma = models.bench_inputs(system, guess)
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))

    def run():
        o = {k: torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for k, n in sizes}
        torch.cuda.synchronize()
        for _ in range(2):
            ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o[k].data_ptr() for k, _ in sizes])
        ev.sync()
        return {k: v.cpu().numpy() for k, v in o.items()}

    single = run()
    ev.set_cycle_mode(False)
    try:
        two = run()
    finally:
        ev.set_cycle_mode(True)
    for k, _ in sizes:
        assert np.array_equal(single[k], two[k]), k
    ref, _, _ = models.planar_quadrotor(_ns("radau", "oracle"), **kw)
    close(single["f"][0], ref.objective(x), what="f")
    close(single["grad"], ref.gradient(x), what="grad")
    close(single["g"], ref.constraints(x), what="g")
    close(single["J"], ref.jacobian(x), what="J")
    system._invalidate()


def test_single_launch_cycle_hand_off_survives_back_to_back_launches():
    """The hand-off slots of pk_cycle are emptied by the launch that consumed them: 600 cycles enqueued back to
    back on alternating iterates, every launch writing f and grad f to its own slot, must each reproduce the value
    of their iterate exactly (a stale or missed partial sum would show up in f or in the t0/tf gradient slots)."""
    import torch

    system, _, guess = models.two_stage_rocket(_ns("radau", "pockit_amd"), 120, 4)
    plan, ev = system.plan, system.evaluator
    xa, lam, sigma = models.bench_inputs(system, guess)
    xb = xa * (1.0 + 0.05 * np.random.default_rng(11).uniform(-1, 1, xa.shape))
    dev = torch.device("cuda", 0)
    dxs = [torch.from_numpy(v).to(dev) for v in (xa, xb)]
    dlam = torch.from_numpy(lam).to(dev)
    reps = 600
    f = torch.full((reps,), float("nan"), dtype=torch.float64, device=dev)
    grad = torch.full((reps, plan.n), float("nan"), dtype=torch.float64, device=dev)
    g, J, H = (torch.zeros(n, dtype=torch.float64, device=dev) for n in (plan.m, plan.nnz_J, plan.nnz_H))
    torch.cuda.synchronize()
    for i in range(reps):
        ev.cycle_dev(dxs[i % 2].data_ptr(), dlam.data_ptr(), sigma, f[i:].data_ptr(), grad[i].data_ptr(), g.data_ptr(),
                     J.data_ptr(), H.data_ptr())
    ev.sync()
    f, grad = f.cpu().numpy(), grad.cpu().numpy()
    for par in (0, 1):
        assert np.all(f[par::2] == f[par]) and np.all(grad[par::2] == grad[par]), par
    assert f[0] != f[1]
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 120, 4)
    for par, x in ((0, xa), (1, xb)):
        close(f[par], ref.objective(x), what="f")
        close(grad[par], ref.gradient(x), what="grad")


@pytest.mark.parametrize("name", ["brach_lgr_3x4", "quad_lgl_4x5", "rocket_lgr_3x4", "humanoid_lgr_2x3", "worked_lgr"])
def test_device_csr_handoff_matches_reference_matrices(name):
    """pk_csr: J and the lower triangle of H gathered into CSR on the device equal the matrices the reference's
    triplets assemble to (scipy COO -> CSR of the golden vectors)."""
    import scipy.sparse
    import torch

    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, _ = builder(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = gold["x"], gold["lam"], float(gold["sigma"])
    n, m = int(gold["n"]), int(gold["m"])
    J = system.jacobian_csr(x)
    H = system.hessian_csr(x, lam, sigma)
    Jref = scipy.sparse.coo_array((gold["J"], (gold["jr"], gold["jc"])), shape=(m, n)).tocsr()
    Href = scipy.sparse.coo_array((gold["H"], (gold["hr"], gold["hc"])), shape=(n, n)).tocsr()
    for got, ref in ((J, Jref), (H, Href)):
        ref.sum_duplicates()
        ref.sort_indices()
        assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
        close(got.data, ref.data, what="csr values")
    # device-pointer route: the fused cycle's triplets gathered without leaving the GPU
    ev, plan = system.evaluator, system.plan
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    o = {k: torch.zeros(max(c, 1), dtype=torch.float64, device=dev) for k, c in
         (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))}
    cj = torch.zeros(ev.csr_map("jac").nnz, dtype=torch.float64, device=dev)
    ch = torch.zeros(ev.csr_map("hess").nnz, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o[k].data_ptr() for k in ("f", "grad", "g", "J", "H")])
    ev.gather_csr_dev("jac", o["J"].data_ptr(), cj.data_ptr())
    ev.gather_csr_dev("hess", o["H"].data_ptr(), ch.data_ptr())
    ev.sync()
    close(cj.cpu().numpy(), Jref.data, what="J csr dev")
    close(ch.cpu().numpy(), Href.data, what="H csr dev")


def test_device_csr_full_size_matches_host_gather():
    """Quadrotor 2000 x 6: device gather == host gather of the device triplets (bit-exact for J: no repeats)."""
    import torch

    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), mesh=2000, num_point=6)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    J, H = ev.jacobian_direct(x), ev.hessian_direct(x, lam, sigma)   # the stand-alone kernels the CSR calls run
    mj, mh = ev.csr_map("jac"), ev.csr_map("hess")
    assert mj.seg is None and mh.seg is not None
    assert np.array_equal(ev.jacobian_csr(x), mj.gather(J))
    assert "hessc" in ev._csr, "the compact Hessian's pattern should be the full pattern's set of entries"
    close(ev.hessian_csr(x, lam, sigma), mh.gather(H), what="H csr (from the compact evaluation)")
    dev = torch.device("cuda", 0)                      # the triplet route stays available for device-resident triplets
    dH, out = torch.from_numpy(H).to(dev), torch.zeros(mh.nnz, dtype=torch.float64, device=dev)
    ev.gather_csr_dev("hess", dH.data_ptr(), out.data_ptr())
    ev.sync()
    close(out.cpu().numpy(), mh.gather(H), 1e-14, what="H csr (gathered triplets)")


# ---------------------------------------------------------------------------------------------------------
# the reference's own check tests (tests/test_radau/test_check_radau.py, tests/test_labatto/test_check_lobatto.py)
# restated against the GPU-backed API
def _check_model(ns, num_point):
    s = ns.System(1)
    p = s.new_phase(1, 1)
    p.set_dynamics([p.u[0]])
    p.set_boundary_condition([None], [None], None, None)
    p.set_phase_constraint([p.u[0] + p.s[0]], [0.0], [2.0], [True])
    p.set_discretization([0, 0.1, 1], num_point)
    s.set_phase([p])
    s.set_objective(s.s[0])
    return s, p


def test_reference_check_discontinuous_radau():
    ns = _ns("radau", "pockit_amd")
    s, p = _check_model(ns, [2, 3])
    v = ns.constant_guess(p, 0.0)
    assert isinstance(s.check_discontinuous([v, [2.0]]), bool)
    assert s.check_discontinuous([v, [2.0]])
    assert s.check_discontinuous([v, [2.01]])
    assert not s.check_discontinuous([v, [1.99]])
    v.u[0] = np.array([-1, -1, 1, 1, 1], dtype=np.float64)
    assert s.check_discontinuous([v, [1.0]])
    assert not s.check_discontinuous([v, [1.01]])
    v.u[0] = np.array([0, 0.01, 2, 2, 2], dtype=np.float64)
    assert not s.check_discontinuous([v, [0.0]])
    with pytest.raises(ValueError):
        s.check_discontinuous(v)


@pytest.mark.parametrize("scheme,num_point", [("radau", [2, 3]), ("lobatto", [3, 4])])
def test_reference_check_continuous(scheme, num_point):
    ns = _ns(scheme, "pockit_amd")
    s, p = _check_model(ns, num_point)
    v = ns.constant_guess(p, 1.0)
    v.x[0] = v.t_x
    assert isinstance(s.check_continuous([v, [0.0]]), bool)
    assert s.check_continuous([v, [0.0]])
    v.u[0] = v.t_u * 2
    v.x[0] = v.t_x**2
    assert s.check_continuous([v, [0.0]])
    v.u[0][0] += 0.01
    assert not s.check_continuous([v, [0.0]])
    v.u[0] = v.t_u * 1.99
    assert not s.check_continuous([v, [0.0]])


def test_reference_check_discontinuous_lobatto_is_not_implemented():
    ns = _ns("lobatto", "pockit_amd")
    s, p = _check_model(ns, [3, 4])
    v = ns.constant_guess(p, 0.0)
    with pytest.raises(NotImplementedError):
        s.check_discontinuous([v, [2.0]])
    v.x[0] = v.t_x
    v.u[0] = v.t_u * 0 + 1.0
    assert s.check([v, [0.0]]) == s.check_continuous([v, [0.0]])   # lobatto: check == check_continuous


# ---------------------------------------------------------------------------------------------------------
# the three tests of the reference's tests/test_base/test_system_base.py:10-70 that were not restated yet
@pytest.mark.parametrize("scheme", ["radau", "lobatto"])
def test_reference_static_only_system(scheme):
    """A system without phases (test_system_base.py:10-20): f = s^2, grad = [2 s], no constraints -- evaluated by the
    system-level workgroups alone (no tile workgroup exists)."""
    system = _ns(scheme, "pockit_amd").System(1)
    system.set_objective(system.s[0] ** 2)
    x = np.array([2.0], dtype=np.float64)
    assert system.objective(x) == pytest.approx(4.0)
    assert np.allclose(system.gradient(x), [4.0])
    assert system.constraints(x).shape == (0,)
    assert system.jacobian(x).shape == (0,)
    ref = _ns(scheme, "oracle").System(1)
    ref.set_objective(ref.s[0] ** 2)
    lam = np.zeros(0)
    hr, hc = system.hessianstructure()
    rr, rc = ref.hessianstructure()
    assert np.array_equal(hr, rr) and np.array_equal(hc, rc)
    close(system.hessian(x, lam, 0.5), ref.hessian(x, lam, 0.5), what="H")
    f, grad, g, J, H = system.evaluator.cycle(x, lam, 0.5)         # the one-launch cycle: three workgroups, no tiles
    assert f == pytest.approx(4.0) and np.allclose(grad, [4.0]) and g.shape == (0,) and J.shape == (0,)
    close(H, ref.hessian(x, lam, 0.5), what="cycle H")
    x3 = np.array([-3.0])
    assert system.objective(x3) == pytest.approx(9.0) and np.allclose(system.gradient(x3), [-6.0])


def test_reference_static_only_system_with_system_constraints():
    """Static parameters only, with an objective and system constraints in them: f, grad f, g, J, H against the oracle."""
    def build(ns):
        system = ns.System(3)
        a, b, c = system.s
        system.set_objective(a ** 2 * b + sp_sin(c) * a)
        system.set_system_constraint([a * b * c, a + b ** 2], [0.0, -1.0], [1.0, 1.0])
        return system

    import sympy

    sp_sin = sympy.sin
    system, ref = build(_ns("radau", "pockit_amd")), build(_ns("radau", "oracle"))
    x = np.array([0.7, -1.3, 0.4])
    lam = np.array([0.3, -2.0])
    assert np.array_equal(system.jacobianstructure()[0], ref.jacobianstructure()[0])
    assert np.array_equal(system.jacobianstructure()[1], ref.jacobianstructure()[1])
    assert np.array_equal(system.hessianstructure()[0], ref.hessianstructure()[0])
    assert np.array_equal(system.hessianstructure()[1], ref.hessianstructure()[1])
    close(system.objective(x), ref.objective(x), what="f")
    close(system.gradient(x), ref.gradient(x), what="grad")
    close(system.constraints(x), ref.constraints(x), what="g")
    close(system.jacobian(x), ref.jacobian(x), what="J")
    close(system.hessian(x, lam, 1.7), ref.hessian(x, lam, 1.7), what="H")


def test_reference_phase_check_uses_discontinuous_tolerance():
    """test_system_base.py:22-32: ``phase.check`` passes its discontinuous tolerance on (u = 0.9995 of a bang-bang
    control in [0, 1] is settled at tolerance 1e-3)."""
    ns = _ns("radau", "pockit_amd")
    system = ns.System(0)
    phase = system.new_phase(1, 1)
    phase.set_dynamics([0]).set_boundary_condition([0], [0], 0, 1)
    phase.set_phase_constraint([phase.u[0]], [0], [1], bang_bang_control=True).set_discretization(1, 3)
    system.set_phase([phase]).set_objective(0)
    variable = ns.constant_guess(phase, 0)
    variable.u[0] = 0.9995
    assert phase.check(variable, tolerance_discontinuous=1.0e-3)
    # (the imported reference answers True at 1e-4 and 6e-4 as well for this input: a constant control has no jump)
    assert phase.check(variable, tolerance_discontinuous=1.0e-4)


def test_reference_reconfiguring_boundary_condition_clears_old_derivatives():
    """test_system_base.py:34-70: a FUNC boundary value replaced by a FREE and then by a FIXED one must leave no trace
    of the static-parameter dependence -- structure equal to the oracle's after the same sequence, J against central
    finite differences of the constraints callback (the reference's own check)."""
    def build(ns):
        system = ns.System(1)
        phase = system.new_phase(1, 0)
        phase.set_dynamics([0]).set_boundary_condition([system.s[0] ** 2], [None], 0, 1).set_discretization(1, 3)
        phase.set_boundary_condition([None], [None], 0, 1)
        phase.set_boundary_condition([0], [None], 0, 1)
        system.set_phase([phase]).set_objective(0)
        return system, phase

    ns = _ns("radau", "pockit_amd")
    system, phase = build(ns)
    ref, _ = build(_ns("radau", "oracle"))
    x = np.concatenate([ns.constant_guess(phase, 0).data, [2.0]])
    row, col = system.jacobianstructure()
    rrow, rcol = ref.jacobianstructure()
    assert np.array_equal(row, rrow) and np.array_equal(col, rcol)
    assert not np.any(col == system.l_s), "no Jacobian entry may depend on the static parameter any more"
    jacobian = np.zeros((len(system.c_lb), system.L), dtype=np.float64)
    np.add.at(jacobian, (row, col), system.jacobian(x.copy()))
    eps = 1.0e-6
    finite_difference = np.empty_like(jacobian)
    for i in range(system.L):
        delta = np.zeros(system.L, dtype=np.float64)
        delta[i] = eps
        finite_difference[:, i] = (system.constraints(x.copy() + delta) - system.constraints(x.copy() - delta)) / (2 * eps)
    assert np.allclose(jacobian, finite_difference)
    close(system.jacobian(x), ref.jacobian(x), what="J")
    # the intermediate configurations evaluate correctly too (FUNC, then FREE)
    for bc in ([None], "func"):
        s2 = ns.System(1)
        p2 = s2.new_phase(1, 0)
        r2 = _ns("radau", "oracle").System(1)
        q2 = r2.new_phase(1, 0)
        for sy, ph in ((s2, p2), (r2, q2)):
            ph.set_dynamics([0]).set_boundary_condition([0], [None], 0, 1).set_discretization(1, 3)
            ph.set_boundary_condition([sy.s[0] ** 2] if bc == "func" else bc, [None], 0, 1)
            sy.set_phase([ph]).set_objective(0)
        close(s2.jacobian(x), r2.jacobian(x), what="J " + str(bc))
        assert np.array_equal(s2.jacobianstructure()[1], r2.jacobianstructure()[1])


def test_check_and_refine_loop_like_the_hyper_sensitive_example():
    """system.check / system.refine in the adaptive loop of examples/hyper_sensitive.py:94-116 (no solver: the
    'solution' is a smooth guess, so each sweep must strictly reduce the number of failing intervals)."""
    ns = _ns("radau", "pockit_amd")
    system, phases, guess = models.brachistochrone(ns, mesh=4, num_point=3)
    value = guess[0] if isinstance(guess, list) and len(guess) == 1 else guess
    tol = 1e-6
    assert not system.check(value, absolute_tolerance_continuous=tol, relative_tolerance_continuous=tol)
    sizes = [phases[0].L]
    for _ in range(3):
        value = system.refine(value, absolute_tolerance_continuous=tol, relative_tolerance_continuous=tol,
                              num_point_min=4, num_point_max=8, mesh_length_min=1e-6)
        sizes.append(phases[0].L)
        assert len(value.data) == phases[0].L == system.L
        assert np.isfinite(system.objective(value.data))
    assert sizes[1] > sizes[0]


# ---------------------------------------------------------------------------------------------------------
# the reference's finite-difference derivative tests (tests/test_radau/test_derivative_radau.py:43-144,
# tests/test_labatto/test_derivative_lobatto.py) restated against the GPU evaluator: same model, same point
# x = arange(L)/10 + 1, same steps and tolerances
def _dense(rows, cols, vals, shape):
    M = np.zeros(shape)
    np.add.at(M, (np.asarray(rows), np.asarray(cols)), vals)
    return M


@pytest.mark.parametrize("scheme", ["radau", "lobatto"])
def test_reference_finite_difference_derivative_checks(scheme):
    s, _, _ = models.derivative_model(_ns(scheme, "pockit_amd"))
    n = s.L
    x = np.arange(n, dtype=np.float64) / 10 + 1
    m = len(s.constraints(x))

    def shifted(*moves):
        y = x.copy()
        for i, d in moves:
            y[i] += d
        return y

    # gradient and Jacobian: central differences, eps = 1e-6, np.allclose defaults
    eps = 1e-6
    fd_grad = np.array([(s.objective(shifted((i, eps))) - s.objective(shifted((i, -eps)))) / (2 * eps) for i in range(n)])
    assert np.allclose(s.gradient(x), fd_grad)
    fd_jac = np.stack([(s.constraints(shifted((i, eps))) - s.constraints(shifted((i, -eps)))) / (2 * eps)
                       for i in range(n)], axis=1)
    jr, jc = s.jacobianstructure()
    assert np.allclose(_dense(jr, jc, s.jacobian(x), (m, n)), fd_jac)

    # Hessians: four-point formula, eps = 2e-3, atol = rtol = 1e-4 on the lower triangle
    eps = 2e-3
    fd_o = np.zeros((n, n))
    fd_c = np.zeros((m, n, n))
    for i in range(n):
        for j in range(i + 1):
            pts = [shifted((i, eps), (j, eps)), shifted((i, eps), (j, -eps)), shifted((i, -eps), (j, eps)),
                   shifted((i, -eps), (j, -eps))]
            f = [s.objective(p) for p in pts]
            g = [s.constraints(p) for p in pts]
            fd_o[i, j] = (f[0] - f[1] - f[2] + f[3]) / eps / eps / 4
            fd_c[:, i, j] = (g[0] - g[1] - g[2] + g[3]) / eps / eps / 4
    hr, hc = s.hessianstructure_o()
    assert np.allclose(_dense(hr, hc, s.hessian_o(x), (n, n)), fd_o, atol=1e-4, rtol=1e-4)
    hr, hc = s.hessianstructure()
    for c in range(m):
        unit = np.zeros(m)
        unit[c] = 1.0
        assert np.allclose(_dense(hr, hc, s.hessian(x, unit, 0.0), (n, n)), fd_c[c], atol=1e-4, rtol=1e-4), c


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_hp_meshes_all_callbacks_match_oracle(seed):
    """Randomised hp meshes (every interval its own width and K, LGR K = 1..12 / LGL K = 2..12; K > 8 takes the
    kernels' global-table path, K <= 8 the LDS-staged one) over all example models, all five callbacks, the
    stand-alone kernels and the fused cycle."""
    rng = np.random.default_rng(100 + seed)
    builder = [models.brachistochrone, models.planar_quadrotor, models.two_stage_rocket, models.humanoid_wbc,
               models.derivative_model, models.brachistochrone][seed]
    scheme = ["radau", "lobatto"][seed % 2]
    n_int = int(rng.integers(5, 40))
    lo = 1 if scheme == "radau" else 2
    K = rng.integers(lo, 13, size=n_int).tolist()
    mesh = np.concatenate(([0.0], np.cumsum(rng.uniform(0.05, 1.0, size=n_int)))).tolist()
    kw = dict(mesh=mesh, num_point=K)
    system, _, guess = builder(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = builder(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    sigma = 0.3 + 0.1 * seed
    assert np.array_equal(system.jacobianstructure()[0], ref.jacobianstructure()[0])
    assert np.array_equal(system.jacobianstructure()[1], ref.jacobianstructure()[1])
    assert np.array_equal(system.hessianstructure()[0], ref.hessianstructure()[0])
    assert np.array_equal(system.hessianstructure()[1], ref.hessianstructure()[1])
    want = dict(f=ref.objective(x), grad=ref.gradient(x), g=ref.constraints(x), J=ref.jacobian(x),
                H=ref.hessian(x, lam, sigma))
    ev = system.evaluator
    close(ev.objective_direct(x), want["f"], what="f")
    close(ev.gradient_direct(x), want["grad"], what="grad")
    close(ev.constraints_direct(x), want["g"], what="g")
    close(ev.jacobian_direct(x), want["J"], what="J")
    close(ev.hessian_direct(x, lam, sigma), want["H"], what="H")
    f, grad, g, J, H = ev.cycle(x, lam, sigma)
    for got, key in ((f, "f"), (grad, "grad"), (g, "g"), (J, "J"), (H, "H")):
        close(got, want[key], what="cycle " + key)


def test_adaptive_solve_check_refine_loop_converges():
    """The whole adaptive workflow on the GPU evaluator, as examples/hyper_sensitive.py:88-120 runs it with IPOPT:
    solve (SciPy trust-constr here), system.check, system.refine, re-solve on the refined mesh from the adapted
    solution -- until the mesh error check passes.  The objective must approach the Riccati optimum."""
    from scipy.integrate import solve_ivp

    from pockit_amd.optimizer import scipy as scipy_solver

    ns = _ns("radau", "pockit_amd")
    system, (phase,), _ = models.lqr(ns, 2, 3)
    value = [ns.constant_guess(phase, 0.0), [0.0]]
    a, b, q, r, sw = -1.0, 1.0, 1.0, 0.1, 1.0
    ric = solve_ivp(lambda t, P: -(2 * a * P - b * b * P * P / r + q), (1.0, 0.0), [sw / 2], rtol=1e-11, atol=1e-12)
    optimum = ric.y[0, -1]
    tol = 1e-6
    errors, sizes = [], []
    for sweep in range(6):
        value, res = scipy_solver.solve(system, value, {"maxiter": 300, "gtol": 1e-10, "xtol": 1e-12})
        errors.append(abs(res.fun - optimum))
        sizes.append(system.L)
        if system.check(value, absolute_tolerance_continuous=tol, relative_tolerance_continuous=tol):
            break
        value = system.refine(value, absolute_tolerance_continuous=tol, relative_tolerance_continuous=tol,
                              num_point_min=3, num_point_max=8, mesh_length_min=1e-4)
        assert len(value[0].data) == phase.L
    else:
        pytest.fail(f"mesh error check never passed: errors {errors}, sizes {sizes}")
    assert len(sizes) >= 2 and sizes[-1] > sizes[0], sizes            # the coarse mesh had to be refined
    assert errors[-1] <= 1e-6 * max(1.0, abs(optimum)) and errors[-1] < errors[0], errors


@pytest.mark.parametrize("layout", ["compact", "reference", "auto small", "auto large", "reference from compact",
                                    "auto small from compact"])
def test_ipopt_adapter_protocol_with_a_stand_in_cyipopt(layout, monkeypatch):
    """cyipopt / Ipopt are not installed here, so the adapter is driven by a stand-in ``cyipopt.Problem`` that
    does what cyipopt does with a ``problem_obj`` (cyipopt's Problem.__init__/solve contract: structure queried
    once, every callback result copied into the solver's own arrays at once, callbacks in IPOPT's per-iteration
    order) and checks every value it receives against the oracle.  This covers the adapter's wiring and the
    evaluator's zero-copy mode (results handed out as views of pinned buffers that the next call reuses).
    ``layout="compact"``: the solver is handed the compact structures; what it assembles from them -- repeated positions
    summed, as IPOPT does -- must equal the oracle's scatter-added reference triplets.  ``layout="reference"``: structures
    and values are the reference's, entry by entry.  The default, ``"auto"``: the compact layouts from AUTO_COMPACT_BYTES of
    reference J + H values per iterate on (here: with the threshold lowered), the reference's below."""
    import sys
    import types

    import scipy.sparse as ssp

    ns = _ns("radau", "pockit_amd")
    system, phases, guess = models.two_stage_rocket(ns, 12, 4)
    ref, _, _ = models.two_stage_rocket(_ns("radau", "oracle"), 12, 4)
    seen = {"iters": 0, "options": {}}
    # ("... from compact": the user had set the system to the compact layouts before -- "reference", and "auto" below its
    #  threshold, must still hand the solver the reference's lists, and the user's setting must be back afterwards: ADVICE r4)
    from_compact = layout.endswith(" from compact")
    layout = layout.replace(" from compact", "")
    if from_compact:
        system.set_hessian_layout("compact")
        system.set_jacobian_layout("compact")
    want_layout = {"auto small": "reference", "auto large": "compact"}.get(layout, layout)
    layout, asked = want_layout, layout
    rjr, rjc = ref.jacobianstructure()
    rhr, rhc = ref.hessianstructure()

    def assembled(vals, rows, cols, shape):
        return ssp.coo_array((np.asarray(vals, dtype=np.float64), (rows, cols)), shape=shape).tocsr()

    def same_matrix(a, b, what):
        diff = abs(a - b)
        assert (diff.max() if diff.nnz else 0.0) <= TOL * max(1.0, abs(b).max()), what

    class Problem:
        def __init__(self, n, m, problem_obj, lb, ub, cl, cu):
            assert n == ref.L and m == len(ref.c_lb) and len(lb) == len(ub) == n and len(cl) == len(cu) == m
            assert np.array_equal(lb, ref.v_lb) and np.array_equal(ub, ref.v_ub)
            assert np.array_equal(cl, ref.c_lb) and np.array_equal(cu, ref.c_ub)
            self.n, self.m, self.obj = n, m, problem_obj
            self.jr, self.jc = problem_obj.jacobianstructure()
            self.hr, self.hc = problem_obj.hessianstructure()
            if layout == "reference":
                assert np.array_equal(self.jr, rjr) and np.array_equal(self.jc, rjc)
                assert np.array_equal(self.hr, rhr) and np.array_equal(self.hc, rhc)
            else:       # fewer values than the reference's lists, lower triangle kept
                assert len(self.hr) < len(rhr) and len(self.jr) <= len(rjr) and np.all(np.asarray(self.hr) >= np.asarray(self.hc))
            self.nnz_j, self.nnz_h = len(self.jr), len(self.hr)

        def add_option(self, key, value):
            seen["options"][key] = value

        def solve(self, x0):
            rng = np.random.default_rng(3)
            x = np.array(x0, dtype=np.float64)
            for it in range(4):
                lam = rng.standard_normal(self.m)
                sigma = 1.0 if it else 0.5
                got = {}
                got["f"] = float(self.obj.objective(x))                       # line search: f and g
                got["g"] = np.array(self.obj.constraints(x), dtype=np.float64)
                got["grad"] = np.array(self.obj.gradient(x), dtype=np.float64)  # accepted point: derivatives
                got["J"] = np.array(self.obj.jacobian(x), dtype=np.float64)
                got["H"] = np.array(self.obj.hessian(x, lam, sigma), dtype=np.float64)
                assert got["J"].shape == (self.nnz_j,) and got["H"].shape == (self.nnz_h,)
                close(got["f"], ref.objective(x), what="f")
                close(got["g"], ref.constraints(x), what="g")
                close(got["grad"], ref.gradient(x), what="grad")
                if layout == "reference":
                    close(got["J"], ref.jacobian(x), what="J")
                    close(got["H"], ref.hessian(x, lam, sigma), what="H")
                same_matrix(assembled(got["J"], self.jr, self.jc, (self.m, self.n)),
                            assembled(ref.jacobian(x), rjr, rjc, (self.m, self.n)), "J as the solver assembles it")
                same_matrix(assembled(got["H"], self.hr, self.hc, (self.n, self.n)),
                            assembled(ref.hessian(x, lam, sigma), rhr, rhc, (self.n, self.n)), "H as the solver assembles it")
                x = x * (1.0 + 1e-3 * rng.uniform(-1, 1, size=len(x)))
                seen["iters"] += 1
            return x, {"status": 0, "status_msg": b"stand-in", "obj_val": got["f"]}

    fake = types.ModuleType("cyipopt")
    fake.Problem = Problem
    monkeypatch.setitem(sys.modules, "cyipopt", fake)
    from pockit_amd.optimizer import ipopt

    if asked.startswith("auto"):
        if asked == "auto large":
            monkeypatch.setattr(ipopt, "AUTO_COMPACT_BYTES", 1)
        else:
            assert 8 * (system.plan.nnz_J + system.plan.nnz_H) < ipopt.AUTO_COMPACT_BYTES
        solution, info = ipopt.solve(system, guess, {"tol": 1e-8, "print_level": 0})          # (the default: "auto")
    else:
        solution, info = ipopt.solve(system, guess, {"tol": 1e-8, "print_level": 0}, layout=layout)
    assert seen["iters"] == 4 and seen["options"] == {"tol": 1e-8, "print_level": 0} and info["status"] == 0
    assert len(solution) == len(phases) + 1 and all(len(v.data) == p.L for v, p in zip(solution, phases))
    assert len(solution[-1]) == system.n_s and system.evaluator.zero_copy is False
    # the system's own layout settings are back (the reference's lists: the drop-in default of the callbacks themselves)
    if from_compact:
        assert system._hessian_layout == system._jacobian_layout == "compact" and len(system.hessianstructure()[0]) < len(rhr)
        system.set_hessian_layout("reference")
        system.set_jacobian_layout("reference")
    assert np.array_equal(system.hessianstructure()[0], rhr) and np.array_equal(system.jacobianstructure()[0], rjr)


@pytest.mark.parametrize("name", sorted(models.BANG_BANG_CASES))
def test_bang_bang_refinement_end_to_end_matches_reference(name):
    """system.check_discontinuous / refine_discontinuous on the GPU-backed system against the reference's results
    (tests/golden/bangbang): the scaled constraint values come from pk_g on the device."""
    kw, _ = models.BANG_BANG_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "bangbang", name + ".npz"))
    ns = _ns("radau", "pockit_amd")
    for tag, (dtol, kmin, kmax, lmin, lmax) in (("a", (1e-3, 4, 8, 1e-3, 1.0)), ("b", (5e-2, 3, 6, 2e-2, 0.3))):
        system, (p,), _ = models.bang_bang_model(ns, **kw)
        value = [ns.Variable(p, gold["data"].copy()), gold["s"].copy()]
        assert system.check_discontinuous(value, dtol, 1e-4) == bool(np.all(gold[f"ok_{tag}"]))
        out = system.refine_discontinuous(value, dtol, num_point_min=kmin, num_point_max=kmax, mesh_length_min=lmin,
                                          mesh_length_max=lmax)
        assert len(p._mesh) == len(gold[f"mesh_{tag}"]) and np.allclose(p._mesh, gold[f"mesh_{tag}"], rtol=0, atol=1e-9)
        assert np.array_equal(p._num_point, gold[f"K_{tag}"])
        close(out[0].data, gold[f"adapt_{tag}"], 1e-9, what="adapted values")
        x_new = np.concatenate([out[0].data, out[1]])
        assert len(x_new) == system.L and np.isfinite(system.objective(x_new))
        # system.refine takes the bang-bang branch first (the check fails), as the reference does
        system2, (p2,), _ = models.bang_bang_model(ns, **kw)
        system2.refine([ns.Variable(p2, gold["data"].copy()), gold["s"].copy()], 1e-8, 1e-8, dtol, kmin, kmax, lmin, lmax)
        assert len(p2._mesh) == len(gold[f"mesh_{tag}"]) and np.allclose(p2._mesh, gold[f"mesh_{tag}"], rtol=0, atol=1e-9)


def test_landing_blocks_keep_the_constant_jacobian_entries_across_recycling():
    """J, grad f and g of an iterate are views of one pinned block [J | grad f | g] that arrives in one copy; the
    x-independent entries of J (translation part, phasebase.py:1071-1081) are filled into a block once and left out of the
    copy afterwards.  Over more iterates than the ring has blocks -- the caller keeping some arrays, dropping others --
    every returned array must equal the oracle's, constant entries included, and arrays the caller still holds must not
    change; a C-ABI caller's arbitrary target array (pk_set_result_targets) still receives the whole Jacobian."""
    import ctypes as C

    from pockit_amd import runtime

    system, _, guess = models.planar_quadrotor(_ns("radau", "pockit_amd"), mesh=60, num_point=6)
    ref, _, _ = models.planar_quadrotor(_ns("radau", "oracle"), mesh=60, num_point=6)
    x0, lam, sigma = models.bench_inputs(system, guess)
    ev, plan = system.evaluator, system.plan
    runs = ev.jac_constant_runs
    assert runs and runs[0][0] == 0 and runs[0][1] >= 1024, runs
    rng = np.random.default_rng(5)
    kept = []
    for k in range(14):                      # (the ring holds 6 blocks)
        x = x0 * (1.0 + 1e-3 * rng.uniform(-1, 1, x0.shape))
        order = rng.permutation(4)
        got = {}
        for what in order:                   # any order of the x-callbacks
            got[what] = (system.objective, system.gradient, system.constraints, system.jacobian)[what](x)
        H = system.hessian(x, lam, sigma)
        want = (ref.objective(x), ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma))
        for what in range(4):
            close(got[what], want[what], what=f"iterate {k} result {what}")
        close(H, want[4], what=f"iterate {k} H")
        assert np.all(np.abs(got[3][: runs[0][1]]) == 1.0)          # the head of J is the +-1 translation part
        if k % 3 == 0:
            kept.append((got[3], want[3].copy(), got[1], want[1].copy(), H, want[4].copy()))
    for J, Jw, g_, gw, H, Hw in kept:        # arrays the caller kept were never recycled under it
        close(J, Jw, what="kept J"); close(g_, gw, what="kept grad"); close(H, Hw, what="kept H")
    # a plain array as the Jacobian target of a C-ABI caller: all of J is copied
    lib, h, dp = ev.ctx.lib, ev.ctx.handle, runtime.as_dp
    out = [np.full(k, np.nan) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    ev.ctx.check(lib.pk_set_result_targets(h, *[dp(a) for a in out]))
    x = x0 * 1.001
    ev.ctx.check(lib.pk_prepare_x(h, dp(x)))
    for what in range(4):
        ev.ctx.check(lib.pk_fetch(h, what, None))
    ev.ctx.check(lib.pk_set_result_targets(h, None, None, None, None, None))
    ev._invalidate_x()
    close(out[3], ref.jacobian(x), what="J in a caller's plain array")
    close(out[1], ref.gradient(x), what="grad in a caller's plain array")
    # every switch of the shim gives the same arrays
    base = [system.gradient(x).copy(), system.jacobian(x).copy(), system.hessian(x, lam, sigma).copy()]
    for name, dflt in ((b"spin_wait", 1), (b"lambda_direct", 1), (b"chunk_upload", 1), (b"kernel_upload", 1),
                       (b"kernel_download", 8), (b"split_copy", 1), (b"speculative_hess", 1)):
        ev.ctx.check(lib.pk_set_host_option(h, name, 0 if dflt else 1))
        now = [system.gradient(x), system.jacobian(x), system.hessian(x, lam, sigma)]
        x_other = x * 1.0005                   # ... also when the Hessian callback is the first to see a new x
        h_new = system.hessian(x_other, lam, sigma)
        g_new = system.gradient(x_other)
        ev.ctx.check(lib.pk_set_host_option(h, name, dflt))
        for a, b in zip(base, now):
            assert np.array_equal(a, b), name
        close(h_new, ref.hessian(x_other, lam, sigma), what=f"H on a new x, {name}")
        close(g_new, ref.gradient(x_other), what=f"grad on that x, {name}")
    assert lib.pk_set_host_option(h, b"no_such_option", 1) != 0


COMPACT_CASES = [("brachistochrone", "radau", dict(mesh=37, num_point=5)),
                 ("brachistochrone", "lobatto", dict(mesh=23, num_point=6)),
                 ("brachistochrone", "radau", dict(mesh=[0, 0.1, 0.15, 0.5, 0.9, 1.0], num_point=[3, 7, 2, 5, 1])),
                 ("two_stage_rocket", "radau", dict(mesh=40, num_point=3)),
                 ("two_stage_rocket", "lobatto", dict(mesh=11, num_point=4)),
                 ("planar_quadrotor", "lobatto", dict(mesh=19, num_point=4)),
                 ("planar_quadrotor", "radau", dict(mesh=300, num_point=6)),
                 ("humanoid_wbc", "radau", dict(mesh=9, num_point=7)),
                 ("humanoid_wbc", "lobatto", dict(mesh=6, num_point=5)),
                 ("brachistochrone", "radau", dict(mesh=[0, 0.3, 1.0], num_point=[12, 20])),      # tables beyond one lane each
                 ("derivative_model", "radau", {}), ("derivative_model", "lobatto", {}),          # system constraints on integrals
                 ("lqr", "lobatto", dict(mesh=6, num_point=6))]


@pytest.mark.parametrize("case", COMPACT_CASES)
def test_compact_jacobian_equals_coalesced_oracle(case):
    """pk_jacc: dense-column entries of the dynamics contracted with the integration block (one value per defect row); scatter-
    added the compact triplets must equal the scatter-add of the oracle's (= the reference's) triplet list, on exactly the
    reference's set of positions; the reference layout is still served afterwards."""
    import scipy.sparse as ssp

    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    shape = (system.plan.m, system.plan.n)
    jr, jc = ref.jacobianstructure()
    want = ssp.coo_array((ref.jacobian(x), (jr, jc)), shape=shape).tocsr()
    system.set_jacobian_layout("compact")
    cr, cc = system.jacobianstructure()
    vals = system.jacobian(x)
    assert len(vals) == len(cr) <= len(jr)
    assert set(zip(cr.tolist(), cc.tolist())) == set(zip(np.asarray(jr).tolist(), np.asarray(jc).tolist()))
    got = ssp.coo_array((vals, (cr, cc)), shape=shape).tocsr()
    diff = abs(got - want)
    scale = max(1.0, abs(want).max())
    assert (diff.max() if diff.nnz else 0.0) <= (TOL if max(np.atleast_1d(kw.get("num_point", 4))) <= 12 else 1e-8) * scale
    # the compact layout rides on the prepared-x protocol like the reference's: the other callbacks on the same iterate, a
    # second iterate, arrays that stay valid; and the one-shot entry point gives the same values
    tol = TOL if max(np.atleast_1d(kw.get("num_point", 4))) <= 12 else 1e-8
    close(system.gradient(x), ref.gradient(x), what="grad beside the compact J", tol=tol)
    close(system.constraints(x), ref.constraints(x), what="g beside the compact J", tol=tol)
    x2 = x * (1 + 1e-4)
    v2 = system.jacobian(x2)
    assert np.array_equal(system.evaluator.jacobian_compact(x2), v2) and not np.shares_memory(v2, vals)
    assert np.array_equal(system.jacobian(x), vals)
    close(system.hessian(x2, lam, sigma), ref.hessian(x2, lam, sigma), what="H beside the compact J", tol=tol)
    # CSR values of J gathered on the device from the compact evaluation
    Jc = system.jacobian_csr(x)
    dj = abs(Jc - want)
    assert (dj.max() if dj.nnz else 0.0) <= tol * scale
    assert ("jacc" in system.evaluator._csr) == (len(cr) < len(jr))
    system.set_jacobian_layout("reference")
    close(system.jacobian(x), ref.jacobian(x), what="reference layout still served", tol=tol)


def test_compact_jacobian_at_full_size_reaches_the_unique_count():
    """BASELINE configs C2 (brachistochrone 200 x 8: 111 965 -> 78 365 triplets, the number of distinct positions) and C5
    (humanoid 5000 x 8) at full size against the oracle's scatter-added triplets."""
    import scipy.sparse as ssp

    for bname, mesh, K, expect in (("brachistochrone", 200, 8, 78365), ("humanoid_wbc", 5000, 8, None)):
        system, _, guess = getattr(models, bname)(_ns("radau", "pockit_amd"), mesh, K)
        ref, _, _ = getattr(models, bname)(_ns("radau", "oracle"), mesh, K)
        x, lam, sigma = models.bench_inputs(system, guess)
        shape = (system.plan.m, system.plan.n)
        jr, jc = ref.jacobianstructure()
        want = ssp.coo_array((ref.jacobian(x), (jr, jc)), shape=shape).tocsr()
        system.set_jacobian_layout("compact")
        cr, cc = system.jacobianstructure()
        vals = system.jacobian(x)
        if expect is not None:
            assert len(vals) == expect == want.nnz
        got = ssp.coo_array((vals, (cr, cc)), shape=shape).tocsr()
        diff = abs(got - want)
        assert (diff.max() if diff.nnz else 0.0) <= TOL * max(1.0, abs(want).max())
        system._invalidate()


@pytest.mark.parametrize("name", sorted(models.HIGH_ORDER_CASES))
def test_high_order_reference_vectors_with_the_reference_table_recipe(name):
    """13 ... 20 points per interval on the GPU against the REFERENCE's own callback vectors (tests/golden/small_hi): with
    ``collocation.use_reference_recipe()`` -- the reference's np.roots-based tables, which lose digits at these orders --
    the kernels reproduce the reference to the stated 1e-11; with the product's accurate tables the difference is the
    reference's table error (tests/test_tables_hiprec.py)."""
    from pockit_amd import collocation

    builder, scheme, kw = models.HIGH_ORDER_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small_hi", name + ".npz"))
    collocation.use_reference_recipe(True)
    try:
        system, _, _ = builder(_ns(scheme, "pockit_amd"), **kw)
        x, lam, sigma = gold["x"], gold["lam"], float(gold["sigma"])
        jr, jc = system.jacobianstructure()
        assert np.array_equal(jr, gold["jr"]) and np.array_equal(jc, gold["jc"])
        close(system.objective(x), gold["f"], what="f")
        close(system.gradient(x), gold["grad"], what="grad")
        close(system.constraints(x), gold["g"], what="g")
        close(system.jacobian(x), gold["J"], what="J")
        close(system.hessian(x, lam, sigma), gold["H"], what="H")
        f, grad, g, J, H = system.evaluator.cycle(x, lam, sigma)
        close(J, gold["J"], what="cycle J")
        close(H, gold["H"], what="cycle H")
        system._invalidate()
    finally:
        collocation.use_reference_recipe(False)


def test_a_hand_off_that_gives_up_is_an_error_not_a_silent_nan():
    """pk_cycle's finalize workgroup receives the partial sums of its own launch through hand-off slots and gives up after a
    bounded number of polls.  With the bound shortened to ONE poll round (host option ``poll_limit``) it gives up before the
    tile workgroups of a 12k-node launch can have published: the callbacks must raise (error 97, pk_last_error says what
    happened) instead of handing the solver NaN; the library resets the hand-off slots, so the next iterate is right again.
    A NaN that comes from the MODEL still flows through unchecked, as in the reference (examples/_plotting.py:58-63)."""
    import pockit_amd.radau as radau

    system, _, guess = models.planar_quadrotor(radau, 2000, 6)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    lib, h = ev.ctx.lib, ev.ctx.handle
    good = ev.cycle(x, lam, sigma)
    assert np.isfinite(good[0])
    ev.ctx.check(lib.pk_set_host_option(h, b"poll_limit", 1))
    try:
        with pytest.raises(RuntimeError, match="gave up waiting"):
            ev.cycle(x * (1.0 + 1e-9), lam, sigma)
        with pytest.raises(RuntimeError, match="error 97"):
            system.objective(x * (1.0 + 2e-9))         # the solver-side callbacks report it too (f is waited for first)
    finally:
        ev.ctx.check(lib.pk_set_host_option(h, b"poll_limit", 0))
    again = ev.cycle(x, lam, sigma)
    for a, b in zip(again, good):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    close(system.objective(x), good[0], what="f after the reset")
    bad = x.copy()
    bad[5] = np.nan                                    # a state value: the model itself now evaluates to NaN
    f = system.objective(bad)                          # no exception
    assert np.isnan(f) or np.isnan(system.constraints(bad)).any()


@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=200, num_point=8)),        # C2: free t_f (dense columns)
                                  ("two_stage_rocket", "radau", dict(mesh=60, num_point=4)),          # two phases, static parameters
                                  ("planar_quadrotor", "lobatto", dict(mesh=40, num_point=6)),        # LGL: shared end nodes
                                  ("humanoid_wbc", "radau", dict(mesh=30, num_point=8)),
                                  ("lqr", "radau", dict(mesh=7, num_point=12))])                     # K > 8: tables from global memory
@pytest.mark.parametrize("split", ["1", "0"])
def test_compact_layouts_ride_in_the_single_launch_cycle(case, split, monkeypatch):
    """pk_set_cycle_layout: the compact Jacobian / Hessian come out of the SAME single launch as f, grad f and g -- the
    Jacobian role of pk_cycle runs pk_jacc's tile code, its Hessian workgroups pk_hessc's.  f, grad f, g must be bit-identical
    to the reference-layout cycle, the compact values bit-identical to the stand-alone compact kernels, and the matrices
    equal to the oracle's scatter-added triplets; split and unsplit x-part."""
    import scipy.sparse as ssp
    import torch

    monkeypatch.setenv("POCKIT_AMD_SPLIT", split)
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    plan.jacc  # noqa: B018
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)

    def run(nj, nh):
        sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", nj), ("H", nh))
        o = {k: torch.full((max(n, 1),), float("nan"), dtype=torch.float64, device=dev) for k, n in sizes}
        torch.cuda.synchronize()
        for _ in range(2):
            ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o[k].data_ptr() for k, _ in sizes])
        ev.sync()
        return {k: v.cpu().numpy()[:n] for (k, n), v in zip(sizes, o.values())}

    full = run(plan.nnz_J, plan.nnz_H)
    ev.set_cycle_layout(True, True)
    try:
        comp = run(plan.nnz_Jc, plan.nnz_Hc)
    finally:
        ev.set_cycle_layout(False, False)
    again = run(plan.nnz_J, plan.nnz_H)
    for k in ("f", "grad", "g"):
        assert np.array_equal(comp[k], full[k]), k
    for k in ("f", "grad", "g", "J", "H"):
        assert np.array_equal(again[k], full[k]), k + " after switching back"
    assert np.array_equal(comp["J"], ev.jacobian_compact(x)), "compact J: the launch's Jacobian role vs pk_jacc"
    assert np.array_equal(comp["H"], ev.hessian_compact(x, lam, sigma)), "compact H: the launch's Hessian role vs pk_hessc"
    jr, jc = ref.jacobianstructure()
    hr, hc = ref.hessianstructure()
    for got, (r, c), want, (rr, rc_), shape, what in (
            (comp["J"], (plan.jacc_row, plan.jacc_col), ref.jacobian(x), (jr, jc), (plan.m, plan.n), "J"),
            (comp["H"], (plan.hessc_row, plan.hessc_col), ref.hessian(x, lam, sigma), (hr, hc), (plan.n, plan.n), "H")):
        a = ssp.coo_array((got, (r, c)), shape=shape).tocsr()
        b = ssp.coo_array((want, (rr, rc_)), shape=shape).tocsr()
        diff = abs(a - b)
        assert (diff.max() if diff.nnz else 0.0) <= TOL * max(1.0, abs(b).max()), what
    # the solver-side callbacks in the compact Jacobian layout take the same launch (no second kernel, no reference-layout J)
    system.set_jacobian_layout("compact")
    try:
        x2 = x * (1.0 + 1e-9)
        vals = system.jacobian(x2)
        assert np.array_equal(vals, ev.jacobian_compact(x2))
        close(system.constraints(x2), ref.constraints(x2), what="g beside the compact J")
        close(system.gradient(x2), ref.gradient(x2), what="grad f beside the compact J")
    finally:
        system.set_jacobian_layout("reference")
    system._invalidate()


def _all_five(system, x, lam, sigma):
    return (system.objective(x), system.gradient(x), system.constraints(x), system.jacobian(x), system.hessian(x, lam, sigma))


@pytest.mark.parametrize("case", [("three_stage_rocket", "radau", dict(mesh=12, num_point=4)),
                                  ("three_stage_rocket", "lobatto", dict(mesh=9, num_point=5)),
                                  ("three_stage_rocket", "radau", dict(mesh=1000, num_point=4)),      # BASELINE configs[3], literally
                                  ("humanoid_team", "radau", dict(mesh=6, num_point=5)),
                                  ("humanoid_team", "lobatto", dict(mesh=5, num_point=4))])
def test_stand_ins_for_the_literal_baseline_configs_match_the_oracle(case):
    """BASELINE.json words configs[3] as "3 phases x 1000 intervals" and configs[4] as "~40-state"; the reference's example
    programs have 2 phases and 10 states.  Synthetic stand-ins with exactly those shapes --
    benchmarks.three_stage_rocket (3 phases, 12 static parameters, FUNC boundary values and times) and
    benchmarks.humanoid_team (40 states + 20 controls, derivative set evaluated in groups) -- against the oracle: structures
    exactly, every entry of the five callbacks and of the one-launch cycle to 1e-11."""
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    for a, b in zip(system.jacobianstructure() + system.hessianstructure(), ref.jacobianstructure() + ref.hessianstructure()):
        assert np.array_equal(a, b)
    want = _all_five(ref, x, lam, sigma)
    for a, b, what in zip(_all_five(system, x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"{bname} callbacks {what}")
    for a, b, what in zip(system.evaluator.cycle(x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"{bname} cycle {what}")
    ev = system.evaluator
    close(ev.jacobian_direct(x), want[3], what=f"{bname} pk_jac")
    close(ev.hessian_direct(x, lam, sigma), want[4], what=f"{bname} pk_hess")
    system._invalidate()


@pytest.mark.parametrize("case", [("radau", 10, dict(mesh=3, num_point=3)), ("lobatto", 30, dict(mesh=[0, 0.3, 1.0], num_point=[3, 4]))])
def test_more_phases_than_the_default_kernel_arguments_hold(case):
    """The reference puts no limit on the number of phases (systembase.py:148-187).  A code object holds its phase records
    by value in the kernel arguments: 8 by default, as many as the model has beyond that (PK_MAX_PHASES emitted by the code
    generator, up to PK_HOST_MAX_PHASES = 128).  benchmarks.phase_relay -- short phases handed over through static
    parameters -- with 10 and 30 phases against the oracle: structures exactly, callbacks and the one-launch cycle to 1e-11."""
    scheme, n_phases, kw = case
    system, _, guess = models.phase_relay(_ns(scheme, "pockit_amd"), phases=n_phases, **kw)
    ref, _, _ = models.phase_relay(_ns(scheme, "oracle"), phases=n_phases, **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    for a, b in zip(system.jacobianstructure() + system.hessianstructure(), ref.jacobianstructure() + ref.hessianstructure()):
        assert np.array_equal(a, b)
    assert system.evaluator.src.max_phases == n_phases
    want = _all_five(ref, x, lam, sigma)
    for a, b, what in zip(_all_five(system, x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"{n_phases} phases, callbacks {what}")
    for a, b, what in zip(system.evaluator.cycle(x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"{n_phases} phases, cycle {what}")
    system.set_hessian_layout("compact")
    (hr, hc), Hc = system.hessianstructure(), system.hessian(x, lam, sigma)
    rr, rc = ref.hessianstructure()
    A, B = np.zeros((x.size, x.size)), np.zeros((x.size, x.size))
    np.add.at(A, (hr, hc), Hc)
    np.add.at(B, (rr, rc), want[4])
    close(A, B, what=f"{n_phases} phases, compact H")
    system._invalidate()


def test_forty_state_stand_in_at_forty_thousand_nodes():
    """humanoid_team at BASELINE configs[4]'s literal size (40 states, 5000 intervals x 8 points = 40 000 nodes; 24 M Jacobian
    and 29 M Hessian values per cycle).  The oracle needs minutes per callback there, so: entry-by-entry oracle parity at
    300 x 8 (the same code object, tiles and groups), and at 5000 x 8 the size-independent properties -- every output finite,
    the one-launch cycle equal to the five callbacks bit for bit, and H linear in (lambda, sigma)."""
    ns = _ns("radau", "pockit_amd")
    small, _, guess = models.humanoid_team(ns, 300, 8)
    ref, _, _ = models.humanoid_team(_ns("radau", "oracle"), 300, 8)
    x, lam, sigma = models.bench_inputs(small, guess)
    for a, b, what in zip(small.evaluator.cycle(x, lam, sigma), _all_five(ref, x, lam, sigma), ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"300 x 8 cycle {what}")
    small._invalidate()
    system, _, guess = models.humanoid_team(ns, 5000, 8)
    assert system.plan.n == 40 * 40001 + 20 * 40000 + 2
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    f, grad, g, J, H = ev.cycle(x, lam, sigma)
    for v in (grad, g, J, H):
        assert np.isfinite(v).all()
    assert np.array_equal(system.constraints(x), g) and np.array_equal(system.jacobian(x), J)
    assert np.array_equal(system.hessian(x, lam, sigma), H)
    rng = np.random.default_rng(5)
    lam2 = rng.standard_normal(lam.shape)
    H1 = np.array(system.hessian(x, lam, 1.0))
    H2 = np.array(system.hessian(x, lam2, 0.25))
    H12 = np.array(system.hessian(x, 2.0 * lam - 3.0 * lam2, 2.0 * 1.0 - 3.0 * 0.25))
    close(H12, 2.0 * H1 - 3.0 * H2, what="H is linear in (lambda, sigma)", tol=1e-10)
    system._invalidate()


@pytest.mark.parametrize("cap", ["2", "5"])
@pytest.mark.parametrize("case", [("brachistochrone", "radau", dict(mesh=37, num_point=5)),
                                  ("brachistochrone", "lobatto", dict(mesh=23, num_point=6)),
                                  ("two_stage_rocket", "radau", dict(mesh=40, num_point=3)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=19, num_point=4)),
                                  ("planar_quadrotor", "radau", dict(mesh=11, num_point=12)),       # K > 8: staged tables of 256 entries
                                  ("lqr", "radau", dict(mesh=5, num_point=20)),                     # K > 16: tables from global memory
                                  ("humanoid_wbc", "radau", dict(mesh=9, num_point=7)),
                                  ("derivative_model", "radau", {}),                                # nonlinear in the integrals
                                  ("func_times_model", "lobatto", {})])
def test_a_derivative_set_evaluated_in_groups_gives_the_single_pass_results(case, cap, monkeypatch):
    """Segment-grouped evaluation (codegen.split_groups; what a model with hundreds of derivative entries gets by itself)
    forced on small models by a group size of 2 / 5: every role -- stand-alone callbacks, fused x-part, one-launch cycle,
    compact layouts, split and unsplit -- must give the oracle's values and the single-pass code's structures; the groups
    only change which pass of a wave writes a segment's run."""
    import scipy.sparse as ssp

    bname, scheme, kw = case
    ref, _, _ = getattr(models, bname)(_ns(scheme, "oracle"), **kw)
    monkeypatch.setenv("POCKIT_AMD_GROUP_CAP", cap)
    system, _, guess = getattr(models, bname)(_ns(scheme, "pockit_amd"), **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    want = _all_five(ref, x, lam, sigma)
    if isinstance(kw.get("num_point"), int) and kw["num_point"] > 12:
        # (the oracle's np.roots-based tables lose digits beyond 12 points per interval: the NumPy execution of the product's
        #  own plan is the reference there, as for every high-order case; structures still come from the oracle)
        from plan_interp import Interp

        it = Interp(system.plan, x, lam, sigma)
        want = (it.objective(), it.gradient(), it.constraints(), it.jacobian(), it.hessian())
    ev = system.evaluator
    assert ev.src.group_cap == int(cap) and (ev.src.grouped or bname == "lqr")      # (lqr: two entries, one group even so)
    for a, b, what in zip(_all_five(system, x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
        close(a, b, what=f"callbacks {what}")
    close(ev.jacobian_direct(x), want[3], what="pk_jac")
    close(ev.hessian_direct(x, lam, sigma), want[4], what="pk_hess")
    close(ev.constraints_direct(x), want[2], what="pk_g")
    if not system.plan.outer:
        for a, b, what in zip(ev.cycle(x, lam, sigma), want, ("f", "grad", "g", "J", "H")):
            close(a, b, what=f"cycle {what}")
        n, m = system.plan.n, system.plan.m
        jr, jc = ref.jacobianstructure()
        hr, hc = ref.hessianstructure()
        system.set_hessian_layout("compact")
        system.set_jacobian_layout("compact")
        try:
            cr, cc = system.hessianstructure()
            a = ssp.coo_array((system.hessian(x, lam, sigma), (cr, cc)), shape=(n, n)).tocsr()
            b = ssp.coo_array((want[4], (hr, hc)), shape=(n, n)).tocsr()
            d = abs(a - b)
            assert (d.max() if d.nnz else 0.0) <= TOL * max(1.0, abs(b).max()), "compact H"
            cr, cc = system.jacobianstructure()
            a = ssp.coo_array((system.jacobian(x), (cr, cc)), shape=(m, n)).tocsr()
            b = ssp.coo_array((want[3], (jr, jc)), shape=(m, n)).tocsr()
            d = abs(a - b)
            assert (d.max() if d.nnz else 0.0) <= TOL * max(1.0, abs(b).max()), "compact J"
        finally:
            system.set_hessian_layout("reference")
            system.set_jacobian_layout("reference")
    system._invalidate()


def test_writable_results_give_the_reference_semantics():
    """where the landing block keeps the x-independent entries of J, ``jacobian()`` hands out a READ-ONLY array by
    default (a caller scaling it in place would corrupt those entries for every later iterate served from the block);
    ``system.writable_results = True`` gives the reference's semantics -- a writable array per callback -- by filling the
    constant entries in again before a block is reused.  Ten iterates with every J scaled in place and dropped."""
    ns = _ns("radau", "pockit_amd")
    system, _, guess = models.planar_quadrotor(ns, 100, 6)
    ref, _, _ = models.planar_quadrotor(_ns("radau", "oracle"), 100, 6)
    x, lam, sigma = models.bench_inputs(system, guess)
    assert system.evaluator.jac_constant_runs, "the mesh is large enough for constant runs to be kept out of the copy"
    J = system.jacobian(x)
    assert not J.flags.writeable and not system.writable_results
    with pytest.raises(ValueError):
        J *= 2.0
    close(J, ref.jacobian(x), what="J (read-only view)")
    del J
    system.writable_results = True
    rng = np.random.default_rng(11)
    for it in range(10):
        xi = x * (1.0 + 1e-3 * rng.uniform(-1, 1, size=x.shape))
        J, g, grad = system.jacobian(xi), system.constraints(xi), system.gradient(xi)
        assert J.flags.writeable and g.flags.writeable and grad.flags.writeable
        close(J, ref.jacobian(xi), what=f"iterate {it}: J")
        close(g, ref.constraints(xi), what=f"iterate {it}: g")
        J *= -3.0                      # (what a caller of the reference may do with ITS array)
        g[:] = 0.0
        del J, g, grad
    system.writable_results = False
    assert not system.jacobian(x * 1.0000001).flags.writeable
    system._invalidate()


def test_the_cycle_layout_survives_new_tables():
    """pk_set_problem starts every problem in the reference layouts; an Evaluator whose caller had chosen the compact
    layouts for cycle_dev (compact-sized device buffers!) must apply them again after ``set_tables`` -- otherwise the next
    launch writes nnz_J + nnz_H values into buffers sized for nnz_Jc + nnz_Hc."""
    import torch

    from pockit_amd.evaluator import Tables

    system, _, guess = models.brachistochrone(_ns("radau", "pockit_amd"), mesh=37, num_point=5)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    plan.jacc  # noqa: B018
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    sizes = (1, plan.n, plan.m, plan.nnz_Jc, plan.nnz_Hc)
    guard = 4096

    def run():
        o = [torch.full((n + guard,), float("nan"), dtype=torch.float64, device=dev) for n in sizes]
        torch.cuda.synchronize()
        ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[t.data_ptr() for t in o])
        ev.sync()
        return [t.cpu().numpy() for t in o]

    ev.set_cycle_layout(True, True)
    try:
        before = run()
        ev.set_tables(Tables(plan, ev.src))
        after = run()
    finally:
        ev.set_cycle_layout(False, False)
    for a, b, n in zip(after, before, sizes):
        assert np.array_equal(a[:n], b[:n]) and np.isnan(a[n:]).all() and np.isnan(b[n:]).all(), "values beyond the compact sizes were written"
    assert plan.nnz_Hc < plan.nnz_H
    system._invalidate()
