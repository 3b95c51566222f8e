"""CPU tests of the product's host side: function compiler ordering rule, collocation tables,
mesh layout, modeling-API behaviour, transcription compiler (layout + value plan).

The known-answer values restate the reference's own tests
(/root/reference/tests/test_base/test_fastfunc.py:33-71, tests/test_radau/test_discretization_radau.py:5-58,
135-189, tests/test_labatto/test_discretization_lobatto.py, tests/test_radau/test_bound_radau.py:7-44,
tests/test_base/test_system_base.py:73-104) against this package's code; golden vectors come from
tests/golden (captured from the reference).  No GPU is needed: the evaluation plan is executed by the
NumPy interpreter of tests/plan_interp.py.
"""
import os

import numpy as np
import pytest
import sympy as sp

import models
import pockit_amd.lobatto as lobatto
import pockit_amd.radau as radau
from plan_interp import Interp
from pockit_amd import collocation
from pockit_amd.layout import MeshLayout
from pockit_amd.symbolic import SparseFunc

HERE = os.path.dirname(os.path.abspath(__file__))
NS = {"radau": radau, "lobatto": lobatto}
NONLINEAR_IN_I = ("derivative_", "functimes_")


def close(a, b, tol=1e-11):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    if a.size:
        assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))


# ------------------------------------------------------------------ function compiler
def test_sparse_func_ordering_rule():
    x, y = sp.symbols("x, y")
    f = SparseFunc(x + y**2, [x, y])
    assert f.G_index.tolist() == [0, 1] and f.H_index_row.tolist() == [1] and f.H_index_col.tolist() == [1]
    f = SparseFunc(x * y + y**3, [x, y])
    assert f.G_index.tolist() == [0, 1]
    assert f.H_index_row.tolist() == [1, 1] and f.H_index_col.tolist() == [0, 1]
    f = SparseFunc(x**2 * y + y**3, [x, y])
    assert f.H_index_row.tolist() == [0, 1, 1] and f.H_index_col.tolist() == [0, 0, 1]
    assert [str(e) for e in f.hess] == ["2*y", "2*x", "6*y"]
    f = SparseFunc(1, [x])
    assert len(f.G_index) == 0 and len(f.H_index_row) == 0
    SparseFunc(x + y**2, [x, y, sp.Symbol("z")], simplify=True)
    with pytest.raises(ValueError):
        SparseFunc(sp.Symbol("q"), [x])


# ------------------------------------------------------------------ collocation tables
def test_lgr_known_nodes_and_weights():
    x, w = collocation.lgr_nodes_weights(3)
    assert np.allclose(x, [-1.0, -0.289898, 0.689898]) and np.allclose(w, [0.222222, 1.02497, 0.752806])
    x, w = collocation.lgr_nodes_weights(5)
    assert np.allclose(x, [-1.0, -0.72048, -0.167181, 0.446314, 0.885792])
    assert np.allclose(w, [0.08, 0.446208, 0.623653, 0.562712, 0.287427])
    x, w = collocation.lgr_nodes_weights(1)
    assert np.allclose(x, [-1.0]) and np.allclose(w, [2.0])


def test_integration_matrices_integrate_from_plus_one():
    for K in (10, 20):
        x, _ = collocation.lgr_nodes_weights(K)
        assert np.allclose(collocation.lgr_integration_matrix(K) @ np.cos(x), np.sin(x) - np.sin(1))
        x, _ = collocation.lgl_nodes_weights(K)
        assert np.allclose(collocation.lgl_integration_matrix(K) @ (2 * x), (x**2 - 1)[:-1])


def test_tables_match_reference_fixture():
    gold = np.load(os.path.join(HERE, "golden", "tables.npz"))
    for K in range(1, 13):
        tol = 1e-13 if K <= 8 else 5e-12
        x, w = collocation.lgr_nodes_weights(K)
        close(x, gold[f"lgr_x_{K}"], tol)
        close(w, gold[f"lgr_w_{K}"], tol)
        close(collocation.lgr_integration_matrix(K), gold[f"lgr_I_{K}"], tol)
        x, w = collocation.lgl_nodes_weights(K)
        close(x, gold[f"lgl_x_{K}"], tol)
        close(w, gold[f"lgl_w_{K}"], tol)
        if K >= 2:
            close(collocation.lgl_integration_matrix(K), gold[f"lgl_I_{K}"], tol)


# ------------------------------------------------------------------ mesh layout
def test_lgr_layout_known_answers():
    lay = MeshLayout("lgr", np.array([0.0, 0.1, 1.0]), np.array([2, 3]), 1, 1)
    assert lay.l_v.tolist() == [0, 6] and lay.r_v.tolist() == [6, 11]
    assert lay.lm.tolist() == [0, 2] and lay.rm.tolist() == [2, 5] and lay.L_m == 5
    assert lay.l_d.tolist() == [0] and lay.r_d.tolist() == [5]
    x2, w2 = collocation.lgr_nodes_weights(2)
    x3, w3 = collocation.lgr_nodes_weights(3)
    assert np.allclose(lay.tau, np.concatenate([(x2 + 1) / 2 * 0.1, 0.1 + (x3 + 1) / 2 * 0.9]))
    assert np.allclose(lay.w, np.concatenate([w2 / 2 * 0.1, w3 / 2 * 0.9]))
    # translation matrix acting on arange: front / middle / back entries together
    v = np.arange(lay.L_m + 1, dtype=float)
    Tx = np.zeros(lay.L_d)
    r, c, d = lay.T_mid_structure()
    np.add.at(Tx, r, d * v[c])
    np.add.at(Tx, lay.Tf_row, lay.Tf_val * v[0])
    np.add.at(Tx, lay.Tb_row, lay.Tb_val * v[lay.L_m])
    assert np.allclose(Tx, [-2, -1, -3, -2, -1])


def test_lgl_layout_shares_interval_ends():
    lay = MeshLayout("lgl", np.array([0.0, 0.1, 1.0]), np.array([2, 3]), 1, 1)
    assert lay.L_m == 4 and lay.L_d == 3 and lay.lm.tolist() == [0, 1]
    assert lay.l_v.tolist() == [0, 4] and lay.r_v.tolist() == [4, 8]
    # the shared node accumulates both intervals' weights
    _, w2 = collocation.lgl_nodes_weights(2)
    _, w3 = collocation.lgl_nodes_weights(3)
    assert np.isclose(lay.w[1], w2[1] / 2 * 0.1 + w3[0] / 2 * 0.9)


def test_tiles_cover_every_interval_once():
    rng = np.random.default_rng(0)
    K = rng.integers(1, 9, size=57)
    mesh = np.concatenate(([0.0], np.cumsum(rng.uniform(0.1, 1, size=57))))
    mesh /= mesh[-1]
    for scheme, KK in (("lgr", K), ("lgl", K + 1)):
        lay = MeshLayout(scheme, mesh, KK, 2, 1)
        for ipw in (None, 1, 3):
            t = lay.tiles(ipw)
            assert t[0, 0] == 0 and np.all(t[1:, 0] == t[:-1, 0] + t[:-1, 1]) and t[-1, 0] + t[-1, 1] == lay.N
            nodes = t[:, 1] * (lay.stride[t[:, 0]]) + (1 if scheme == "lgl" else 0)
            assert nodes.max() <= 64


def test_magic_numbers_divide_exactly_including_the_unit_divisor():
    """The kernels replace p // d (d = defect rows, integration entries, translation entries per interval) by
    magic_div(p, magic): exact for every p a tile can produce, d = 1 included (LGR K = 1, LGL K = 2)."""
    from pockit_amd.evaluator import Tables, magic_div, magic_number
    from pockit_amd.codegen import ModelSource

    p = np.arange(1 << 16, dtype=np.uint64)
    for d in list(range(1, 700)) + [1023, 1024, 4095, 4096]:
        mg = magic_number(d)
        assert 0 <= mg < 1 << 32
        q = p if mg == 0 else (p * np.uint64(mg)) >> np.uint64(32)
        assert np.array_equal(q, p // np.uint64(d)), d
        assert magic_div(65535, mg) == 65535 // d
    # the tables of unit-divisor meshes: several intervals per tile, magic 0, every position decodes correctly
    for ns, K in ((radau, 1), (lobatto, 2)):
        system, _, _ = models.brachistochrone(ns, 800, K)
        tb = Tables(system.plan, ModelSource(system.plan))
        live = tb.tiles[tb.tiles["nj"] > 0]
        assert live["nj"].max() >= 3
        for t in live[:5]:
            for field, d in (("magicI", int(t["nnzI"])), ("magicT", int(t["nnzT"]))):
                for pos in range(int(t["nj"]) * d):
                    assert magic_div(pos, int(t[field])) == pos // d
            assert int(t["magicR"]) == 0      # R = 1


# ------------------------------------------------------------------ modeling API behaviour
def test_variable_and_constraint_bounds_layout():
    s = radau.System(4)
    p = s.new_phase(2, 2)
    p.set_dynamics([0, 0]).set_boundary_condition([0, 0], [s.s[0], 0], None, s.s[2]).set_discretization(
        [0, 0.2, 1], [3, 4]).set_phase_constraint([p.x[0], p.u[1], p.t, p.s[3]], [2, 4, 6, 8], [3, np.inf, 7, 9])
    s.set_phase([p]).set_objective(0).set_system_constraint([s.s[1]], [0], [1])
    lb = [2] * 8 + [-np.inf] * 8 + [-np.inf] * 7 + [4] * 7 + [6] * 2 + [2, 0, 6, 8]
    ub = [3] * 8 + [np.inf] * 8 + [np.inf] * 7 + [np.inf] * 7 + [7] * 2 + [3, 1, 7, 9]
    assert np.allclose(lb, s.v_lb) and np.allclose(ub, s.v_ub)

    s = radau.System(2)
    p = s.new_phase(2, 2)
    p.set_dynamics([0, 0]).set_boundary_condition([0, 0], [s.s[0], 0], None, 1).set_discretization(
        [0, 0.2, 1], [3, 4]).set_phase_constraint([p.x[0], p.u[1], p.x[0] + p.u[1]], [2, 4, -1], [3, np.inf, 1])
    p2 = s.new_phase(1, 1)
    p2.set_dynamics([0]).set_discretization(4, 4).set_boundary_condition([0], [s.s[0] * 0.1], None, 3 * s.s[1]
                                                                         ).set_phase_constraint([p2.x[0], p2.t], [0, 1], [0, 2])
    s.set_phase([p, p2]).set_objective(0).set_system_constraint([s.s[1], s.s[0] + s.s[1]], [0, -2], [1, 2])
    assert np.allclose([-2, 0, 1] + [0] * 14 + [-1] * 7 + [0] * 16, s.c_lb)
    assert np.allclose([2, 0, 2] + [0] * 14 + [1] * 7 + [0] * 16, s.c_ub)


@pytest.mark.parametrize("ns,kmin", [(radau, 1), (lobatto, 2)])
def test_discretization_validation_is_atomic(ns, kmin):
    system = ns.System(0)
    phase = system.new_phase(1, 0)
    phase.set_dynamics([0]).set_boundary_condition([0], [0], 0, 1)
    phase.set_discretization(1, max(kmin, 3))
    before = (phase._mesh.copy(), phase._num_point.copy(), phase.layout)
    for mesh, k in [(0, 3), ([0], [3]), ([0, 0], [3]), ([1, 0], [3]), ([0, np.inf], [3]), ([0, 0.5, 1], [3]),
                    ([0, 1], [kmin - 1]), ([0, 1], [2.5])]:
        with pytest.raises(ValueError):
            phase.set_discretization(mesh, k)
        assert phase.ok and phase.layout is before[2]
        assert np.array_equal(phase._mesh, before[0]) and np.array_equal(phase._num_point, before[1])


def test_setter_errors_match_reference_messages():
    s = radau.System(0)
    p = s.new_phase(2, 1)
    with pytest.raises(ValueError, match="number of dynamics"):
        p.set_dynamics([0])
    with pytest.raises(ValueError, match="same length"):
        p.set_phase_constraint([p.x[0]], [0, 1], [1])
    with pytest.raises(ValueError, match="same length"):
        p.set_boundary_condition([0], [0, 0], 0, 1)
    with pytest.raises(ValueError, match="reserved for time"):
        s.new_phase(["t"], 1)
    with pytest.raises(ValueError, match="not fully set"):
        s.set_phase([p])
    with pytest.raises(ValueError):
        radau.System(1.5)


# ------------------------------------------------------------------ transcription compiler
@pytest.mark.parametrize("name", sorted(n for n in models.SMALL_CASES if n not in models.SLOW_ON_CPU))
def test_plan_reproduces_reference(name):
    """Triplet structure identical to the reference; the evaluation plan (executed in NumPy exactly
    as the kernels consume it) reproduces the reference's f, grad f, g, J, H."""
    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, guess = builder(NS[scheme], **kw)
    plan = system.plan
    assert (plan.n, plan.m) == (int(gold["n"]), int(gold["m"]))
    assert np.array_equal(plan.jac_row, gold["jr"]) and np.array_equal(plan.jac_col, gold["jc"])
    assert np.array_equal(plan.hess_row, gold["hr"]) and np.array_equal(plan.hess_col, gold["hc"])
    for k in ("v_lb", "v_ub", "c_lb", "c_ub"):
        assert np.array_equal(getattr(plan, k), gold[k])
    assert np.allclose(models.pack_guess(system, guess), gold["x0"], rtol=0, atol=1e-13)
    it = Interp(plan, gold["x"], gold["lam"], float(gold["sigma"]))
    close(it.objective(), gold["f"])
    close(it.gradient(), gold["grad"])
    close(it.constraints(), gold["g"])
    close(it.jacobian(), gold["J"])
    close(it.hessian(), gold["H"])


def test_nonlinear_in_integrals_uses_outer_blocks():
    """Objective (I0+I1+s0)^2 and constraint s1/2*I0 of the reference's FD test model need the
    outer-product Hessian blocks (easyderiv.py:323-459); their counts must add up to the layout."""
    system, _, _ = models.derivative_model(radau)
    plan = system.plan
    assert plan.outer and plan.hess.needs_I and plan.jac.needs_I and plan.needs_I_grad
    covered = sum(b.count for b in plan.outer) + len(plan.hess.items)
    covered += sum(pp.layout.L_mid for k, pp in enumerate(plan.phase_plans) for sg in plan.hess.segs[k] if sg.kind == "N")
    covered += sum(pp.layout.nnzI_mid for k, pp in enumerate(plan.phase_plans) for sg in plan.hess.segs[k] if sg.kind == "I")
    assert covered == plan.nnz_H


@pytest.mark.parametrize("name", sorted(n for n in models.SMALL_CASES if not n.startswith(NONLINEAR_IN_I) and n not in models.SLOW_ON_CPU))
def test_compact_hessian_plan_coalesces_to_the_reference_matrix(name):
    """Compact layout (mu = I^T lambda, entries of a node summed per position): fewer triplets, same matrix
    as the scatter-add of the reference's triplets (the accumulation IPOPT performs)."""
    import scipy.sparse as ssp

    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, _ = builder(NS[scheme], **kw)
    plan = system.plan
    plan.hessc  # noqa: B018
    assert np.all(plan.hessc_row >= plan.hessc_col) and plan.nnz_Hc <= plan.nnz_H
    it = Interp(plan, gold["x"], gold["lam"], float(gold["sigma"]))
    n = plan.n
    want = ssp.coo_array((gold["H"], (gold["hr"], gold["hc"])), shape=(n, n)).toarray()
    got = ssp.coo_array((it.hessian_compact(), (plan.hessc_row, plan.hessc_col)), shape=(n, n)).toarray()
    close(got, want)
    if "brach" in name:                 # K-fold (and more) reduction where the dynamics are nonlinear
        assert plan.nnz_Hc * 4 <= plan.nnz_H


@pytest.mark.parametrize("name", sorted(n for n in models.SMALL_CASES if n not in models.SLOW_ON_CPU))
def test_compact_jacobian_plan_coalesces_to_the_reference_matrix(name):
    """Compact Jacobian layout (dense-column entries of the dynamics contracted with the integration block, scalar items
    that meet on one position summed): no more triplets than the reference, every (row, column) of the reference and no
    other, and the same matrix as the scatter-add of the reference's triplets; exactly one triplet per position wherever
    no state's translation entry meets its own d f_i / d x_i and no system constraint depends on an integral."""
    import scipy.sparse as ssp

    builder, scheme, kw = models.SMALL_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small", name + ".npz"))
    system, _, _ = builder(NS[scheme], **kw)
    plan = system.plan
    plan.jacc  # noqa: B018
    assert plan.nnz_Jc <= plan.nnz_J
    ref_pos = set(zip(gold["jr"].tolist(), gold["jc"].tolist()))
    got_pos = set(zip(plan.jacc_row.tolist(), plan.jacc_col.tolist()))
    assert got_pos == ref_pos
    it = Interp(plan, gold["x"])
    shape = (plan.m, plan.n)
    want = ssp.coo_array((gold["J"], (gold["jr"], gold["jc"])), shape=shape).toarray()
    got = ssp.coo_array((it.jacobian_compact(), (plan.jacc_row, plan.jacc_col)), shape=shape).toarray()
    close(got, want)
    self_dependent = any(i in fn.G_index.tolist() for pp in plan.phase_plans for i, fn in enumerate(pp.phase.F_d))
    if not self_dependent and not plan.needs_I_con:
        assert plan.nnz_Jc == len(ref_pos)
    system.set_jacobian_layout("compact")
    jr, jc = system.jacobianstructure()
    assert len(jr) == plan.nnz_Jc
    system.set_jacobian_layout("reference")
    assert len(system.jacobianstructure()[0]) == plan.nnz_J
    with pytest.raises(ValueError):
        system.set_jacobian_layout("dense")
