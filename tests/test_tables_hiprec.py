"""Collocation tables at every order against an independent multiprecision pin, and against the reference beyond K = 12.

``tests/golden/hiprec_tables.npz`` (generator: tests/golden/make_hiprec.py, mpmath, 100-450 digits, nothing of the product
or the reference involved) holds LGR / LGL nodes, weights and integration matrices for K = 2 ... 128 rounded to float64.
``tests/golden/tables_hi.npz`` holds the REFERENCE's tables for K = 13 ... 24 (tests/golden/make_golden.py:
radau/discretization.py:89-114,185-196, lobatto/discretization.py:80-110,155-166), ``tests/golden/small_hi`` its callback
vectors on meshes with 13 ... 20 points per interval.

What this pins:
  * the product's tables (pockit_amd/collocation.py, recurrence + Newton) are right to a few ulp up to K = 128 -- so the
    GPU tests of 64 < K <= 256 (which execute the product's own plan in NumPy) compare against tables that are known;
  * the reference's tables lose digits with K (np.roots of a monomial-basis polynomial): 2e-12 at K = 12, 8e-11 at K = 16,
    4e-9 at K = 20 -- the product differs from the reference by exactly that, never by more (triangle inequality, asserted);
  * ``collocation.use_reference_recipe()`` reproduces the reference's tables to rounding, and with it the reference's callback
    vectors at K = 13 ... 20 to the stated 1e-11; with the accurate tables the difference to the reference's vectors stays
    within a tolerance derived from the measured table error of the reference at that K.
"""
import os

import numpy as np
import pytest

import models
import pockit_amd.lobatto as lobatto
import pockit_amd.radau as radau
from plan_interp import Interp
from pockit_amd import collocation

HERE = os.path.dirname(os.path.abspath(__file__))
HP = np.load(os.path.join(HERE, "golden", "hiprec_tables.npz"))
REF_LO = np.load(os.path.join(HERE, "golden", "tables.npz"))
REF_HI = np.load(os.path.join(HERE, "golden", "tables_hi.npz"))
ORDERS = sorted({int(k.split("_")[-1]) for k in HP.files})
NS = {"radau": radau, "lobatto": lobatto}


def product_tables(tag, K):
    if tag == "lgr":
        (x, w), I = collocation.lgr_nodes_weights(K), collocation.lgr_integration_matrix(K)
    else:
        (x, w), I = collocation.lgl_nodes_weights(K), collocation.lgl_integration_matrix(K)
    return {"x": x, "w": w, "I": I}


def err(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


@pytest.fixture
def reference_recipe():
    collocation.use_reference_recipe(True)
    yield
    collocation.use_reference_recipe(False)


@pytest.mark.parametrize("K", ORDERS)
def test_product_tables_equal_the_multiprecision_tables(K):
    for tag in ("lgr", "lgl"):
        got = product_tables(tag, K)
        assert err(got["x"], HP[f"{tag}_x_{K}"]) <= 4e-16, (tag, "nodes")
        assert err(got["w"], HP[f"{tag}_w_{K}"]) <= 6e-14, (tag, "weights")       # (measured: 3.4e-14 at K = 128, LGR)
        assert err(got["I"], HP[f"{tag}_I_{K}"]) <= 2e-15, (tag, "integration matrix")
        assert got["I"].shape == HP[f"{tag}_I_{K}"].shape


@pytest.mark.parametrize("K", [8, 12, 13, 14, 15, 16, 17, 18, 19, 20, 24])
def test_product_differs_from_the_reference_by_the_reference_table_error_only(K):
    ref = REF_LO if K <= 12 else REF_HI
    worst_ref = 0.0
    for tag in ("lgr", "lgl"):
        got = product_tables(tag, K)
        for part in ("x", "w", "I"):
            truth = HP[f"{tag}_{part}_{K}"]
            e_prod, e_ref = err(got[part], truth), err(ref[f"{tag}_{part}_{K}"], truth)
            worst_ref = max(worst_ref, e_ref)
            # |product - reference| <= |product - truth| + |reference - truth|  (+ the rounding of the stored truth)
            assert err(got[part], ref[f"{tag}_{part}_{K}"]) <= e_prod + e_ref + 2e-16, (tag, part)
            assert e_prod <= 6e-14
    if K >= 16:
        assert worst_ref >= 1e-11          # the reference's tables are the inaccurate side there (8e-11 ... 2e-7)


@pytest.mark.parametrize("K", [3, 8, 12, 13, 16, 20, 24])
def test_reference_recipe_reproduces_the_reference_tables(K, reference_recipe):
    ref = REF_LO if K <= 12 else REF_HI
    for tag in ("lgr", "lgl"):
        got = product_tables(tag, K)
        assert np.array_equal(got["x"], ref[f"{tag}_x_{K}"]) and np.array_equal(got["w"], ref[f"{tag}_w_{K}"])
        assert err(got["I"], ref[f"{tag}_I_{K}"]) <= 4e-16


def _table_error_of_the_reference(scheme, orders):
    tag = "lgr" if scheme == "radau" else "lgl"
    return max(err(REF_HI[f"{tag}_{part}_{K}"], HP[f"{tag}_{part}_{K}"]) for K in orders if K >= 13 for part in ("x", "w", "I"))


@pytest.mark.parametrize("name", sorted(models.HIGH_ORDER_CASES))
def test_high_order_cases_match_the_reference_vectors(name):
    """The plan executed in NumPy (tests/plan_interp.py) on the reference's own evaluation point at 13 ... 20 points per
    interval: with the reference's table recipe the reference's f, grad f, g, J, H to 1e-11; with the product's accurate
    tables to a tolerance derived from the reference's measured table error at these orders (x 4; measured on these cases:
    0.04 ... 0.4 times the largest table error)."""
    builder, scheme, kw = models.HIGH_ORDER_CASES[name]
    gold = np.load(os.path.join(HERE, "golden", "small_hi", name + ".npz"))
    orders = sorted(set(np.atleast_1d(kw["num_point"]).tolist()))
    table_err = _table_error_of_the_reference(scheme, orders)
    for recipe, tol in (("reference", 1e-11), ("accurate", 4 * table_err)):
        collocation.use_reference_recipe(recipe == "reference")
        try:
            system, _, _ = builder(NS[scheme], **kw)
            plan = system.plan
            assert np.array_equal(plan.jac_row, gold["jr"]) and np.array_equal(plan.jac_col, gold["jc"])
            assert np.array_equal(plan.hess_row, gold["hr"]) and np.array_equal(plan.hess_col, gold["hc"])
            it = Interp(plan, gold["x"], gold["lam"], float(gold["sigma"]))
            for got, key in ((it.objective(), "f"), (it.gradient(), "grad"), (it.constraints(), "g"), (it.jacobian(), "J"),
                             (it.hessian(), "H")):
                want = gold[key]
                assert err(got, want) <= tol * max(1.0, float(np.max(np.abs(want)))), (recipe, key, err(got, want), tol)
        finally:
            collocation.use_reference_recipe(False)
