#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the *reference* (container only).

Run from the repo root:   python tests/golden/make_golden.py [--full]

The reference at /root/reference is imported unmodified behind two harness shims that live here
(SURVEY.md section 8(c), Appendix B): an identity-``njit`` ``numba`` stand-in
(tests/golden/refharness) and ``typing.Self`` for Python 3.10.  Under the stub the reference's
generated FastFunc modules run as the same array expressions in NumPy.  Nothing of the reference
is copied: only inputs (x, lambda, sigma) and the outputs of its NLP callbacks are stored.

Outputs (committed):
  tests/golden/small/<case>.npz     full callback vectors for the small cases of tests/models.py
  tests/golden/error/<case>.npz     mesh error estimation data, per-interval verdicts and refined meshes
  tests/golden/bangbang/<case>.npz  bang-bang check data and switch-point refinement results
  tests/golden/tables.npz           xw_lgr/I_lgr/xw_lgl/I_lgl for K = 1..12
  tests/golden/full.json            sizes, structure hashes, checksums and strided samples for the
                                    BASELINE.json configs at full size   (--full; takes minutes)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
import types
import typing

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

sys.dont_write_bytecode = True
if not hasattr(typing, "Self"):
    typing.Self = typing.TypeVar("Self")
sys.path.insert(0, os.path.join(HERE, "refharness"))  # stub numba first
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)          # (tests/models.py re-exports pockit_amd.benchmarks: the repo root must be importable)
sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))

import numpy as np  # noqa: E402

import pockit.lobatto as ref_lobatto  # noqa: E402
import pockit.radau as ref_radau  # noqa: E402
from pockit.lobatto.discretization import I_lgl, xw_lgl  # noqa: E402
from pockit.radau.discretization import I_lgr, xw_lgr  # noqa: E402

import models  # noqa: E402

NS = {"radau": ref_radau, "lobatto": ref_lobatto}


def evaluate(system, x, lam, sigma):
    out = {}
    out["f"] = np.float64(system.objective(x.copy()))
    out["grad"] = system.gradient(x.copy())
    out["g"] = system.constraints(x.copy())
    jr, jc = system.jacobianstructure()
    out["jr"], out["jc"] = np.asarray(jr, np.int64), np.asarray(jc, np.int64)
    out["J"] = system.jacobian(x.copy())
    hr, hc = system.hessianstructure()
    out["hr"], out["hc"] = np.asarray(hr, np.int64), np.asarray(hc, np.int64)
    out["H"] = system.hessian(x.copy(), lam, sigma)
    return out


def small_case(name):
    builder, scheme, kw = models.SMALL_CASES[name]
    system, phases, guess = builder(NS[scheme], **kw)
    x, lam, _ = models.bench_inputs(system, guess)
    if name.startswith("derivative_"):
        # the evaluation point of the reference's own FD tests (test_derivative_radau.py:40)
        x = np.arange(int(system.L), dtype=np.float64) / 10 + 1
    sigma = 0.7
    out = evaluate(system, x, lam, sigma)
    hro, hco = system.hessianstructure_o()
    hrc, hcc = system.hessianstructure_c()
    out.update(
        x=x, lam=lam, sigma=np.float64(sigma),
        hro=np.asarray(hro, np.int64), hco=np.asarray(hco, np.int64),
        Ho=system.hessian_o(x.copy()),
        hrc=np.asarray(hrc, np.int64), hcc=np.asarray(hcc, np.int64),
        Hc=system.hessian_c(x.copy(), lam),
        v_lb=system.v_lb, v_ub=system.v_ub, c_lb=system.c_lb, c_ub=system.c_ub,
        n=np.int64(system.L), m=np.int64(len(system.c_lb)),
        l_p=np.asarray(system.l_p, np.int64), r_p=np.asarray(system.r_p, np.int64),
        l_s=np.int64(system.l_s), r_s=np.int64(system.r_s),
        x0=models.pack_guess(system, guess),
    )
    return out


def error_case(name):
    """Mesh error estimation + continuous refinement of the reference on the small case's evaluation point
    (phasebase.py:1339-1437,1522-1617): per phase T_x_aug, I_f_aug, the per-interval verdicts and the refined
    mesh for two tolerance settings."""
    builder, scheme, kw = models.ERROR_CASES[name]
    system, phases, guess = builder(NS[scheme], **kw)
    x, _, _ = models.bench_inputs(system, guess)
    s = x[system.l_s: system.r_s].copy()
    out = {"x": x}
    for k, p in enumerate(phases):
        xp = x[system.l_p[k]: system.r_p[k]].copy()
        T, I = p._error_estimation_data_continuous(xp.copy(), s.copy())
        out[f"T_{k}"], out[f"I_{k}"] = T, I
        for nm in ("l_x", "r_x", "l_u", "r_u", "l_m_aug", "r_m_aug", "t_m_aug", "t_x", "t_u"):
            out[f"{nm}_{k}"] = np.asarray(getattr(p, nm))
        out[f"L_m_aug_{k}"] = np.int64(p.L_m_aug)
        out[f"w_aug_{k}"] = np.concatenate(p.w_aug)
        out[f"P5_{k}"] = p.P(5)
        # interpolation / differentiation matrices of the Variable at sample times that include interior mesh
        # points once and twice (variablebase.py:137-317)
        v0 = NS[scheme].Variable(p, xp.copy())
        tm = v0.t_0 + p._mesh[1:-1] * (v0.t_f - v0.t_0)
        t_out = np.sort(np.concatenate([np.linspace(v0.t_0, v0.t_f, 23), tm, tm[::2]]))
        out[f"tout_{k}"] = t_out
        for nm in ("V_x", "V_u", "D_x", "D_u"):
            out[f"{nm}_{k}"] = getattr(v0, nm)(t_out.copy()).toarray()
        for tag, (atol, rtol) in (("a", (1e-3, 1e-3)), ("b", (1e-7, 1e-6))):
            out[f"ok_{tag}_{k}"] = p._error_check_interval_continuous(T, I, atol, rtol, 1e-4)
            var = NS[scheme].Variable(p, xp.copy())
            mesh0, K0 = p._mesh.copy(), p._num_point.copy()
            p.refine_continuous(var, s.copy() if len(s) else None, atol, rtol, num_point_min=3, num_point_max=7,
                                mesh_length_min=1e-3, mesh_length_max=1.0)
            out[f"mesh_{tag}_{k}"], out[f"K_{tag}_{k}"] = p._mesh.copy(), p._num_point.copy()
            out[f"adapt_{tag}_{k}"] = var.adapt(p).data.copy()     # variablebase.py:365-391
            p.set_discretization(mesh0, K0)          # restore
    return out


def bang_bang_case(name):
    """Bang-bang check and switch-point refinement of the reference (phasebase.py:1368-1400,1439-1474,1619-1868) on a
    prescribed control history: the scaled constraint values, per-interval verdicts, the refined mesh / orders and
    the values adapted to it, for two tolerance settings."""
    kw, profile = models.BANG_BANG_CASES[name]
    system, (p,), guess = models.bang_bang_model(ref_radau, **kw)
    v = guess[0]
    for i, u in enumerate(models.bang_bang_controls(p.t_u, profile)):
        v.u[i] = u
    v.x[0] = np.linspace(0.0, 1.0, len(v.x[0]))
    s = np.array([0.0])
    out = {"data": v.data.copy(), "s": s}
    out["f_bb"] = p._error_estimation_data_discontinuous(v.data.copy(), s.copy())
    for tag, (dtol, kmin, kmax, lmin, lmax) in (("a", (1e-3, 4, 8, 1e-3, 1.0)), ("b", (5e-2, 3, 6, 2e-2, 0.3))):
        out[f"ok_{tag}"] = p._error_check_interval_discontinuous(out["f_bb"], dtol, 1e-4)
        var = ref_radau.Variable(p, v.data.copy())
        mesh0, K0 = p._mesh.copy(), p._num_point.copy()
        p.refine_discontinuous(var, s.copy(), dtol, num_point_min=kmin, num_point_max=kmax, mesh_length_min=lmin,
                               mesh_length_max=lmax)
        out[f"mesh_{tag}"], out[f"K_{tag}"] = p._mesh.copy(), p._num_point.copy()
        out[f"adapt_{tag}"] = var.adapt(p).data.copy()
        p.set_discretization(mesh0, K0)
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def summary(v, k=64):
    v = np.asarray(v, dtype=np.float64)
    idx = np.unique(np.linspace(0, len(v) - 1, k).astype(np.int64)) if len(v) else np.array([], np.int64)
    return {
        "len": int(len(v)),
        "sum": float(np.sum(v)),
        "sumabs": float(np.sum(np.abs(v))),
        "max": float(np.max(np.abs(v))) if len(v) else 0.0,
        "idx": idx.tolist(),
        "samples": v[idx].tolist(),
    }


def full_case(name):
    builder, scheme, kw = models.FULL_CASES[name]
    t0 = time.time()
    system, phases, guess = builder(NS[scheme], **kw)
    t_build = time.time() - t0
    x, lam, sigma = models.bench_inputs(system, guess)
    t0 = time.time()
    out = evaluate(system, x, lam, sigma)
    t_eval = time.time() - t0
    rec = {
        "n": int(system.L), "m": int(len(system.c_lb)),
        "nnz_J": int(len(out["J"])), "nnz_H": int(len(out["H"])),
        "sha_jr": sha(out["jr"].astype(np.int64)), "sha_jc": sha(out["jc"].astype(np.int64)),
        "sha_hr": sha(out["hr"].astype(np.int64)), "sha_hc": sha(out["hc"].astype(np.int64)),
        "x": summary(x), "lam": summary(lam), "sigma": sigma,
        "f": float(out["f"]),
        "grad": summary(out["grad"]), "g": summary(out["g"]),
        "J": summary(out["J"]), "H": summary(out["H"]),
        "ref_build_s": t_build, "ref_eval_s": t_eval,
    }
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also (re)generate full.json")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    os.makedirs(os.path.join(HERE, "small"), exist_ok=True)
    for name in models.SMALL_CASES:
        if args.only and args.only not in name:
            continue
        out = small_case(name)
        np.savez_compressed(os.path.join(HERE, "small", name + ".npz"), **out)
        print(f"{name:24s} n={int(out['n']):5d} m={int(out['m']):5d} "
              f"nnzJ={len(out['J']):6d} nnzH={len(out['H']):6d}")

    os.makedirs(os.path.join(HERE, "error"), exist_ok=True)
    for name in models.ERROR_CASES:
        if args.only and args.only not in name:
            continue
        np.savez_compressed(os.path.join(HERE, "error", name + ".npz"), **error_case(name))
        print("error-estimation fixture", name)

    os.makedirs(os.path.join(HERE, "bangbang"), exist_ok=True)
    for name in models.BANG_BANG_CASES:
        if args.only and args.only not in name:
            continue
        np.savez_compressed(os.path.join(HERE, "bangbang", name + ".npz"), **bang_bang_case(name))
        print("bang-bang fixture", name)

    tabs = {}
    for K in range(1, 13):
        x, w = xw_lgr(K)
        tabs[f"lgr_x_{K}"], tabs[f"lgr_w_{K}"], tabs[f"lgr_I_{K}"] = x, w, I_lgr(K)
        x, w = xw_lgl(K)
        tabs[f"lgl_x_{K}"], tabs[f"lgl_w_{K}"] = x, w
        if K >= 2:
            tabs[f"lgl_I_{K}"] = I_lgl(K)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tabs)

    # the reference's tables beyond K = 12 (its np.roots-based nodes lose digits with K: the product's Newton-polished
    # tables are compared with these AND with the multiprecision tables of make_hiprec.py, tests/test_product_host.py)
    tabs = {}
    for K in range(13, 25):
        x, w = xw_lgr(K)
        tabs[f"lgr_x_{K}"], tabs[f"lgr_w_{K}"], tabs[f"lgr_I_{K}"] = x, w, I_lgr(K)
        x, w = xw_lgl(K)
        tabs[f"lgl_x_{K}"], tabs[f"lgl_w_{K}"], tabs[f"lgl_I_{K}"] = x, w, I_lgl(K)
    np.savez_compressed(os.path.join(HERE, "tables_hi.npz"), **tabs)

    # callback vectors of the reference on meshes with 13 ... 20 points per interval
    os.makedirs(os.path.join(HERE, "small_hi"), exist_ok=True)
    for name, (builder, scheme, kw) in models.HIGH_ORDER_CASES.items():
        if args.only and args.only not in name:
            continue
        system, phases, guess = builder(NS[scheme], **kw)
        x, lam, _ = models.bench_inputs(system, guess)
        out = evaluate(system, x, lam, 0.7)
        out.update(x=x, lam=lam, sigma=np.float64(0.7))
        np.savez_compressed(os.path.join(HERE, "small_hi", name + ".npz"), **out)
        print("high-order fixture", name, len(out["J"]), len(out["H"]))

    if args.full:
        path = os.path.join(HERE, "full.json")
        full = {}
        if os.path.exists(path):
            full = json.load(open(path))
        for name in models.FULL_CASES:
            if args.only and args.only not in name:
                continue
            full[name] = full_case(name)
            print(name, {k: full[name][k] for k in ("n", "m", "nnz_J", "nnz_H", "ref_build_s", "ref_eval_s")})
            json.dump(full, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
