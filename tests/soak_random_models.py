#!/usr/bin/env python3
"""Soak run (not collected by pytest): MANY seeded random models (tests/random_models.py) beyond the fixtures, the HIP kernels
against the oracle on the same inputs -- structures exactly, the five callbacks, the stand-alone kernels, the one-launch cycle
and the compact layouts (scatter-added) to 1e-11; every model once with the default code generation and once with its
derivative set in groups of three and every pass a workgroup of its own.

    python tests/soak_random_models.py --compile-only --jobs 7 200 260     # build container: fills the code-object cache
    python tests/soak_random_models.py 200 260                             # GPU box: seeds 200 ... 259, both schemes
"""
import argparse
import importlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import numpy as np  # noqa: E402

import random_models as rm  # noqa: E402

TOL = 1e-11
VARIANTS = ({}, {"POCKIT_AMD_GROUP_CAP": "3", "POCKIT_AMD_PASS_PARALLEL": "1"})


def cases(lo, hi, scale):
    return [(scheme, seed, scale) for seed in range(lo, hi) for scheme in ("radau", "lobatto")]


def set_variant(v):
    for key in ("POCKIT_AMD_GROUP_CAP", "POCKIT_AMD_PASS_PARALLEL"):
        os.environ.pop(key, None)
    os.environ.update(v)


def compile_one(job):
    scheme, seed, scale = job
    from pockit_amd.evaluator import compile_plan

    from pockit_amd import hipbuild

    keys = []
    for v in VARIANTS:
        set_variant(v)
        system, _ = rm.random_model(importlib.import_module(f"pockit_amd.{scheme}"), seed, scheme, mesh_scale=scale)
        src, _ = compile_plan(system.plan)
        keys.append(hipbuild._key(src.source, system.plan.system._fastmath))
    return keys


def err(a, b):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def dense(n, r, c, v):
    m = np.zeros((n, n))
    np.add.at(m, (np.asarray(r), np.asarray(c)), np.asarray(v))
    return m


def check_one(job):
    scheme, seed, scale = job
    ref, _ = rm.random_model(importlib.import_module(f"oracle.{scheme}"), seed, scheme, mesh_scale=scale)
    x, lam, sigma = rm.random_inputs(ref, seed)
    want = dict(f=ref.objective(x.copy()), grad=ref.gradient(x.copy()), g=ref.constraints(x.copy()), J=ref.jacobian(x.copy()),
                H=ref.hessian(x.copy(), lam, sigma))
    rjs, rhs = ref.jacobianstructure(), ref.hessianstructure()
    worst, what = 0.0, ""
    for v in VARIANTS:
        set_variant(v)
        system, _ = rm.random_model(importlib.import_module(f"pockit_amd.{scheme}"), seed, scheme, mesh_scale=scale)
        tag = "groups" if v else "default"
        js, hs = system.jacobianstructure(), system.hessianstructure()
        if not all(np.array_equal(a, b) for a, b in zip(js + hs, rjs + rhs)):
            return float("inf"), f"{tag}: structure"
        for key in ("v_lb", "v_ub", "c_lb", "c_ub"):
            if not np.array_equal(getattr(system, key), getattr(ref, key)):
                return float("inf"), f"{tag}: {key}"
        ev = system.evaluator
        got = [("f", system.objective(x)), ("grad", system.gradient(x)), ("g", system.constraints(x)), ("J", system.jacobian(x)),
               ("H", system.hessian(x, lam, sigma)), ("f", ev.objective_direct(x)), ("grad", ev.gradient_direct(x)),
               ("g", ev.constraints_direct(x)), ("J", ev.jacobian_direct(x)), ("H", ev.hessian_direct(x, lam, sigma))]
        got += list(zip(("f", "grad", "g", "J", "H"), ev.cycle(x, lam, sigma)))
        for k, (key, val) in enumerate(got):
            e = err(val, want[key])
            if e > worst:
                worst, what = e, f"{tag}: {key} ({('callbacks', 'stand-alone', 'cycle')[k // 5]})"
        n, m = x.size, lam.size
        if ev.src.compact:
            system.set_hessian_layout("compact")
            (r, c), val = system.hessianstructure(), system.hessian(x, lam, sigma)
            e = err(dense(n, r, c, val), dense(n, rhs[0], rhs[1], want["H"]))
            if e > worst:
                worst, what = e, f"{tag}: compact H"
            system.set_hessian_layout("reference")
        if ev.src.compact_j:
            system.set_jacobian_layout("compact")
            (r, c), val = system.jacobianstructure(), system.jacobian(x)
            e = err(dense(max(n, m), r, c, val), dense(max(n, m), rjs[0], rjs[1], want["J"]))
            if e > worst:
                worst, what = e, f"{tag}: compact J"
            system.set_jacobian_layout("reference")
        system._invalidate()
    return worst, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lo", type=int)
    ap.add_argument("hi", type=int)
    ap.add_argument("--scale", type=int, default=4, help="mesh intervals x this factor (default 4: 4 ... 20 intervals per phase)")
    ap.add_argument("--compile-only", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--keys", default=None, help="with --compile-only: append the cache keys of the code objects to this file")
    args = ap.parse_args()
    jobs = cases(args.lo, args.hi, args.scale)
    t0 = time.time()
    if args.compile_only:
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(args.jobs) as pool:
            done = 0
            for keys in pool.imap_unordered(compile_one, jobs):
                done += len(keys)
                if args.keys:
                    with open(args.keys, "a") as fh:
                        fh.write("\n".join(keys) + "\n")
                if done % 20 == 0:
                    print(f"{done} code objects, {time.time() - t0:.0f} s", flush=True)
        print(f"compiled / found {done} code objects in {time.time() - t0:.0f} s")
        return 0
    bad, worst = [], (0.0, "", None)
    for k, job in enumerate(jobs):
        try:
            e, what = check_one(job)
        except Exception as exc:  # noqa: BLE001 -- the soak goes on and reports
            e, what = float("inf"), repr(exc)
        if not e <= TOL:
            bad.append((job, e, what))
            print("FAILED", job, e, what, flush=True)
        if e > worst[0] and np.isfinite(e):
            worst = (e, what, job)
        if (k + 1) % 20 == 0:
            print(f"{k + 1} / {len(jobs)} models, worst so far {worst[0]:.2e} ({worst[1]}, {worst[2]}), {time.time() - t0:.0f} s", flush=True)
    print(f"{len(jobs)} models x {len(VARIANTS)} code generations: {len(bad)} failed; worst relative error {worst[0]:.3e} "
          f"({worst[1]}, {worst[2]}); tolerance {TOL}; {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
