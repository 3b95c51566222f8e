#!/usr/bin/env python3
"""GPU box: the five callbacks of example models at their OWN meshes through the host shim (NumPy in / out, a new x per cycle):
us per callback.  Models: tests/golden/examples/*.model.json.  Usage: example_callbacks_probe.py [--compact] [name ...]
(--compact: the compact Jacobian / Hessian layouts, what the IPOPT adapter uses by default)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import model_io  # noqa: E402

compact = "--compact" in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["orbit_transfer", "rocket_powered_descent", "drone_stabilization", "humanoid_whole_body_control",
                         "neural_ode_xor", "planar_quadrotor", "brachistochrone"]
for name in names:
    with open(os.path.join(ROOT, "tests", "golden", "examples", name + ".model.json")) as fh:
        system = model_io.load_system(json.load(fh))
    plan = system.plan
    if compact:
        try:
            system.set_hessian_layout("compact")
            system.set_jacobian_layout("compact")
        except Exception as exc:  # noqa: BLE001 -- a model nonlinear in the integrals has no compact Hessian
            print(f"{name}: {exc}")
    rng = np.random.default_rng(3)
    x0, lam = rng.uniform(0.6, 1.4, size=plan.n), rng.standard_normal(plan.m)
    calls = [("objective", lambda x: system.objective(x)), ("gradient", lambda x: system.gradient(x)),
             ("constraints", lambda x: system.constraints(x)), ("jacobian", lambda x: system.jacobian(x)),
             ("hessian", lambda x: system.hessian(x, lam, 0.7))]
    acc = {k: [] for k, _ in calls}
    for it in range(260):
        x = x0 * (1.0 + 1e-6 * it)
        for k, fn in calls:
            t0 = time.perf_counter()
            fn(x)
            acc[k].append(time.perf_counter() - t0)
    med = {k: float(np.median(v[60:])) * 1e6 for k, v in acc.items()}
    ev = system.evaluator
    groups = {cb: len(g) for (cb, k), g in ev.src.groups.items() if len(g) > 1}
    print(f"{name:30s} nodes={sum(int(pp.layout.L_m) for pp in plan.phase_plans):5d} n={plan.n:6d} nnz_H={plan.nnz_H:7d} groups={groups or '-'} "
          + "  ".join(f"{k} {v:6.1f}" for k, v in med.items()) + f"  cycle {sum(med.values()):7.1f} us", flush=True)
    system._invalidate()
