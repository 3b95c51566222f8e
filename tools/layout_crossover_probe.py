#!/usr/bin/env python3
"""GPU box: the five callbacks through the host shim (NumPy in / out, a new x per iterate) on the reference layouts and on the
compact layouts, over mesh sizes -- where does shipping fewer bytes start to pay for the compact kernels' longer chain?
(decides optimizer.ipopt.solve's layout="auto")"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from pockit_amd import benchmarks, radau  # noqa: E402

CASES = [("planar_quadrotor", m, 6) for m in (50, 150, 300, 600, 1000, 2000)] + [("humanoid_wbc", m, 8) for m in (25, 100, 300, 1000)] + \
        [("brachistochrone", m, 8) for m in (20, 200, 1250)]
for name, mesh, K in CASES:
    system, _, guess = getattr(benchmarks, name)(radau, mesh=mesh, num_point=K)
    x0, lam, sigma = benchmarks.bench_inputs(system, guess)
    plan = system.plan
    out = []
    for layout in ("reference", "compact"):
        system.set_hessian_layout(layout)
        system.set_jacobian_layout(layout)
        t = []
        for it in range(160):
            x = x0 * (1.0 + 1e-7 * it)
            t0 = time.perf_counter()
            system.objective(x); system.gradient(x); system.constraints(x); system.jacobian(x); system.hessian(x, lam, sigma)
            t.append(time.perf_counter() - t0)
        out.append(float(np.median(t[40:])) * 1e6)
    jb = 8 * (plan.nnz_J + plan.nnz_H)
    print(f"{name:18s} {mesh:5d} x {K}: J + H reference layout {jb / 1e3:9.1f} KB   iterate reference {out[0]:8.1f} us   compact {out[1]:8.1f} us   "
          f"compact / reference {out[1] / out[0]:.2f}", flush=True)
    system._invalidate()
