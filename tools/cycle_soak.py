#!/usr/bin/env python3
"""Developer helper (GPU box): repeat the single-launch cycle on fixed inputs for several models and sizes and compare every
launch BIT FOR BIT with the first one and with the stand-alone kernels (pk_hess for H; the fused x-kernel pk_xall of the
host shim for grad f, g, J).  A wave-level hazard or race shows up as a launch that differs (round 2 found one this
way: the 16-byte streaming store needed a wait state)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.lobatto as lobatto  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

REPS = int(os.environ.get("REPS", "200"))
CASES = [("humanoid 5000x8", models.humanoid_wbc, radau, dict(mesh=5000, num_point=8)),
         ("humanoid 2000x12", models.humanoid_wbc, radau, dict(mesh=2000, num_point=12)),
         ("quadrotor 60000x6", models.planar_quadrotor, radau, dict(mesh=60000, num_point=6)),
         ("quadrotor LGL 20000x7", models.planar_quadrotor, lobatto, dict(mesh=20000, num_point=7)),
         ("rocket 2x20000x4", models.two_stage_rocket, radau, dict(mesh=20000, num_point=4)),
         ("brachistochrone 30000x8", models.brachistochrone, radau, dict(mesh=30000, num_point=8)),
         ("quadrotor 2000x6", models.planar_quadrotor, radau, dict(mesh=2000, num_point=6))]
total_bad = 0
for name, builder, ns, kw in CASES:
    t0 = time.perf_counter()
    system, _, guess = builder(ns, **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    H0 = np.array(ev.hessian_direct(x, lam, sigma))
    g0, grad0, J0 = np.array(system.constraints(x)), np.array(system.gradient(x)), np.array(system.jacobian(x))
    f0 = float(system.objective(x))
    bad = {"f": 0, "grad": 0, "g": 0, "J": 0, "H": 0}
    for rep in range(REPS):
        f1, grad1, g1, J1, H1 = ev.cycle(x, lam, sigma)
        bad["f"] += int(float(f1) != f0)
        bad["grad"] += int(not np.array_equal(grad1, grad0))
        bad["g"] += int(not np.array_equal(g1, g0))
        bad["J"] += int(not np.array_equal(J1, J0))
        bad["H"] += int(not np.array_equal(H1, H0))
    total_bad += sum(bad.values())
    print(f"{name:26s} nodes {sum(pp.layout.L_m for pp in system.plan.phase_plans):7d}  {REPS} launches, launches differing from "
          f"the stand-alone kernels: {bad}   ({time.perf_counter() - t0:.0f} s)", flush=True)
    system._invalidate()
print("TOTAL differing launches:", total_bad)
sys.exit(1 if total_bad else 0)
