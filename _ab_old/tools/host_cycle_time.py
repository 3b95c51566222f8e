#!/usr/bin/env python3
"""Developer helper: time the host-buffer (PCIe-inclusive) NLP-callback cycle of the product, i.e. what
the cyipopt shim pays per iteration: objective, gradient, constraints, jacobian, hessian with NumPy arrays."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

for name, builder, kw in (("quadrotor 2000x6", models.planar_quadrotor, dict(mesh=2000, num_point=6)),
                          ("brachistochrone 1250x8", models.brachistochrone, dict(mesh=1250, num_point=8)),
                          ("humanoid 5000x8", models.humanoid_wbc, dict(mesh=5000, num_point=8))):
    system, _, guess = builder(radau, **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    xs = [x * (1 + 1e-9 * k) for k in range(8)]

    def cycle(xk):
        system.objective(xk); system.gradient(xk); system.constraints(xk); system.jacobian(xk)
        system.hessian(xk, lam, sigma)

    def cycle_direct(xk):
        ev.objective_direct(xk); ev.gradient_direct(xk); ev.constraints_direct(xk); ev.jacobian_direct(xk)
        ev.hessian_direct(xk, lam, sigma)

    def cycle_zero_copy(xk):
        ev.zero_copy = True
        cycle(xk)
        ev.zero_copy = False

    for fn, label in ((cycle_zero_copy, "x-cached fused shim, pinned views"), (cycle, "x-cached fused shim, fresh copies"),
                      (cycle_direct, "one upload + launch per callback")):
        for k in range(5):
            fn(xs[k % 8])
        t0 = time.perf_counter()
        n = 60
        for k in range(n):
            fn(xs[k % 8])
        dt = (time.perf_counter() - t0) / n
        mb = 8 * (system.plan.n * 2 + system.plan.m * 2 + 1 + system.plan.nnz_J + system.plan.nnz_H) / 1e6
        print(f"{name:24s} {label:34s} {dt*1e3:8.3f} ms/cycle  {1/dt:9.1f} cycles/s   D2H+H2D {mb:.1f} MB/cycle")
