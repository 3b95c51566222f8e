#!/usr/bin/env python3
"""Developer helper: time the host-buffer (PCIe-inclusive) NLP-callback cycle of the product, i.e. what
the cyipopt shim pays per iteration: objective, gradient, constraints, jacobian, hessian with NumPy arrays --
in every host mode of the evaluator (prefetch / on demand, DMA / kernels storing into host memory, zero-copy views /
caller-owned arrays), plus the one-call cycle and the per-callback one-shot path."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

CASES = (("quadrotor 2000x6", models.planar_quadrotor, dict(mesh=2000, num_point=6)),
         ("brachistochrone 1250x8", models.brachistochrone, dict(mesh=1250, num_point=8)),
         ("humanoid 5000x8", models.humanoid_wbc, dict(mesh=5000, num_point=8)))
only = os.environ.get("CASE")

for name, builder, kw in CASES:
    if only and only not in name:
        continue
    system, _, guess = builder(radau, **kw)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    xs = [x * (1 + 1e-9 * k) for k in range(8)]
    mb = 8 * (system.plan.n * 2 + system.plan.m * 2 + 1 + system.plan.nnz_J + system.plan.nnz_H) / 1e6

    def cycle(xk):
        t = [time.perf_counter()]
        for fn in (system.objective, system.gradient, system.constraints, system.jacobian):
            fn(xk)
            t.append(time.perf_counter())
        system.hessian(xk, lam, sigma)
        t.append(time.perf_counter())
        return [t[i + 1] - t[i] for i in range(5)]

    def cycle_direct(xk):
        t0 = time.perf_counter()
        ev.objective_direct(xk); ev.gradient_direct(xk); ev.constraints_direct(xk); ev.jacobian_direct(xk)
        ev.hessian_direct(xk, lam, sigma)
        return [time.perf_counter() - t0, 0, 0, 0, 0]

    def cycle_one_call(xk):
        t0 = time.perf_counter()
        ev.cycle(xk, lam, sigma)
        return [time.perf_counter() - t0, 0, 0, 0, 0]

    rows = []
    for prefetch in (True, False):
        for direct in (False, True):
            for zc in (True, False):
                rows.append((f"prefetch={int(prefetch)} host_direct={int(direct)} zero_copy={int(zc)}", cycle, prefetch, direct, zc))
    rows.append(("one call (pk_cycle), caller-owned pinned arrays", cycle_one_call, True, False, False))
    rows.append(("one upload + launch per callback", cycle_direct, True, False, False))
    for label, fn, prefetch, direct, zc in rows:
        ev.set_host_mode(prefetch, direct)
        ev.zero_copy = zc
        for k in range(10):
            fn(xs[k % 8])
        n = 100
        per = [fn(xs[k % 8]) for k in range(n)]
        tot = statistics.median(sum(p) for p in per)
        cb = [statistics.median(p[i] for p in per) * 1e6 for i in range(5)]
        print(f"{name:24s} {label:52s} {tot*1e3:8.3f} ms/cycle {1/tot:9.1f} cycles/s  {mb/tot/1e3:6.1f} GB/s   "
              f"f/grad/g/J/H us: " + " ".join(f"{c:7.1f}" for c in cb), flush=True)
    ev.set_host_mode(True, False)
    ev.zero_copy = False
    system._invalidate()
