#!/usr/bin/env python3
"""Developer helper: cycles/s of the single-launch cycle for several polynomial orders K at ~12k nodes
(K > 8: the pattern tables of a tile exceed the 64-entry LDS copies)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

dev = torch.device("cuda", 0)
for K in (4, 6, 8, 9, 10, 12, 16, 20):
    n_int = 12000 // K
    system, _, guess = models.planar_quadrotor(radau, mesh=n_int, num_point=K)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    outs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    args = (dx.data_ptr(), dlam.data_ptr(), sigma, *[o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    for _ in range(200):
        ev.cycle_dev(*args)
    ev.sync()
    t0 = time.perf_counter()
    n = 3000
    for _ in range(n):
        ev.cycle_dev(*args)
    ev.sync()
    dt = (time.perf_counter() - t0) / n
    mb = 8 * (plan.n * 2 + plan.m * 2 + 1 + plan.nnz_J + plan.nnz_H) / 1e6
    print(f"K={K:2d} intervals={n_int:5d} tiles={len(ev.tables.tiles):4d} ipw={ev.tables.intervals_per_wave}: "
          f"{1 / dt:9.0f} cycles/s  {dt * 1e6:7.2f} us  outputs {mb:6.1f} MB  {mb / dt / 1e6:6.2f} TB/s", flush=True)
    ev.close()
