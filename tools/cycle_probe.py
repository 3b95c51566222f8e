#!/usr/bin/env python3
"""GPU box: the one-launch cycle of the benchmark configurations, device-resident -- us per launch and the fraction of the HBM
roof on the cycle's algorithmic bytes.  A light stand-in for bench.py in A/B loops over build switches (POCKIT_AMD_*).
Usage: cycle_probe.py [C2 C3 C4 C5 | name:intervals:points ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pockit_amd import benchmarks, radau  # noqa: E402

CONFIGS = {"C2": ("brachistochrone", 200, 8), "C3": ("planar_quadrotor", 2000, 6), "C4": ("two_stage_rocket", 1000, 4),
           "C5": ("humanoid_wbc", 5000, 8)}
dev = torch.device("cuda", 0)
for tag in (sys.argv[1:] or ["C3", "C5"]):
    name, mesh, K = CONFIGS[tag] if tag in CONFIGS else (tag.split(":")[0], int(tag.split(":")[1]), int(tag.split(":")[2]))
    system, _, guess = getattr(benchmarks, name)(radau, mesh=mesh, num_point=K)
    x, lam, sigma = benchmarks.bench_inputs(system, guess)
    plan, ev = system.plan, system.evaluator
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    outs = [torch.zeros(max(k, 1), dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    args = (dx.data_ptr(), dlam.data_ptr(), sigma, *[o.data_ptr() for o in outs])
    for _ in range(200):
        ev.cycle_dev(*args)
    ev.sync()
    best = None
    for _ in range(5):
        reps = 2000
        t0 = time.perf_counter()
        for _ in range(reps):
            ev.cycle_dev(*args)
        ev.sync()
        wall = (time.perf_counter() - t0) / reps * 1e6
        best = wall if best is None else min(best, wall)
    B = 8 * (5 * plan.n + plan.m + 1 + plan.n + plan.m + plan.nnz_J + plan.nnz_H)
    finite = all(bool(torch.isfinite(o).all()) for o in outs)
    subs = f" groups={ {cb: len(g) for (cb, k), g in ev.src.groups.items() if len(g) > 1} } subs={ev.src.cycle_subs}" if ev.src.grouped else ""
    print(f"{tag} {name} {mesh} x {K}:{subs} ipw={ev.tables.intervals_per_wave} tiles={len(ev.tables.tiles)}  {best:.2f} us/cycle  "
          f"{B / best / 1e6 / 8:.3f} of 8 TB/s  checksum J {float(outs[3].sum()):.9e} H {float(outs[4].sum()):.9e} finite={finite}", flush=True)
    system._invalidate()
