#!/usr/bin/env python3
"""Developer helper: soak test of pk_cycle's in-launch hand-off.  N cycles are queued back to back on alternating
iterates, every launch writes f and grad f to its own slot; afterwards every slot must equal, bit for bit, the value
of its iterate (a missed, stale or torn partial sum would show up in f or in the shared gradient slots).
usage: handoff_soak.py [launches] [workload] [intervals]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
name = sys.argv[2] if len(sys.argv) > 2 else "two_stage_rocket"
intervals = int(sys.argv[3]) if len(sys.argv) > 3 else 200
system, _, guess = bench.build_workload(name, intervals, radau)
plan, ev = system.plan, system.evaluator
xa, lam, sigma = models.bench_inputs(system, guess)
xb = xa * (1.0 + 0.05 * np.random.default_rng(11).uniform(-1, 1, xa.shape))
dev = torch.device("cuda", 0)
dxs = [torch.from_numpy(v).to(dev) for v in (xa, xb)]
dlam = torch.from_numpy(lam).to(dev)
chunk = 2000
f = torch.empty(chunk, dtype=torch.float64, device=dev)
grad = torch.empty((chunk, plan.n), dtype=torch.float64, device=dev)
g, J, H = (torch.zeros(n, dtype=torch.float64, device=dev) for n in (plan.m, plan.nnz_J, plan.nnz_H))
ref = None
bad = 0
for start in range(0, total, chunk):
    f.fill_(float("nan"))
    grad.fill_(float("nan"))
    torch.cuda.synchronize()
    for i in range(chunk):
        ev.cycle_dev(dxs[i % 2].data_ptr(), dlam.data_ptr(), sigma, f[i:].data_ptr(), grad[i].data_ptr(), g.data_ptr(),
                     J.data_ptr(), H.data_ptr())
    ev.sync()
    if ref is None:
        ref = (f[:2].clone(), grad[:2].clone())
    for par in (0, 1):
        bad += int((f[par::2] != ref[0][par]).sum()) + int((grad[par::2] != ref[1][par]).any(dim=1).sum())
print(f"{name} {intervals} intervals: {total} launches, {bad} slots differ from their iterate's value")
sys.exit(1 if bad else 0)
