#!/usr/bin/env python3
"""GPU box: the one-launch cycle of the reference's three LARGE example models (derivative set evaluated in groups, DESIGN.md
section 3c) re-meshed to benchmark size -- cycles/s, us per launch and the fraction of the HBM roof on the cycle's algorithmic
bytes, next to the humanoid (single pass) at the same mesh.  Models come from tests/golden/examples/*.model.json (the model
the example program configured, as data); x = 0.6 ... 1.4 seeded, lambda ~ N(0, 1).
Usage: fat_model_probe.py [intervals] [num_point]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import model_io  # noqa: E402

intervals = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
for name in ("humanoid_whole_body_control", "drone_stabilization", "rocket_powered_descent", "orbit_transfer"):
    with open(os.path.join(ROOT, "tests", "golden", "examples", name + ".model.json")) as fh:
        desc = json.load(fh)
    for pd in desc["phases"]:
        pd["mesh"] = [float(v) for v in np.linspace(0.0, 1.0, intervals + 1)]
        pd["num_point"] = [K] * intervals
    t0 = time.time()
    system = model_io.load_system(desc)
    plan, ev = system.plan, system.evaluator
    rng = np.random.default_rng(7)
    x, lam = rng.uniform(0.6, 1.4, size=plan.n), rng.standard_normal(plan.m)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    outs = [torch.zeros(max(k, 1), dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    setup = time.time() - t0
    args = (dx.data_ptr(), dlam.data_ptr(), 0.7, *[o.data_ptr() for o in outs])
    for _ in range(50):
        ev.cycle_dev(*args)
    ev.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 300
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        ev.cycle_dev(*args)
    ev.sync()
    wall = (time.perf_counter() - t0) / reps * 1e6
    B = 8 * (5 * plan.n + plan.m + 1 + plan.n + plan.m + plan.nnz_J + plan.nnz_H)
    groups = {f"{cb}": len(g) for (cb, k), g in ev.src.groups.items() if len(g) > 1}
    finite = all(bool(torch.isfinite(o).all()) for o in outs)
    print(f"{name:30s} {intervals} x {K}: n={plan.n} nnz_J={plan.nnz_J} nnz_H={plan.nnz_H} B={B / 1e6:.1f} MB  groups={groups or 'single pass'}  "
          f"ipw={ev.tables.intervals_per_wave} tiles={len(ev.tables.tiles)} subs={ev.src.cycle_subs}  "
          f"{wall:.1f} us/cycle  {B / wall / 1e6:.2f} TB/s = {B / wall / 1e6 / 8:.3f} of 8 TB/s  finite={finite}  setup {setup:.1f} s",
          flush=True)
    system._invalidate()
