#!/usr/bin/env python3
"""GPU box: the one-launch cycle of the reference's three LARGE example models (derivative set evaluated in groups, DESIGN.md
section 3c) re-meshed to benchmark size -- cycles/s, us per launch and the fraction of the HBM roof on the cycle's algorithmic
bytes, next to the humanoid (single pass) at the same mesh.  Models come from tests/golden/examples/*.model.json (the model
the example program configured, as data); x = 0.6 ... 1.4 seeded, lambda ~ N(0, 1).
Beside it the same cycle on the compact layouts (pk_cyclec).  Usage: fat_model_probe.py [intervals] [num_point] [--chain]
(--chain: the wide synthetic models benchmarks.state_chain with 52 / 80 / 128 states instead)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import model_io  # noqa: E402

args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
intervals = int(args_[0]) if len(args_) > 0 else 2000
K = int(args_[1]) if len(args_) > 1 else 4
dev = torch.device("cuda", 0)


def timed(ev, plan, dx, dlam, nj, nh, reps=300):
    outs = [torch.zeros(max(k, 1), dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, nj, nh)]
    args = (dx.data_ptr(), dlam.data_ptr(), 0.7, *[o.data_ptr() for o in outs])
    for _ in range(50):
        ev.cycle_dev(*args)
    ev.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ev.cycle_dev(*args)
    ev.sync()
    wall = (time.perf_counter() - t0) / reps * 1e6
    return wall, all(bool(torch.isfinite(o).all()) for o in outs)


def probe(name, system, setup):
    plan, ev = system.plan, system.evaluator
    rng = np.random.default_rng(7)
    x, lam = rng.uniform(0.6, 1.4, size=plan.n), rng.standard_normal(plan.m)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    wall, finite = timed(ev, plan, dx, dlam, plan.nnz_J, plan.nnz_H)
    B = 8 * (5 * plan.n + plan.m + 1 + plan.n + plan.m + plan.nnz_J + plan.nnz_H)
    groups = {f"{cb}": len(g) for (cb, k), g in ev.src.groups.items() if len(g) > 1}
    text = (f"{name:30s} {intervals} x {K}: n={plan.n} nnz_J={plan.nnz_J} nnz_H={plan.nnz_H} B={B / 1e6:.1f} MB  groups={groups or 'single pass'}  "
            f"cap={ev.src.group_cap} ipw={ev.tables.intervals_per_wave} tiles={len(ev.tables.tiles)} subs={ev.src.cycle_subs}  "
            f"{wall:.1f} us/cycle  {B / wall / 1e6:.2f} TB/s = {B / wall / 1e6 / 8:.3f} of 8 TB/s  finite={finite}  setup {setup:.1f} s")
    if ev.src.compact:                   # the compact layouts from the same single launch (pk_cyclec)
        plan.jacc  # noqa: B018
        ev.set_cycle_layout(True, True)
        try:
            wc, fc = timed(ev, plan, dx, dlam, plan.nnz_Jc, plan.nnz_Hc)
        finally:
            ev.set_cycle_layout(False, False)
        Bc = 8 * (5 * plan.n + plan.m + 1 + plan.n + plan.m + plan.nnz_Jc + plan.nnz_Hc)
        text += (f"  | compact layouts: nnz_Jc={plan.nnz_Jc} nnz_Hc={plan.nnz_Hc} {wc:.1f} us/cycle ({wc / wall:.2f} x the reference "
                 f"layouts' time) {Bc / wc / 1e6 / 8:.3f} of 8 TB/s on {Bc / 1e6:.1f} MB finite={fc}")
    print(text, flush=True)
    system._invalidate()


if "--chain" in sys.argv:               # the wide synthetic models of round 5 (benchmarks.state_chain)
    from pockit_amd import benchmarks
    import pockit_amd.radau as radau

    for n_states in (52, 80, 128):
        t0 = time.time()
        system, _, _ = benchmarks.state_chain(radau, states=n_states, mesh=intervals, num_point=K)
        system.evaluator  # noqa: B018
        probe(f"state_chain {n_states} states", system, time.time() - t0)
    sys.exit(0)
only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
for name in only or ("humanoid_whole_body_control", "drone_stabilization", "rocket_powered_descent", "orbit_transfer"):
    with open(os.path.join(ROOT, "tests", "golden", "examples", name + ".model.json")) as fh:
        desc = json.load(fh)
    for pd in desc["phases"]:
        pd["mesh"] = [float(v) for v in np.linspace(0.0, 1.0, intervals + 1)]
        pd["num_point"] = [K] * intervals
    t0 = time.time()
    system = model_io.load_system(desc)
    system.evaluator  # noqa: B018
    probe(name, system, time.time() - t0)
