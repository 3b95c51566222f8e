#!/usr/bin/env python3
"""Developer helper (GPU box): batches of pk_cycle replayed from ONE hipGraph (pk_set_cycle_graph + pk_eval_cycle_dev_repeat,
the form bench.py may choose for its timed batches) -- after every replay of `count` kernel nodes all five outputs must equal
a single cycle's bit for bit, with x changing between the replays (the graph holds pointers, not values).
Usage: graph_soak.py [workload] [intervals] [count] [replays]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "planar_quadrotor"
intervals = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
replays = int(sys.argv[4]) if len(sys.argv) > 4 else 300
system, _, guess = bench.build_workload(name, intervals, radau)
x, lam, sigma = models.bench_inputs(system, guess)
ev, plan = system.evaluator, system.plan
dev = torch.device("cuda", 0)
xs = [x * (1 + 1e-7 * k) for k in range(4)]
want = [[np.asarray(v).ravel().copy() for v in ev.cycle(xk, lam, sigma)] for xk in xs]
dx, dlam = torch.from_numpy(xs[0].copy()).to(dev), torch.from_numpy(lam).to(dev)
sizes = (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))
o = {k: torch.zeros(n, dtype=torch.float64, device=dev) for k, n in sizes}
lib, h = ev.ctx.lib, ev.ctx.handle
args = (h, C.c_void_p(dx.data_ptr()), C.c_void_p(dlam.data_ptr()), C.c_double(float(sigma)),
        *[C.c_void_p(o[k].data_ptr()) for k, _ in sizes], None, count, 0, None)
ev.set_cycle_graph(True)
bad = 0
for r in range(replays):
    k = r % 4
    dx.copy_(torch.from_numpy(xs[k]))
    torch.cuda.synchronize()
    ev.ctx.check(lib.pk_eval_cycle_dev_repeat(*args))
    ev.sync()
    for (key, _), w in zip(sizes, want[k]):
        if not np.array_equal(o[key].cpu().numpy(), w):
            bad += 1
            print(f"replay {r}: {key} differs")
ev.set_cycle_graph(False)
print(f"{name} {intervals} intervals: {replays} replays of a graph of {count} cycles = {replays * count} launches, {bad} outputs differ")
sys.exit(1 if bad else 0)
