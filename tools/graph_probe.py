#!/usr/bin/env python3
"""Developer helper: device-resident cycle time with plain stream launches vs the cached hipGraph replay."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

for name, builder, kw in (("quadrotor 2000x6", models.planar_quadrotor, dict(mesh=2000, num_point=6)),
                          ("brachistochrone 1250x8", models.brachistochrone, dict(mesh=1250, num_point=8)),
                          ("humanoid 5000x8", models.humanoid_wbc, dict(mesh=5000, num_point=8))):
    system, _, guess = builder(radau, **kw)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    lib, h = ev.ctx.lib, ev.ctx.handle
    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    o = {k: torch.zeros(n, dtype=torch.float64, device=dev) for k, n in
         (("f", 1), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H))}
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for stream in (None, torch.cuda.current_stream().cuda_stream):
        args = (h, ptr(dx), ptr(dlam), C.c_double(float(sigma)), ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]), ptr(o["J"]),
                ptr(o["H"]), C.c_void_p(stream))
        ref = None
        for graph in (0, 1, 0, 1):
            ev.ctx.check(lib.pk_set_cycle_graph(h, graph))
            for _ in range(30):
                ev.ctx.check(lib.pk_eval_cycle_dev(*args))
            ev.sync(C.c_void_p(stream)); torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 500
            for _ in range(n):
                lib.pk_eval_cycle_dev(*args)
            ev.sync(C.c_void_p(stream)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            H = o["H"].cpu().numpy().copy()
            if ref is None:
                ref = H
            print(f"{name:24s} stream={'ctx' if stream is None else 'torch'} graph={graph} {dt*1e6:8.2f} us/cycle "
                  f"{1/dt:9.0f} cycles/s  same={np.array_equal(H, ref)} f={float(o['f'][0]):.6g}", flush=True)
    system._invalidate()
