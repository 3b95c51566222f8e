#!/usr/bin/env python3
"""Developer helper (GPU box): where polynomial order makes the per-interval products real matrix products -- the cycle
(pk_cycle) and the mesh error estimation (pk_err) on meshes whose intervals hold 16 ... 256 points, with the big-interval
products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, the default) and on the VALU (POCKIT_AMD_BIG_MFMA=0).
Run both:  python tools/big_k_sweep.py; POCKIT_AMD_BIG_MFMA=0 python tools/big_k_sweep.py
MODEL=quadrotor|humanoid|brachistochrone (default quadrotor); about NODES (default 12288) nodes per mesh."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

builder = {"quadrotor": models.planar_quadrotor, "humanoid": models.humanoid_wbc,
           "brachistochrone": models.brachistochrone}[os.environ.get("MODEL", "quadrotor")]
nodes = int(os.environ.get("NODES", "12288"))
dev = torch.device("cuda", 0)
print(f"big-interval products on the {'VALU' if os.environ.get('POCKIT_AMD_BIG_MFMA', '1') == '0' else 'fp64 matrix cores'}; "
      f"{os.environ.get('MODEL', 'quadrotor')}, ~{nodes} nodes", flush=True)
for K in (16, 32, 64, 65, 96, 128, 192, 256):
    n_int = max(nodes // K, 1)
    system, _, guess = builder(radau, mesh=n_int, num_point=K)
    plan, ev = system.plan, system.evaluator
    x, lam, sigma = models.bench_inputs(system, guess)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    outs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    args = (dx.data_ptr(), dlam.data_ptr(), sigma, *[o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    n = 300
    for _ in range(30):
        ev.cycle_dev(*args)
    ev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        ev.cycle_dev(*args)
    ev.sync()
    dt = (time.perf_counter() - t0) / n
    ev.mesh_error(x)
    eT = torch.zeros(ev._err_len, dtype=torch.float64, device=dev)
    eI = torch.zeros_like(eT)
    lib, h = ev.ctx.lib, ev.ctx.handle
    torch.cuda.synchronize()
    for _ in range(20):
        lib.pk_eval_mesh_error_dev(h, dx.data_ptr(), eT.data_ptr(), eI.data_ptr(), None)
    ev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        lib.pk_eval_mesh_error_dev(h, dx.data_ptr(), eT.data_ptr(), eI.data_ptr(), None)
    ev.sync()
    de = (time.perf_counter() - t0) / n
    mb = 8 * (plan.n * 2 + plan.m * 2 + 1 + plan.nnz_J + plan.nnz_H) / 1e6
    print(f"K={K:3d} intervals={n_int:4d}: cycle {dt * 1e6:8.2f} us ({mb:6.1f} MB of outputs, {mb / dt / 1e6:5.2f} TB/s)   "
          f"mesh error estimation {de * 1e6:8.2f} us", flush=True)
    ev.close()
