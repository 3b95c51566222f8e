"""GPU box: the host-landed cycle (five callbacks, NumPy in / out, new x per iterate) of the planar quadrotor from 60 to 96 000
nodes: microseconds per iterate, bytes over PCIe, the rate they correspond to, and which regime of the host shim carried it
(small_direct / copy kernels / DMA for the large pieces / helper threads).
usage: python3 tools/host_size_sweep.py"""
import statistics
import sys
import time

sys.path.insert(0, ".")
from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

print(f"{'intervals':>9s} {'nodes':>7s} {'x KB':>8s} {'down MB':>8s} {'five callbacks us':>18s} {'one call us':>12s} {'GB/s':>6s}  regime")
for mesh in (10, 30, 100, 300, 1000, 2000, 4000, 8000, 16000):
    system, _, guess = models.planar_quadrotor(radau, mesh, 6)
    x, lam, sigma = models.bench_inputs(system, guess)
    xs = [x * (1 + 1e-9 * k) for k in range(4)]
    ev = system.evaluator
    reps = 200 if mesh <= 2000 else 40
    rows, one = [], []
    for k in range(20 + reps):
        xk = xs[k % 4]
        t0 = time.perf_counter()
        system.objective(xk); system.gradient(xk); system.constraints(xk); system.jacobian(xk); system.hessian(xk, lam, sigma)
        if k >= 20:
            rows.append(time.perf_counter() - t0)
    for k in range(20 + reps):
        t0 = time.perf_counter()
        ev.cycle(xs[k % 4], lam, sigma)
        if k >= 20:
            one.append(time.perf_counter() - t0)
    p = system.plan
    kept = sum(b - a for a, b in ev.jac_constant_runs)
    down = 8 * (p.n + p.m + p.nnz_J - kept + p.nnz_H)
    us = statistics.median(rows) * 1e6
    regime = ("kernels read x in place, store into the landing block" if 8 * p.n <= 128 << 10 and 8 * (p.nnz_J + p.n + p.m) <= 1 << 20
              else "copy kernels" if 8 * (p.nnz_J - kept + p.n + p.m) <= 8 << 20 else "DMA for the large pieces")
    if ev.host_helper_threads:
        regime += f", {ev.host_helper_threads} helper threads"
    print(f"{mesh:9d} {6 * mesh:7d} {8 * p.n / 1024:8.1f} {down / 1e6:8.2f} {us:18.1f} {statistics.median(one) * 1e6:12.1f} "
          f"{(down + 16 * p.n) / us / 1e3:6.1f}  {regime}", flush=True)
    system._invalidate()
