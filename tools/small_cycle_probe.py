"""GPU box: the host-landed cycle (five callbacks, NumPy in / out) on SMALL systems, where fixed costs are everything.
usage: python3 tools/small_cycle_probe.py"""
import statistics
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.lobatto as lobatto  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

CASES = [("lqr LGL 10x10", models.lqr, lobatto, (10, 10)), ("brachistochrone LGR 20x8", models.brachistochrone, radau, (20, 8)),
         ("brachistochrone LGR 200x8", models.brachistochrone, radau, (200, 8)), ("quadrotor LGR 100x6", models.planar_quadrotor, radau, (100, 6)),
         ("quadrotor LGR 500x6", models.planar_quadrotor, radau, (500, 6))]
med = lambda v: statistics.median(v) * 1e6  # noqa: E731
for label, builder, ns, args in CASES:
    system, _, guess = builder(ns, *args)
    x, lam, sigma = models.bench_inputs(system, guess)
    xs = [x * (1 + 1e-9 * k) for k in range(4)]
    rows = []
    for k in range(40 + 200):
        xk = xs[k % 4]
        t = [time.perf_counter()]
        system.objective(xk); t.append(time.perf_counter())
        system.gradient(xk); t.append(time.perf_counter())
        system.constraints(xk); t.append(time.perf_counter())
        system.jacobian(xk); t.append(time.perf_counter())
        system.hessian(xk, lam, sigma); t.append(time.perf_counter())
        if k >= 40:
            rows.append([t[i + 1] - t[i] for i in range(5)] + [t[5] - t[0]])
    m = [med([r[i] for r in rows]) for i in range(6)]
    ev = system.evaluator
    one = []
    for k in range(40 + 200):
        t0 = time.perf_counter()
        ev.cycle(xs[k % 4], lam, sigma)
        if k >= 40:
            one.append(time.perf_counter() - t0)
    p = system.plan
    print(f"{label:28s} n={p.n:6d} nnz_J={p.nnz_J:7d} nnz_H={p.nnz_H:7d}  f/grad/g/J/H {m[0]:6.1f} {m[1]:6.1f} {m[2]:6.1f} {m[3]:6.1f} {m[4]:6.1f} "
          f"| five callbacks {m[5]:7.1f} us | one call {med(one):7.1f} us", flush=True)
