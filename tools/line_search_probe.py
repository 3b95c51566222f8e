"""GPU box: what a backtracking line search costs through the host shim.  One solver iteration here = `r` rejected trial
points (objective + constraints on a new x, nothing else) followed by an accepted point (all five callbacks on a new x), for
r = 0, 1, 2, 4 -- with the adaptive prefetch of grad f / J (pk_set_host_option "adaptive_prefetch") and without it.
usage: CASE=quadrotor|humanoid|rocket python3 tools/line_search_probe.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, ".")
from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

CASES = {"quadrotor": (models.planar_quadrotor, 2000, 6), "humanoid": (models.humanoid_wbc, 5000, 8), "rocket": (models.two_stage_rocket, 1000, 4)}
case = os.environ.get("CASE", "quadrotor")
builder, mesh, K = CASES[case]
system, _, guess = builder(radau, mesh, K)
x, lam, sigma = models.bench_inputs(system, guess)
xs = [x * (1 + 1e-9 * k) for k in range(16)]
ev = system.evaluator
lib, h = ev.ctx.lib, ev.ctx.handle
print(f"{case} {mesh}x{K}: n = {system.plan.n}")
for adaptive in (1, 0):
    ev.ctx.check(lib.pk_set_host_option(h, b"adaptive_prefetch", adaptive))
    for r in (0, 1, 2, 4):
        k, rows = 0, []
        for it in range(10 + 60):
            t0 = time.perf_counter()
            for _ in range(r):
                xk = xs[k % 16]; k += 1
                system.objective(xk)
                system.constraints(xk)
            xk = xs[k % 16]; k += 1
            system.objective(xk); system.gradient(xk); system.constraints(xk); system.jacobian(xk); system.hessian(xk, lam, sigma)
            if it >= 10:
                rows.append(time.perf_counter() - t0)
        print(f"   adaptive_prefetch = {adaptive}: {r} rejected trial points + 1 accepted point: {statistics.median(rows) * 1e6:8.1f} us per iteration", flush=True)
ev.ctx.check(lib.pk_set_host_option(h, b"adaptive_prefetch", 1))
