"""GPU box: soak of the host path (progress-mark waits, early grad f / g, speculative Hessian, direct Hessian stores, one-launch
x-part, one-call cycle): many iterates of the five callbacks and of Evaluator.cycle over a handful of inputs, EVERY array of
EVERY iterate compared bit for bit with what the same callbacks gave for the same input under the conservative host options
(event waits instead of polling and marks, one joined copy, no speculative launch, no direct stores: the same kernels, so the
values must agree to the last bit).  A result that was read before it had landed shows up as a mismatch.
usage: python3 tools/host_soak.py [iterates per case]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
CASES = [("quadrotor LGR 2000x6", models.planar_quadrotor, (2000, 6), N), ("rocket LGR 2x1000x4", models.two_stage_rocket, (1000, 4), N),
         ("brachistochrone LGR 20x8", models.brachistochrone, (20, 8), 2 * N), ("humanoid LGR 5000x8", models.humanoid_wbc, (5000, 8), max(N // 20, 50)),
         # (pass-parallel x-part and Hessian callback: the humanoid on a mesh that underfills the chip)
         ("humanoid LGR 100x8", models.humanoid_wbc, (100, 8), N)]
total_bad = 0
for label, builder, args, iters in CASES:
    system, _, guess = builder(radau, *args)
    x, lam, sigma = models.bench_inputs(system, guess)
    ev = system.evaluator
    rng = np.random.default_rng(3)
    inputs, want = [], []
    lib, h = ev.ctx.lib, ev.ctx.handle
    careful = {"spin_wait": 0, "mark_wait": 0, "split_copy": 0, "speculative_hess": 0, "hess_direct": 0, "kernel_download": 0}
    for name, v in careful.items():
        ev.ctx.check(lib.pk_set_host_option(h, name.encode(), v))
    for k in range(6):
        xk = x * (1 + 1e-4 * rng.standard_normal(x.size))
        lk = lam * (1 + 1e-2 * rng.standard_normal(lam.size))
        sk = float(sigma * (1 + 0.1 * k))
        inputs.append((xk, lk, sk))
        want.append((float(system.objective(xk)), np.array(system.gradient(xk)), np.array(system.constraints(xk)),
                     np.array(system.jacobian(xk)), np.array(system.hessian(xk, lk, sk))))
    ev.sync()
    for name, v in {"spin_wait": 1, "mark_wait": 1, "split_copy": 1, "speculative_hess": 1, "hess_direct": 1, "kernel_download": 8}.items():
        ev.ctx.check(lib.pk_set_host_option(h, name.encode(), v))
    bad, t0 = 0, time.perf_counter()
    for it in range(iters):
        k = int(rng.integers(6))
        xk, lk, sk = inputs[k]
        wf, wg, wc, wj, wh = want[k]
        mode = it % 7
        if it % 11 == 5:                                # a rejected trial point of a line search: f and g only
            ok = float(system.objective(xk)) == wf and np.array_equal(system.constraints(xk), wc)
            bad += 0 if ok else 1
            continue
        if mode == 6:                                   # all five from one call
            f, grad, g, J, H = ev.cycle(xk, lk, sk)
        else:
            order = [0, 1, 2, 3] if mode < 3 else list(rng.permutation(4))
            res = {}
            if mode == 5:                               # the Hessian callback sees the iterate first
                res[4] = system.hessian(xk, lk, sk)
            for w in order:
                res[w] = (system.objective, system.gradient, system.constraints, system.jacobian)[w](xk)
            if 4 not in res:
                res[4] = system.hessian(xk, lk, sk)
            f, grad, g, J, H = (res[w] for w in range(5))
        ok = (float(f) == wf and np.array_equal(grad, wg) and np.array_equal(g, wc) and np.array_equal(J, wj) and np.array_equal(H, wh))
        if not ok:
            bad += 1
            if bad <= 5:
                print(f"   MISMATCH {label} iterate {it} mode {mode}: f {float(f) == wf} grad {np.array_equal(grad, wg)} g {np.array_equal(g, wc)} "
                      f"J {np.array_equal(J, wj)} H {np.array_equal(H, wh)}", flush=True)
        del f, grad, g, J, H
    total_bad += bad
    print(f"{label:28s} {iters:6d} iterates, {bad} with a mismatch, {time.perf_counter() - t0:.1f} s", flush=True)
    # the sharded host path with one rank (progress marks stored by the GPU, early grad f / g, speculative Hessian): same kernels
    # for grad f, g, J, H (f is finished on the host there: compared to 1e-13)
    from pockit_amd.hostshard import HostShardedEvaluator

    hs = HostShardedEvaluator(system.plan, 0, 1, None, device=0)
    bad, t0 = 0, time.perf_counter()
    for it in range(iters // 2):
        k = int(rng.integers(6))
        xk, lk, sk = inputs[k]
        wf, wg, wc, wj, wh = want[k]
        res = {}
        order = [0, 1, 2, 3] if it % 3 else list(rng.permutation(4))
        if it % 5 == 4:
            res[4] = hs.hessian(xk, lk, sk)
        for w in order:
            res[w] = (hs.objective, hs.gradient, hs.constraints, hs.jacobian)[w](xk)
        if 4 not in res:
            res[4] = hs.hessian(xk, lk, sk)
        ok = (abs(float(res[0]) - wf) <= 1e-13 * max(1.0, abs(wf)) and np.array_equal(res[1], wg) and np.array_equal(res[2], wc)
              and np.array_equal(res[3], wj) and np.array_equal(res[4], wh))
        if not ok:
            bad += 1
            if bad <= 5:
                print(f"   MISMATCH (sharded path) {label} iterate {it}: f {float(res[0]) - wf:.3e} grad {np.array_equal(res[1], wg)} "
                      f"g {np.array_equal(res[2], wc)} J {np.array_equal(res[3], wj)} H {np.array_equal(res[4], wh)}", flush=True)
    hs.close()
    total_bad += bad
    print(f"{label:28s} {iters // 2:6d} iterates through the sharded host path (1 rank), {bad} with a mismatch, {time.perf_counter() - t0:.1f} s",
          flush=True)
print("host soak:", "all iterates bit-identical to the synchronous API" if total_bad == 0 else f"{total_bad} MISMATCHES")
sys.exit(1 if total_bad else 0)
