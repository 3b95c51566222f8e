"""Workloads of the benchmark: BASELINE.json's configurations re-meshed on LGR, their seeded inputs, the algorithmic
bytes of a cycle and the CPU baseline (the oracle timed on the host).  No GPU code here."""
import os
import time

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBPS = 6300.0   # what the guide's stream-type kernels reach (MI355X_MICROARCH.md: "~6.3 TB/s achievable")
LAUNCH_FLOOR_US = 4.3          # a pk_cycle-shaped launch before its first output byte leaves
KERNEL_IDS = {"pk_int": 0, "pk_fin": 1, "pk_g": 2, "pk_grad": 3, "pk_jac": 4, "pk_hess": 5, "pk_xall": 6, "pk_cycle": 12}
PARITY_TOL = 1e-11             # |a - b| <= tol * max(1, max|b|) per array


def side_roofline(nbytes, us):
    """Algorithmic bytes of one launch of a side kernel over its per-dispatch time, against the HBM roof."""
    gbps = nbytes / (us * 1e-6) / 1e9 if us else None
    return {"bound": "hbm", "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": us, "achieved": gbps, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": (gbps / HBM_PEAK_GBPS if gbps else None),
            "frac_of_achievable": (gbps / HBM_ACHIEVABLE_GBPS if gbps else None)}


MIN_REGION_S = 0.05        # the timed region lasts at least this long (R batches of `steps` cycles)
EVENT_SPACING = 200        # cycles between two timing events of the region (an event costs ~3 us of GPU time)
MIN_WARMUP = 500           # untimed launches before the timed region (single GPU), whatever --warmup says
ISOLATED_SAMPLES = 200     # per-dispatch kernel timings on an idle stream (what a profiler's kernel trace measures)
POINTS = {"planar_quadrotor": 6, "brachistochrone": 8, "two_stage_rocket": 4, "humanoid_wbc": 8, "three_stage_rocket": 4,
          "humanoid_team": 8}


def algorithmic_bytes(plan):
    """Per-kernel algorithmic traffic of one cycle:
    B = 8 (5n + m + 1 + n + m + nnz_J + nnz_H): x read by each callback, lambda once, outputs once."""
    n, m = plan.n, plan.m
    per = {
        "f": 8 * (n + 1),
        "grad": 8 * (n + n),
        "g": 8 * (n + m),
        "jac": 8 * (n + plan.nnz_J),
        "hess": 8 * (n + m + plan.nnz_H),
    }
    per["cycle"] = sum(per.values())
    per["xall"] = 8 * (n + 1 + n + m + plan.nnz_J)      # fused x-kernel: x read once, f partials + grad + g + J written
    per["cycle_x_once"] = per["cycle"] - 8 * 4 * n       # the single-launch cycle reads x once from HBM
    return per


def build_workload(name, intervals, ns):
    from pockit_amd import benchmarks as models

    if name.endswith("_lgl"):                       # Lobatto variant of a workload (side line of the bench)
        import pockit_amd.lobatto as lobatto

        return build_workload(name[:-4], intervals, lobatto)
    return getattr(models, name)(ns, intervals, POINTS[name])


def cpu_baseline(name, intervals, budget_s=12.0, max_cycles=5000, gpu_outputs=None):
    """The oracle (NumPy restatement of the reference algorithm, single thread) on the same workload: the same five
    callbacks with host arrays in and out, a bounded sample (``budget_s`` seconds of cycles).  ``gpu_outputs`` = (f, grad,
    g, J, H) of the GPU path on the same inputs: compared with the oracle's here (the line's ``parity``)."""
    from pockit_amd import benchmarks as models
    import numpy as np
    import oracle.radau

    system, _, guess = build_workload(name, intervals, oracle.radau)
    x, lam, sigma = models.bench_inputs(system, guess)

    def cycle():
        return (system.objective(x), system.gradient(x), system.constraints(x), system.jacobian(x), system.hessian(x, lam, sigma))

    for _ in range(2):
        ref = cycle()
    parity = None
    if gpu_outputs is not None:
        errs = {}
        for nm, a, b in zip(("f", "grad", "g", "J", "H"), gpu_outputs, ref):
            a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
            errs[nm] = float("inf") if a.shape != b.shape else (float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0)
        worst = max(errs.values())
        parity = {"vs": "oracle (CPU restatement of the reference), every entry of f, grad f, g, J, H at the full size",
                  "max_rel_err": worst, "tol": PARITY_TOL, "ok": bool(worst <= PARITY_TOL), "per_array": errs}
    t0 = time.perf_counter()
    n = 0
    while n < max_cycles and time.perf_counter() - t0 < budget_s:
        cycle()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "cycles/s", "cores": 1, "kind": "port",
            "sample": f"{n} full cycles of the same workload ({name} LGR {intervals} intervals) in {dt:.1f} s, "
                      f"NumPy oracle, 1 thread of {os.cpu_count()} host CPUs"}, parity


def solver_inputs(system, guess):
    """x of consecutive cycles: two arrays with different values used in turn (every cycle sees a new x; a solver's iterate
    was just written by the solver, i.e. it is warm in the host's caches -- two arrays keep that, eight 4.8 MB arrays of
    the 40k-node system would come from DRAM every time)."""
    from pockit_amd import benchmarks as models

    x, lam, sigma = models.bench_inputs(system, guess)
    return [x * (1.0 + 1e-9 * k) for k in range(2)], lam, sigma

