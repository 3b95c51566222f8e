#!/usr/bin/env python3
"""Benchmark of the NLP-callback hot path: cycles/sec of (f, grad f, g, J, H) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one NLP-callback cycle -- objective, gradient, constraints, Jacobian, Hessian of the
Lagrangian on the same x, in IPOPT's order -- with x and lambda already
resident in HBM and all outputs left in HBM.

Workload (BASELINE.json metric: "at 10k LGR nodes"): planar_quadrotor re-meshed on LGR 2000
intervals x 6 points = 12 000 nodes (configs[2], the ~10k-node headline of BASELINE.md), x =
example guess * (1 + 1e-3 U), lambda ~ N(0,1), sigma = 1, all seeded.  For N > 1 the mesh is
2000*N intervals of the same model, sharded by mesh interval over the N GPUs (weak scaling: 2000
intervals per GPU) with RCCL gather reassembly of grad/g/J/H on rank 0 (where the host-side solver runs); ``value`` is
then reported in 12k-node-equivalent cycles/s (= N * steps / time).

One JSON line is printed by rank 0 (see the repository prompt for the contract), carrying
``roofline`` for the dominant kernel (HIP-event timed on the launch stream inside the timed
region) and ``cpu_baseline`` (the oracle = CPU restatement of the reference, timed on the host).
"""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import argparse  # noqa: E402
import ctypes as C  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
KERNEL_IDS = {"pk_int": 0, "pk_fin": 1, "pk_g": 2, "pk_grad": 3, "pk_jac": 4, "pk_hess": 5, "pk_xall": 6, "pk_cycle": 12}


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
    # fused x-kernel (pk_xall): x is read once, f partials + grad + g + J written once
    per["xall"] = 8 * (n + 1 + n + m + plan.nnz_J)
    # the same cycle when x is counted once (the single-launch pk_cycle reads it once per wave role, from L2)
    per["cycle_x_once"] = per["cycle"] - 8 * 4 * n
    return per


def build_workload(name, intervals, ns):
    import models

    if name.endswith("_lgl"):                       # Lobatto variant of a workload (side line of the bench)
        import pockit_amd.lobatto as lobatto

        return build_workload(name[:-4], intervals, lobatto)
    if name == "planar_quadrotor":
        return models.planar_quadrotor(ns, intervals, 6)
    if name == "brachistochrone":
        return models.brachistochrone(ns, intervals, 8)
    if name == "two_stage_rocket":
        return models.two_stage_rocket(ns, intervals, 4)
    if name == "humanoid_wbc":
        return models.humanoid_wbc(ns, intervals, 8)
    raise ValueError(name)


def cpu_baseline(name, intervals, budget_s=12.0, max_cycles=5000):
    """The oracle (NumPy restatement of the reference algorithm, single thread) on the same workload."""
    import models
    import oracle.radau

    system, _, guess = build_workload(name, intervals, oracle.radau)
    x, lam, sigma = models.bench_inputs(system, guess)

    def cycle():
        system.objective(x)
        system.gradient(x)
        system.constraints(x)
        system.jacobian(x)
        system.hessian(x, lam, sigma)

    for _ in range(2):
        cycle()
    t0 = time.perf_counter()
    n = 0
    while n < max_cycles and time.perf_counter() - t0 < budget_s:
        cycle()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "cycles/s", "cores": 1, "kind": "port",
            "sample": f"{n} full cycles of the same workload ({name} LGR {intervals} intervals) in {dt:.1f} s, "
                      f"NumPy oracle, 1 thread of {os.cpu_count()} host CPUs"}


EVENT_PERIOD = 64
MIN_WARMUP = 500          # untimed launches before the timed region (single GPU), whatever --warmup says


def run_gpu(name, intervals, steps, warmup, rank, world, dist, time_kernel=None):
    """Returns dict(ms_per_step, kernel timings, plan facts) for one workload on this rank."""
    import torch

    import models
    import pockit_amd.radau as radau
    from pockit_amd.sharding import ShardedEvaluator

    t0 = time.perf_counter()
    system, _, guess = build_workload(name, intervals, radau)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    dev = torch.device("cuda", torch.cuda.current_device())
    sev = ShardedEvaluator(plan, rank, world, device=dev.index)
    setup_s = time.perf_counter() - t0
    ev = sev.ev
    lib, h = ev.ctx.lib, ev.ctx.handle
    dx = torch.from_numpy(x).to(dev)
    dlam = torch.from_numpy(lam).to(dev)
    o = sev.out
    # a stream of our own: torch's default stream has the null handle, which the C ABI reads as "the context's stream"
    stream = torch.cuda.Stream(device=dev)
    st = C.c_void_p(stream.cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    if world == 1:
        cycle_fn = lib.pk_eval_cycle_dev
        cycle_args = (h, ptr(dx), ptr(dlam), C.c_double(float(sigma)), ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]),
                      ptr(o["J"]), ptr(o["H"]), st)       # built once: the loop below is host-launch bound

        def step():
            rc = cycle_fn(*cycle_args)
            if rc:
                ev.ctx.check(rc)
        ev.ctx.check(lib.pk_set_shard(h, 0, 0, None))
    else:
        exchange = {"root": None if os.environ.get("POCKIT_AMD_BENCH_EXCHANGE") == "allgather" else 0}

        def step():
            sev.cycle(dx, dlam, sigma, dist, root=exchange["root"])   # root 0: triplets reassembled where the solver runs

        if exchange["root"] is not None:
            try:                                       # one untimed cycle: a backend without gather falls back to all-gather
                step()
                torch.cuda.synchronize()
            except (RuntimeError, NotImplementedError) as exc:
                print(f"[bench] gather-to-root exchange not available ({exc!r}); using the all-gather form", file=sys.stderr)
                exchange["root"] = None

    B = algorithmic_bytes(plan)
    fused = not (plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I)
    if fused and world == 1 and os.environ.get("POCKIT_AMD_CYCLE_MODE", "1") == "0":   # A/B: the two-launch form
        ev.set_cycle_mode(False)
        dominant = "pk_xall" if B["xall"] >= B["hess"] else "pk_hess"
    elif fused:
        dominant = "pk_cycle"           # ONE launch does the whole cycle: its algorithmic bytes are SURVEY 8(d)'s B
    else:
        dominant = "pk_jac" if B["jac"] >= B["hess"] else "pk_hess"
    # untimed: the W warm-up steps asked for, and at least MIN_WARMUP launches in total (a cold GPU -- clocks, TLBs,
    # code and tables not yet in the caches -- needs a few hundred launches of 5 us each to reach its steady state)
    for _ in range(max(warmup, MIN_WARMUP if world == 1 else warmup)):
        step()
    torch.cuda.synchronize()
    if dist is not None and world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event timing of the dominant kernel inside the timed region, on every 64th launch: a timed launch
    # (hipExtModuleLaunchKernel + event pair) costs the loop ~3 us, timing all of them would slow it by a third
    ev.profile(1 << KERNEL_IDS[time_kernel or dominant], period=EVENT_PERIOD)
    # ... and one HIP event pair around the whole timed region, recorded on the launch stream
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(stream)
    for _ in range(steps):
        step()
    ev_b.record(stream)
    torch.cuda.synchronize()
    if dist is not None and world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_us = ev_a.elapsed_time(ev_b) * 1e3
    prof = ev.profile_read()
    ev.profile(0)
    launches, total_ms = prof[time_kernel or dominant]
    # N > 1: the same loop without the exchange (every rank keeps its slices), to separate the kernels from the collectives
    no_exchange_elapsed = None
    if world > 1 and fused:
        cargs = (h, ptr(dx), ptr(dlam), C.c_double(float(sigma)), ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]), ptr(o["J"]),
                 ptr(o["H"]), st)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            lib.pk_eval_cycle_dev(*cargs)
        stream.synchronize()
        dist.barrier()
        no_exchange_elapsed = time.perf_counter() - t1
    # per-kernel event timing of every kernel, outside the timed region (diagnostic)
    ev.profile(0x1FFF)
    for _ in range(min(steps, 50)):
        step()
    torch.cuda.synchronize()
    allk = {k: (v[1] / v[0] * 1e3 if v[0] else 0.0) for k, v in ev.profile_read().items()}
    ev.profile(0)
    # compact Hessian layout (optional mode): kernel time and size beside the reference layout
    compact = None
    if ev.src.compact and world == 1:
        hc = torch.zeros(max(plan.nnz_Hc, 1), dtype=torch.float64, device=dev)
        hargs = (h, ptr(dx), ptr(dlam), C.c_double(float(sigma)), ptr(hc), st)
        ev.profile(1 << 9)
        for _ in range(50):
            lib.pk_eval_hessc_dev(*hargs)
        torch.cuda.synchronize()
        n_l, ms_l = ev.profile_read()["pk_hessc"]
        ev.profile(0)
        compact = {"nnz_H_compact": int(plan.nnz_Hc), "nnz_H_reference": int(plan.nnz_H),
                   "pk_hessc_us": ms_l / max(n_l, 1) * 1e3, "finite": bool(torch.isfinite(hc).all())}
    # mesh error estimation: one pk_err launch over all intervals
    mesh_err = None
    if world == 1:
        ev.mesh_error(x)                                           # uploads the tables on first use
        eT = torch.zeros(ev._err_len, dtype=torch.float64, device=dev)
        eI = torch.zeros_like(eT)
        torch.cuda.synchronize()
        ev.profile(1 << 10)
        for _ in range(50):
            lib.pk_eval_mesh_error_dev(h, ptr(dx), ptr(eT), ptr(eI), st)
        torch.cuda.synchronize()
        n_l, ms_l = ev.profile_read()["pk_err"]
        ev.profile(0)
        mesh_err = {"pk_err_us": ms_l / max(n_l, 1) * 1e3, "rows": int(ev._err_len),
                    "finite": bool(torch.isfinite(eT).all() and torch.isfinite(eI).all())}
    # device-resident CSR hand-off: gather of the cycle's J / H triplets into CSR order
    csr = None
    if world == 1:
        mj, mh = ev.csr_map("jac"), ev.csr_map("hess")
        cj = torch.zeros(mj.nnz, dtype=torch.float64, device=dev)
        ch = torch.zeros(mh.nnz, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        csr = {"nnz_J_csr": int(mj.nnz), "nnz_H_csr": int(mh.nnz)}
        for which, src, dst in ((0, o["J"], cj), (1, o["H"], ch)):
            ev.profile(1 << 11)
            n0, ms0 = ev.profile_read()["pk_csr"]
            for _ in range(50):
                lib.pk_gather_csr_dev(h, which, ptr(src), ptr(dst), st)
            torch.cuda.synchronize()
            n1, ms1 = ev.profile_read()["pk_csr"]
            ev.profile(0)
            csr["pk_csr_J_us" if which == 0 else "pk_csr_H_us"] = (ms1 - ms0) / max(n1 - n0, 1) * 1e3
        csr["finite"] = bool(torch.isfinite(cj).all() and torch.isfinite(ch).all())
    # correctness spot-check against what the kernels are supposed to produce: finite outputs
    finite = all(bool(torch.isfinite(o[k]).all()) for k in ("f", "grad", "g", "J", "H"))
    res = dict(name=name, intervals=intervals, nodes=int(sum(pp.layout.L_m for pp in plan.phase_plans)),
               n=plan.n, m=plan.m, nnz_J=plan.nnz_J, nnz_H=plan.nnz_H, elapsed=elapsed, steps=steps,
               ms_per_step=elapsed / steps * 1e3, setup_s=setup_s, bytes=B, dominant=dominant,
               dominant_us=(total_ms / launches * 1e3 if launches else None), kernel_us=allk, finite=finite,
               region_us_per_step=region_us / steps, no_exchange_elapsed=no_exchange_elapsed,
               exchange=("single GPU" if world == 1 else "gather to rank 0" if exchange["root"] == 0 else "all-gather"),
               tiles=int(len(ev.tables.tiles)), ipw=int(ev.tables.intervals_per_wave), compact=compact,
               mesh_err=mesh_err, csr=csr)
    ev.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)      # ~0.15 s of GPU time: a 300-step region (2 ms) sits inside
    ap.add_argument("--warmup", type=int, default=2000)      # one clock-management interval and varies +-7 % run to run
    ap.add_argument("--workload", default="planar_quadrotor")
    ap.add_argument("--intervals", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the supplementary workloads")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in pockit_amd)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("POCKIT_AMD_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
        if backend != "nccl":       # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (not a measurement)
            from pockit_amd.sharding import HostStagedCollectives

            dist = HostStagedCollectives(dist)
    n_gpus = world

    intervals = args.intervals * n_gpus          # weak scaling: per-GPU share stays args.intervals
    res = run_gpu(args.workload, intervals, args.steps, args.warmup, rank, world, dist)
    t = torch.tensor([res["elapsed"]], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = n_gpus * args.steps / elapsed
        dom_bytes = res["bytes"][res["dominant"][3:]] / n_gpus
        sampled_us = res["dominant_us"]
        # A cycle that is ONE launch (pk_cycle): the event pair around the timed region / steps is the average
        # launch duration plus the gap to the next launch -- an upper bound of the kernel's own duration that costs
        # the loop nothing.  (The per-dispatch events of every 64th launch go through hipExtModuleLaunchKernel, whose
        # own overhead shows up inside the pair: they read ~1 us more than rocprofv3 for the same kernel.)
        # (N > 1: the region also holds the exchange, so the kernel's own figure is the sampled one)
        dom_us = res["region_us_per_step"] if res["dominant"] == "pk_cycle" and n_gpus == 1 else sampled_us
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.workload}_{intervals}", {}).get(res["dominant"])
            except Exception:
                traffic = None
        line = {
            "metric": "NLP-callback cycles/sec (f + grad f + g + J + H)",
            "value": value,
            "unit": "cycles/s" if n_gpus == 1 else "12k-node-equivalent cycles/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "untimed_launches_before_the_timed_region": max(args.warmup, MIN_WARMUP if n_gpus == 1 else args.warmup),
            "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload} LGR {intervals} intervals x "
                                   f"{6 if args.workload == 'planar_quadrotor' else 8 if args.workload != 'two_stage_rocket' else 4}"
                                   f" points ({res['nodes']} nodes; n={res['n']}, m={res['m']}, nnz_J={res['nnz_J']}, "
                                   f"nnz_H={res['nnz_H']})",
                       "sharding": "single GPU" if n_gpus == 1 else f"mesh intervals over {n_gpus} GPUs, one pk_cycle launch per "
                                                                    f"rank, RCCL {res['exchange']} of the owned runs of grad/g/J/H "
                                                                    f"(+ the partial sums)",
                       "tiles": res["tiles"], "intervals_per_wave": res["ipw"],
                       "inputs": "example guess*(1+1e-3 U(-1,1)) seed 0; lambda N(0,1) seed 1; sigma 1"},
            "roofline": {"bound": "hbm", "kernel": res["dominant"], "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS if achieved else None), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "algorithmic_bytes_per_launch_x_counted_once": (res["bytes"]["cycle_x_once"] / n_gpus
                                                                         if res["dominant"] == "pk_cycle" and n_gpus == 1 else None),
                         "avg_launch_us": dom_us,
                         "sampled_dispatch_us": sampled_us,
                         "timing": ("one HIP event pair on the launch stream around the timed region / steps "
                                    "(launch duration + gap to the next launch); sampled_dispatch_us: per-dispatch "
                                    f"events on every {EVENT_PERIOD}th launch of the same region"
                                    if res["dominant"] == "pk_cycle" and n_gpus == 1 else
                                    f"HIP events on the launch stream, every {EVENT_PERIOD}th launch of the timed region")},
            "kernels_only_without_exchange": (None if res["no_exchange_elapsed"] is None else {
                "value": n_gpus * args.steps / res["no_exchange_elapsed"], "unit": "12k-node-equivalent cycles/s",
                "note": "rank 0's clock around the same number of per-rank pk_cycle launches with no collective "
                        "(every rank keeps its own slices of grad/g/J/H)"}),
            "kernel_us": res["kernel_us"],
            "cycle_algorithmic_bytes": res["bytes"]["cycle"],
            "setup_s": res["setup_s"],
            "outputs_finite": res["finite"],
            "compact_hessian_mode": res["compact"],
            "mesh_error_estimation": res["mesh_err"],
            "csr_handoff": res["csr"],
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            cb = cpu_baseline(args.workload, intervals)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = value / cb["value"]
        if not args.no_extra and n_gpus == 1:
            extra = {}
            for nm, iv in (("brachistochrone", 1250), ("brachistochrone", 200), ("two_stage_rocket", 1000),
                           ("humanoid_wbc", 5000), ("planar_quadrotor_lgl", 2000)):
                try:
                    r = run_gpu(nm, iv, max(20, args.steps // 3), max(5, args.warmup // 3), 0, 1, None)
                    b = r["bytes"][r["dominant"][3:]]
                    extra[f"{nm}_{iv}"] = {
                        "nodes": r["nodes"], "cycles_per_s": 1e3 / r["ms_per_step"], "ms_per_step": r["ms_per_step"],
                        "dominant": r["dominant"], "dominant_us": r["dominant_us"],
                        "dominant_GBps": (b / ((r["region_us_per_step"] if r["dominant"] == "pk_cycle" else r["dominant_us"])
                                               * 1e-6) / 1e9 if r["dominant_us"] else None),
                        "kernel_us": r["kernel_us"], "cycle_bytes": r["bytes"]["cycle"], "setup_s": r["setup_s"],
                        "compact_hessian_mode": r["compact"]}
                except Exception as exc:  # keep the headline line even if a side workload fails
                    extra[f"{nm}_{iv}"] = {"error": repr(exc)}
            line["other_workloads"] = extra
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
