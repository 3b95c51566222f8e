#!/usr/bin/env python3
"""Benchmark of the NLP-callback hot path: cycles/sec of (f, grad f, g, J, H) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one NLP-callback cycle (systembase.py:602-835 of the reference): objective, gradient,
constraints, jacobian, hessian on one x, lambda, sigma.  Workload (BASELINE.json metric: "at 10k LGR nodes"):
planar_quadrotor re-meshed on LGR 2000 intervals x 6 points = 12 000 nodes (configs[2]), x = example guess * (1 + 1e-3 U),
lambda ~ N(0,1), sigma = 1, all seeded.  N > 1: 2000*N intervals of the same model sharded by mesh interval over the N GPUs
(weak scaling), figures in 12k-node-equivalent cycles/s.

What is printed: rank 0 prints ONE short JSON line LAST (strict JSON, < 4 KB, fixed keys: tools/benchlib/line.py) and
writes everything else it measured -- timing statistics, per-callback tables, side kernels, the other BASELINE workloads --
to ``bench_detail.json`` beside this file (and to gpurun_out/ when that directory exists).
* ``value`` / ``ms_per_step`` / ``device_resident``: the cycle with x, lambda and all outputs resident in HBM, ONE pk_cycle
  launch per cycle; R back-to-back batches of EXACTLY ``steps`` launches, HIP events on the launch stream, barrier +
  synchronize around the region, median batch / steps, max over ranks.  ``roofline`` is computed from this region.
* ``end_to_end``: the host-landed (solver-visible) cycle -- the five callbacks of ``System`` on a new x, NumPy arrays in,
  caller-owned NumPy arrays out, reference triplet layout; PCIe-inclusive, therefore never ``value``.
* ``cpu_baseline``: the oracle (CPU restatement of the reference) on the same workload and inputs, one host thread, a
  bounded sample; ``parity``: every entry of the GPU path's f, grad f, g, J, H on those inputs against the oracle's.

Layout: tools/benchlib/workloads.py (workloads, inputs, algorithmic bytes, CPU baseline), timing.py (the GPU legs),
line.py (assembly of the line and of the detail record; importable without a GPU)."""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
# (multi-process GPU work on this pool: the host driver only supports dmabuf IPC -- hipIpc handles of the peer mailboxes, RCCL)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse  # noqa: E402
import json  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchlib import line as L  # noqa: E402
from tools.benchlib.workloads import algorithmic_bytes, build_workload, cpu_baseline, solver_inputs  # noqa: E402,F401


def __getattr__(name):
    """The developer tools under tools/ take the timing legs (GpuWorkload, measure, five_callbacks, timed_cycles ...) from this
    module, as before the split: they live in tools/benchlib/timing.py."""
    from tools.benchlib import timing

    try:
        return getattr(timing, name)
    except AttributeError:
        raise AttributeError(f"module 'bench' has no attribute {name!r}") from None


def write_detail(detail):
    """bench_detail.json beside bench.py (+ a copy under gpurun_out/ when present, so that it comes back from a GPU box)."""
    text = json.dumps(L.sanitize(detail), allow_nan=False, indent=1)
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as fh:
                fh.write(text)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    return written


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks under torch.distributed.run (nothing in this
    process has touched the GPU yet) and forward their output and exit code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)        # cycles per batch; the region is R batches, >= 50 ms in total
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", default="planar_quadrotor")
    ap.add_argument("--intervals", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the supplementary workloads")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    import torch

    from tools.benchlib.timing import cold_compile_seconds, measure, other_workloads, strong_scaled_workloads

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
    res = measure(args.workload, intervals, args.steps, args.warmup, rank, world, dist,
                  with_e2e=not args.no_end_to_end, keep_outputs=not args.no_cpu_baseline)
    t = torch.tensor([res["ms_per_step"], res["wall_ms_per_step"]], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms, wall_ms = float(t[0].item()), float(t[1].item())

    # N > 1: BASELINE.json's sharded configs at their full size, STRONG-scaled over the N GPUs (every rank takes part)
    # (they feed the line's fixed key multi_gpu.strong, so they run unless --no-extra asks for the headline alone)
    strong = strong_scaled_workloads(args, rank, world, dist) if (world > 1 and not args.no_extra) else {}

    if rank == 0:
        e2e = res.get("end_to_end") if world == 1 else res.get("end_to_end_host_sharded")
        detail = L.detail_record(args, res, e2e, n_gpus, intervals, ms, wall_ms, ROOT)
        cb, parity = None, None
        if not args.no_cpu_baseline and n_gpus == 1:      # (required by the contract: a failure here is a failure of the run)
            cb, parity = cpu_baseline(args.workload, intervals, gpu_outputs=res.get("outputs"))
            detail["cpu_baseline"], detail["parity"] = cb, parity
        if n_gpus == 1 and not args.no_extra:
            try:
                detail["compile_s_cold"] = cold_compile_seconds(args.workload, intervals)
            except Exception as exc:  # noqa: BLE001
                detail["compile_s_cold"] = repr(exc)
            detail["other_workloads"] = other_workloads(args)
        elif n_gpus > 1 and not args.no_extra:
            detail["other_workloads"] = strong
        short_parity = None if parity is None else {k: parity[k] for k in ("max_rel_err", "tol", "ok")}
        line = L.short_line(args, res, e2e, n_gpus, intervals, ms, ROOT, cpu_baseline=cb, parity=short_parity, strong=strong)
        detail["line"] = line
        line["detail_file"] = write_detail(detail)
        sys.stdout.flush()
        print(L.dumps_line(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
