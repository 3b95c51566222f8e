"""GPU box: soak of the host-landed sharded cycle across PROCESSES (pockit_amd.hostshard with 3 ranks on one GPU): rank 0 plays
the solver for many iterates, the other ranks serve; every array of every iterate is compared bit for bit (f to 1e-13) with what
the same shards' kernels gave when rank 0 drained all ranks before reading (the first pass over the inputs, with a full
wait).  Exercises the GPU-written progress marks across processes, the early grad f / g mark, the speculative Hessian.
usage: python3 tools/hostshard_soak.py [iterates]"""
import os
import socket
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402


def worker(rank, world, port, iters, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pockit_amd import benchmarks as models
    from pockit_amd.hostshard import HostShardedEvaluator
    import pockit_amd.radau as radau

    torch.cuda.set_device(0)
    out = []
    for label, builder, args, n_it in (("quadrotor LGR 900x6", models.planar_quadrotor, (900, 6), iters),
                                       ("rocket LGR 2x600x4", models.two_stage_rocket, (600, 4), iters),
                                       ("brachistochrone LGR 24x8", models.brachistochrone, (24, 8), 2 * iters)):
        system, _, guess = builder(radau, *args)
        hs = HostShardedEvaluator(system.plan, rank, world, dist, device=0, timeout_s=60.0)
        if rank != 0:
            assert hs.serve(), "no command arrived"
            hs.close()
            continue
        x, lam, sigma = models.bench_inputs(system, guess)
        rng = np.random.default_rng(5)
        inputs, want = [], []
        for k in range(6):
            xk = x * (1 + 1e-4 * rng.standard_normal(x.size))
            lk = lam * (1 + 1e-2 * rng.standard_normal(lam.size))
            sk = float(sigma * (1 + 0.1 * k))
            inputs.append((xk, lk, sk))
            f = float(hs.objective(xk))
            H = hs.hessian(xk, lk, sk)
            J = hs.jacobian(xk)
            time.sleep(0.02)                      # (everything of this iterate has long landed, on every rank)
            want.append((f, np.array(hs.gradient(xk)), np.array(hs.constraints(xk)), np.array(J), np.array(H)))
        bad, t0 = 0, time.perf_counter()
        for it in range(n_it):
            k = int(rng.integers(6))
            xk, lk, sk = inputs[k]
            wf, wg, wc, wj, wh = want[k]
            res = {}
            if it % 7 == 3:                    # a rejected trial point: f and g only (the next iterate's J then comes on request)
                ok = float(hs.objective(xk)) == wf and np.array_equal(hs.constraints(xk), wc)
                bad += 0 if ok else 1
                continue
            order = [0, 1, 2, 3] if it % 3 else list(rng.permutation(4))
            if it % 5 == 4:
                res[4] = hs.hessian(xk, lk, sk)
            for w in order:
                res[w] = (hs.objective, hs.gradient, hs.constraints, hs.jacobian)[w](xk)
            if 4 not in res:
                res[4] = hs.hessian(xk, lk, sk)
            ok = (float(res[0]) == wf and np.array_equal(res[1], wg) and np.array_equal(res[2], wc) and np.array_equal(res[3], wj)
                  and np.array_equal(res[4], wh))
            bad += 0 if ok else 1
        out.append(f"{label:26s} {n_it:6d} iterates over {world} processes, {bad} with a mismatch, {time.perf_counter() - t0:.1f} s")
        hs.close()
    if rank == 0:
        ret.put(out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp

    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=worker, args=(r, world, port, iters, ret)) for r in range(world)]
    for p in procs:
        p.start()
    lines = ret.get(timeout=700)
    for p in procs:
        p.join(60)
    print("\n".join(lines))
    bad = any(" 0 with a mismatch" not in ln for ln in lines) or any(p.exitcode != 0 for p in procs)
    print("hostshard soak:", "MISMATCHES or a failed rank" if bad else "all iterates bit-identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
