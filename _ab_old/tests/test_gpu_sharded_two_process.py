"""Two real processes run pockit_amd.sharding.ShardedEvaluator.cycle -- the exact code `bench.py --gpus N` runs
per rank -- on one MI355X (both ranks on GPU 0; RCCL refuses two ranks on one device, so the two collectives go
through gloo with the host-staged adapter of pockit_amd.sharding).  Every rank must end up with the complete, oracle-equal outputs."""
import os
import socket

import numpy as np
import pytest

import models

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib

        from pockit_amd.sharding import HostStagedCollectives, ShardedEvaluator

        name, scheme, kw = case
        builder = getattr(models, name)
        system, _, guess = builder(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
        ref, _, _ = builder(importlib.import_module(f"oracle.{scheme}"), **kw)
        x, lam, sigma = models.bench_inputs(system, guess)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        sev = ShardedEvaluator(system.plan, rank, world, device=0, intervals_per_wave=2)
        dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
        hd = HostStagedCollectives(dist)
        err = 0.0
        for rep in range(2):                       # twice: buffers are reused between cycles
            torch.cuda.synchronize()
            o = sev.cycle(dx, dlam, sigma, hd)
            torch.cuda.synchronize()
            for key, want in (("grad", ref.gradient(x)), ("g", ref.constraints(x)), ("J", ref.jacobian(x)),
                              ("H", ref.hessian(x, lam, sigma)), ("f", np.array([ref.objective(x)]))):
                got = o[key].cpu().numpy()
                err = max(err, float(np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want)))))
        # gather mode (what bench.py --gpus N times): the complete outputs on rank 0 only
        sev.full.zero_()
        torch.cuda.synchronize()
        o = sev.cycle(dx, dlam, sigma, hd, root=0)
        torch.cuda.synchronize()
        keys = ("grad", "g", "J", "H", "f") if rank == 0 else ()      # (the other ranks keep slices and partial sums)
        want = dict(grad=ref.gradient(x), g=ref.constraints(x), J=ref.jacobian(x), H=ref.hessian(x, lam, sigma),
                    f=np.array([ref.objective(x)]))
        for key in keys:
            got = o[key].cpu().numpy()
            err = max(err, float(np.max(np.abs(got - want[key])) / max(1.0, np.max(np.abs(want[key])))))
        flag = torch.tensor([err])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if rank == 0:
            ret.put(float(flag.item()))
    except Exception as exc:  # noqa: BLE001 -- report instead of hanging the other rank
        if rank == 0:
            ret.put(repr(exc))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [("two_stage_rocket", "radau", dict(mesh=40, num_point=4)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=33, num_point=5)),
                                  ("brachistochrone", "radau", dict(mesh=61, num_point=6))])
def test_two_process_sharded_cycle_matches_oracle(case):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    err = ret.get(timeout=5)
    assert isinstance(err, float), err
    assert err <= 1e-11, err
