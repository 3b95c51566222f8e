"""Two real processes run pockit_amd.sharding.ShardedEvaluator.cycle -- the exact code `bench.py --gpus N` runs
per rank -- on one MI355X (both ranks on GPU 0; RCCL refuses two ranks on one device, so the two collectives go
through gloo with the host-staged adapter of pockit_amd.sharding).  Every rank must end up with the complete, oracle-equal outputs."""
import os
import socket

import numpy as np
import pytest

import models

pytestmark = pytest.mark.gpu


class _PlanReference:
    """The NumPy execution of the product's own plan (tests/plan_interp.py) behind the oracle's callback names: the
    reference for meshes whose orders the oracle's np.roots-based tables cannot represent (num_point > 16)."""

    def __init__(self, plan):
        self.plan = plan

    def _it(self, x, lam=None, sigma=1.0):
        from plan_interp import Interp

        return Interp(self.plan, x, lam, sigma)

    def objective(self, x):
        return self._it(x).objective()

    def gradient(self, x):
        return self._it(x).gradient()

    def constraints(self, x):
        return self._it(x).constraints()

    def jacobian(self, x):
        return self._it(x).jacobian()

    def hessian(self, x, lam, sigma):
        return self._it(x, lam, sigma).hessian()


def _reference(system, builder, scheme, kw):
    import importlib

    k = kw.get("num_point", 0)
    if np.max(k) > 16:
        return _PlanReference(system.plan)
    return builder(importlib.import_module(f"oracle.{scheme}"), **kw)[0]



def _worker(rank, world, port, case, ret, backend="gloo"):
    """backend "gloo": both ranks on GPU 0, collectives host-staged (RCCL refuses two ranks on one device).  backend "nccl"
    (= RCCL on ROCm): one GPU per rank, the gather / all-gather of the owned runs go over RCCL itself, device buffers in and out."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    gpu = rank if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(gpu)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import importlib

        from pockit_amd.sharding import HostStagedCollectives, ShardedEvaluator

        name, scheme, kw = case
        builder = getattr(models, name)
        system, _, guess = builder(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
        ref = _reference(system, builder, scheme, kw)
        x, lam, sigma = models.bench_inputs(system, guess)
        dev = torch.device("cuda", gpu)
        torch.cuda.set_device(dev)
        sev = ShardedEvaluator(system.plan, rank, world, device=gpu, intervals_per_wave=2)
        dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
        hd = dist if backend == "nccl" else HostStagedCollectives(dist)
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
        flag = torch.tensor([float(err)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if rank == 0:
            ret.put(float(flag.item()))
    except Exception as exc:  # noqa: BLE001 -- report instead of hanging the other rank
        if rank == 0:
            ret.put(repr(exc))
        raise
    finally:
        dist.destroy_process_group()


def _peer_worker(rank, world, port, case, ret):
    """The collective-free forms of the exchange: "sums" (every rank keeps its slices; the sums over all nodes go through
    peer-mapped mailboxes inside one launch) and "direct" (the other rank's kernels store into rank 0's buffer through
    a hipIpc mapping).  gloo only carries the 64-byte handles at set-up."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib

        from pockit_amd.sharding import ShardedEvaluator, shared_gradient_slots

        name, scheme, kw = case
        builder = getattr(models, name)
        system, _, guess = builder(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
        ref = _reference(system, builder, scheme, kw)
        x, lam, sigma = models.bench_inputs(system, guess)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        plan = system.plan
        sev = ShardedEvaluator(plan, rank, world, device=0, intervals_per_wave=2)
        sev.enable_peer_exchange(dist, root=0)
        dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
        want = np.concatenate([ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma)])
        f_want = ref.objective(x)
        shared = shared_gradient_slots(plan)
        scale = lambda a: max(1.0, float(np.max(np.abs(a)))) if a.size else 1.0  # noqa: E731
        err = 0.0
        for rep in range(4):                                      # the mailboxes alternate between their two halves
            sev.inline_exchange = rep % 2 == 0                     # exchange inside pk_cycle's launch / as pk_xchg behind it
            xk = x * (1.0 + 1e-3 * rep)
            if rep:
                want = np.concatenate([ref.gradient(xk), ref.constraints(xk), ref.jacobian(xk), ref.hessian(xk, lam, sigma)])
                f_want = ref.objective(xk)
            dx.copy_(torch.from_numpy(xk))
            # ---- "sums": own slices + complete shared slots and f on EVERY rank
            sev.full.fill_(float("nan"))
            torch.cuda.synchronize()
            dist.barrier()
            o = sev.cycle(dx, dlam, sigma, dist, exchange="sums")
            torch.cuda.synchronize()
            got = sev.full.cpu().numpy()[:-1]
            mine = np.zeros(len(want), dtype=bool)
            for a, b in sev.runs[rank]:
                mine[a:b] = True
            mine[shared] = True
            err = max(err, float(np.max(np.abs(got[mine] - want[mine])) / scale(want)))
            rest = ~mine
            assert np.all(np.isnan(got[rest])), "a rank wrote outside its own runs"
            err = max(err, abs(float(o["f"].cpu()[0]) - f_want) / max(1.0, abs(f_want)))
            dist.barrier()
            # ---- "direct": everything reassembled in rank 0's buffer by peer stores
            if rank == 0:
                sev.full.fill_(float("nan"))
            torch.cuda.synchronize()
            dist.barrier()
            o = sev.cycle(dx, dlam, sigma, dist, exchange="direct")
            torch.cuda.synchronize()
            dist.barrier()
            if rank == 0:
                got = sev.full.cpu().numpy()[:-1]
                err = max(err, float(np.max(np.abs(got - want)) / scale(want)))
                err = max(err, abs(float(o["f"].cpu()[0]) - f_want) / max(1.0, abs(f_want)))
            dist.barrier()
        flag = torch.tensor([float(err)], dtype=torch.float64)   # (one dtype on every rank, whatever produced the maximum)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        sev.close()
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
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.2, 0.3, 0.5, 0.8, 1.0], num_point=[70, 5, 6, 66, 7]))])
def test_two_process_peer_exchange_without_collectives(case):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, 2, port, case, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    err = ret.get(timeout=5)
    assert isinstance(err, float), err
    assert err <= 1e-11, err


@pytest.mark.parametrize("case", [("two_stage_rocket", "radau", dict(mesh=40, num_point=4)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=33, num_point=5)),
                                  ("brachistochrone", "radau", dict(mesh=61, num_point=6)),
                                  # objective / system constraints nonlinear in the integrals: the integrals are summed over the
                                  # ranks before the other kernels, the outer-product blocks are formed by rank 0 from the summed
                                  # auxiliary entries (easyderiv.py:323-459); the second mesh has a workgroup-wide interval
                                  ("derivative_model", "radau", dict(mesh=(0, .1, .2, .35, .5, .6, .75, .9, 1), num_point=(3, 4, 3, 4, 5, 4, 3, 4))),
                                  ("derivative_model", "lobatto", dict(mesh=(0, .2, .5, .7, 1), num_point=(4, 70, 5, 6)))])
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


def _host_worker(rank, world, port, case, ret):
    """The host-landed sharded cycle (pockit_amd.hostshard): rank 0 serves the five callbacks from ONE shared pinned host
    segment into which every rank's run-copy kernel stores its own slices; the other ranks ``serve()``."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib

        from pockit_amd.hostshard import HostShardedEvaluator

        name, scheme, kw = case
        builder = getattr(models, name)
        system, _, guess = builder(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
        torch.cuda.set_device(0)
        hs = HostShardedEvaluator(system.plan, rank, world, dist, device=0, intervals_per_wave=2, timeout_s=60.0)
        err = 0.0
        if rank != 0:
            assert hs.serve(), "no command arrived"
        else:
            ref = _reference(system, builder, scheme, kw)
            x, lam, sigma = models.bench_inputs(system, guess)
            scale = lambda a: max(1.0, float(np.max(np.abs(a)))) if np.size(a) else 1.0  # noqa: E731
            for rep in range(3):
                xk = x * (1.0 + 1e-3 * rep)
                want = dict(f=ref.objective(xk), grad=ref.gradient(xk), g=ref.constraints(xk), J=ref.jacobian(xk),
                            H=ref.hessian(xk, lam * (1 + rep), sigma + rep))
                hs.h_out[:] = np.nan                                   # every position must be written by somebody
                got = dict(f=hs.objective(xk), grad=hs.gradient(xk), g=hs.constraints(xk), J=hs.jacobian(xk))
                got["H"] = hs.hessian(xk, lam * (1 + rep), sigma + rep)
                for key in ("f", "grad", "g", "J", "H"):
                    a, b = np.asarray(got[key], dtype=np.float64), np.asarray(want[key], dtype=np.float64)
                    assert a.shape == b.shape, key
                    err = max(err, float(np.max(np.abs(a - b)) / scale(b)) if b.size else 0.0)
            # a line search: two rejected trial points (objective and constraints only), then an accepted one whose J slices
            # were not sent ahead and come on request (CMD_X_TRIAL / CMD_J)
            for rep in range(3, 6):
                xk = x * (1.0 + 1e-3 * rep)
                got_f, got_g = hs.objective(xk), hs.constraints(xk)
                err = max(err, abs(float(got_f) - ref.objective(xk)) / scale(np.array([ref.objective(xk)])),
                          float(np.max(np.abs(np.asarray(got_g) - ref.constraints(xk))) / scale(ref.constraints(xk))))
            xk = x * (1.0 + 7e-3)
            hs.h_out[:] = np.nan
            got = dict(f=hs.objective(xk), g=hs.constraints(xk), grad=hs.gradient(xk), J=hs.jacobian(xk), H=hs.hessian(xk, lam, sigma))
            want = dict(f=ref.objective(xk), grad=ref.gradient(xk), g=ref.constraints(xk), J=ref.jacobian(xk), H=ref.hessian(xk, lam, sigma))
            for key in ("f", "grad", "g", "J", "H"):
                a, b = np.asarray(got[key], dtype=np.float64), np.asarray(want[key], dtype=np.float64)
                err = max(err, float(np.max(np.abs(a - b)) / scale(b)) if b.size else 0.0)
            err = float("inf") if not np.isfinite(err) else err
        hs.close()
        flag = torch.tensor([float(err)], dtype=torch.float64)   # (one dtype on every rank, whatever produced the maximum)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if rank == 0:
            ret.put(float(flag.item()))
    except Exception as exc:  # noqa: BLE001 -- report instead of hanging the other rank
        if rank == 0:
            ret.put(repr(exc))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [("two_stage_rocket", "radau", dict(mesh=40, num_point=4)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=33, num_point=5)),
                                  ("brachistochrone", "radau", dict(mesh=[0, 0.2, 0.3, 0.5, 0.8, 1.0], num_point=[70, 5, 6, 66, 7])),
                                  # nonlinear in the integrals: the integrals go round first (CMD_INT), rank 0 forms the outer-product
                                  # Hessian blocks from every rank's auxiliary entries
                                  ("derivative_model", "radau", dict(mesh=(0, .1, .2, .35, .5, .6, .75, .9, 1), num_point=(3, 4, 3, 4, 5, 4, 3, 4))),
                                  ("derivative_model", "lobatto", dict(mesh=(0, .2, .5, .7, 1), num_point=(4, 70, 5, 6)))])
def test_host_landed_sharded_cycle_matches_oracle(case, world):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_host_worker, args=(r, world, port, case, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    err = ret.get(timeout=5)
    assert isinstance(err, float), err
    assert err <= 1e-11, err


@pytest.mark.parametrize("form", ["device", "peer", "host"])
@pytest.mark.parametrize("case", [("humanoid_wbc", "radau", dict(mesh=24, num_point=4)),
                                  ("two_stage_rocket", "lobatto", dict(mesh=30, num_point=5))])
def test_sharded_cycle_of_a_model_evaluated_in_groups(case, form, monkeypatch):
    """The derivative set in groups of three, every pass of the cycle a workgroup of its own, each rank
    on its share of the mesh intervals: the sharded forms (collectives, peer stores, host-landed) against the oracle."""
    import torch.multiprocessing as mp

    monkeypatch.setenv("POCKIT_AMD_GROUP_CAP", "3")         # (the spawned ranks inherit the environment)
    monkeypatch.setenv("POCKIT_AMD_PASS_PARALLEL", "1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    target = dict(device=_worker, peer=_peer_worker, host=_host_worker)[form]
    procs = [ctx.Process(target=target, args=(r, 2, port, case, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    err = ret.get(timeout=5)
    assert isinstance(err, float), err
    assert err <= 1e-11, err


@pytest.mark.parametrize("case", [("two_stage_rocket", "radau", dict(mesh=40, num_point=4)),
                                  ("planar_quadrotor", "lobatto", dict(mesh=33, num_point=5))])
def test_two_process_sharded_cycle_over_rccl(case):
    """The RCCL forms of the reassembly -- all-gather and gather-to-root of the packed owned runs (sharding.py, what
    ``bench.py --gpus N`` times as the "gather" exchange form) -- with backend ``nccl`` (= RCCL on ROCm) and ONE GPU PER RANK:
    device buffers straight into the collectives over xGMI.  Needs two visible GPUs; on a one-GPU box (this pool's) it is
    skipped, and the same code runs over gloo with host staging in test_two_process_sharded_cycle_matches_oracle."""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, ret, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    err = ret.get(timeout=5)
    assert isinstance(err, float), err
    assert err <= 1e-11, err
