"""world_size-2 test of the mesh-interval sharding logic on CPU (gloo).

Each rank takes its share of the tiles (pockit_amd.sharding.tile_filter), "evaluates" only the
output positions those tiles own -- taken from the NumPy plan interpreter, masked by tile ownership
computed from the same tables the kernels consume -- and the ranks reassemble with an all-reduce,
exactly the data path of ShardedEvaluator (RCCL on the GPUs).  The result must equal the unsharded
arrays, i.e. the shards are disjoint and cover everything, and only rank 0 emits the boundary /
system-level entries."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import models


def _owned_mask(plan, tables, cbname, nnz, primary):
    """Positions of the J/H value array written by this shard's tiles (+ scalar items on the primary)."""
    cb = getattr(plan, cbname)
    mask = np.zeros(nnz, dtype=bool)
    for t in tables.tiles:
        k = int(t["phase"])
        lay = plan.phase_plans[k].layout
        kd = lay.kinds[int(t["kid"]) - sum(len(pp.layout.kinds) for pp in plan.phase_plans[:k])]
        nj, j0 = int(t["nj"]), int(t["j0"])
        stride = int(lay.stride[j0])
        q0 = int(t["q0"])
        nq = nj * stride + (1 if lay.scheme == "lgl" else 0)
        nown = nq - 1 if (lay.scheme == "lgl" and not t["last"]) else nq
        for seg in cb.segs[k]:
            if seg.kind == "I":
                lo = seg.base + int(t["offI"])
                mask[lo: lo + nj * kd.nnzI] = True
            else:
                for q in range(q0, q0 + nown):
                    if lay.mid_lo <= q < lay.mid_hi:
                        mask[seg.base + q - lay.mid_lo] = True
        if cbname == "jac":
            for base in cb.tconst[k]:
                lo = base + int(t["offT"])
                mask[lo: lo + nj * kd.nnzT] = True
    if primary:
        for it in cb.items:
            mask[it.pos] = True
    return mask


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pockit_amd.radau as radau
    from plan_interp import Interp
    from pockit_amd.codegen import ModelSource
    from pockit_amd.evaluator import Tables
    from pockit_amd.sharding import tile_filter

    system, _, guess = models.two_stage_rocket(radau, 7, 3)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    src = ModelSource(plan)
    tb = Tables(plan, src, intervals_per_wave=2, tile_filter=tile_filter(rank, world))
    it = Interp(plan, x, lam, sigma)
    full = {"jac": it.jacobian(), "hess": it.hessian()}
    ok = True
    for cbname, nnz in (("jac", plan.nnz_J), ("hess", plan.nnz_H)):
        mask = _owned_mask(plan, tb, cbname, nnz, primary=(rank == 0))
        mine = torch.from_numpy(np.where(mask, full[cbname], 0.0))
        count = torch.from_numpy(mask.astype(np.float64))
        dist.all_reduce(mine)
        dist.all_reduce(count)
        ok &= bool(np.all(count.numpy() == 1.0))          # disjoint and complete
        ok &= bool(np.array_equal(mine.numpy(), full[cbname]))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(flag.item()))
    dist.destroy_process_group()


def test_two_rank_sharding_reassembles_exactly():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get() == 1.0


def test_contiguous_share_partitions():
    from pockit_amd.sharding import contiguous_share

    for n in (0, 1, 7, 8, 2001):
        for world in (1, 2, 3, 8):
            parts = [contiguous_share(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
