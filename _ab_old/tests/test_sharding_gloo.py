"""world_size-2 test of the mesh-interval sharding data path on CPU (gloo).

Each rank takes its share of the tiles (pockit_amd.sharding.tile_filter).  Its kernels' outputs are
emulated with the NumPy plan interpreter: the positions the rank's tiles own carry the true values,
every other position is poisoned (NaN), the gradient's shared slots carry the rank's partial sums.  The
REAL exchange code of the product (pockit_amd.sharding.Reassembler: pack by owned runs, all-gather,
unpack, tiny all-reduce of the shared slots) then reassembles over gloo exactly as it does over RCCL on
the GPUs.  Result must equal the unsharded arrays on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import models


def _worker(rank, world, port, ret, root=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pockit_amd.radau as radau
    from plan_interp import Interp
    from pockit_amd.codegen import ModelSource
    from pockit_amd.evaluator import Tables
    from pockit_amd.sharding import Reassembler, owned_runs, shared_gradient_slots, tile_filter

    system, _, guess = models.two_stage_rocket(radau, 7, 3)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    src = ModelSource(plan)
    runs = [owned_runs(plan, Tables(plan, src, 2, tile_filter(r, world)), r == 0) for r in range(world)]
    re = Reassembler(torch, plan, runs, rank, world, torch.device("cpu"))
    it = Interp(plan, x, lam, sigma)
    truth = np.concatenate([it.gradient(), it.constraints(), it.jacobian(), it.hessian()])
    shared = shared_gradient_slots(plan)
    # what this rank's kernels would have produced
    mine = np.full(len(truth) + 1, np.nan)
    for a, b in runs[rank]:
        mine[a:b] = truth[a:b]
    share = 0.25 if rank == 0 else 0.75            # the shared slots are partial sums per rank
    mine[shared] = truth[shared] * share
    full = torch.from_numpy(mine.copy())
    small = torch.zeros(1 + len(shared), dtype=torch.float64)
    small[0] = 1.0 + rank                           # stands for a partial integral
    re.exchange(full, small, dist, root)
    got = full.numpy()[:-1]
    if root is not None and rank != root:            # gather mode: the other ranks keep their own slices and partial sums
        keep = np.zeros(len(truth), dtype=bool)
        for a, b in runs[rank]:
            keep[a:b] = True
        rest = ~keep
        rest[shared] = False
        assert np.all(np.isnan(got[rest]))
        assert np.array_equal(got[shared], truth[shared] * share)
        got, truth = got[keep], truth[keep]
    ok = bool(np.array_equal(got, truth) or np.allclose(got, truth, rtol=0, atol=1e-15 * np.abs(truth).max()))
    if root is None or rank == root:
        ok &= bool(abs(float(small[0]) - 3.0) < 1e-15)  # integrals summed over the two ranks
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(float(flag.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize("root", [None, 0, 1])
def test_two_rank_sharding_reassembles_exactly(root):
    """root None: all-gather, every rank complete; root r: gather to rank r (what bench.py --gpus N times)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, root)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get() == 1.0


def test_owned_runs_partition_every_output_position():
    """For 1..5 ranks, LGR and LGL, multi-phase: the ranks' runs + the shared gradient slots cover every
    position of [grad | g | J | H] exactly once (Reassembler raises otherwise)."""
    import pockit_amd.lobatto as lobatto
    import pockit_amd.radau as radau
    from pockit_amd.codegen import ModelSource
    from pockit_amd.evaluator import Tables
    from pockit_amd.sharding import Reassembler, owned_runs, tile_filter

    for ns, builder, kw in ((radau, models.two_stage_rocket, dict(mesh=9, num_point=3)),
                            (lobatto, models.brachistochrone, dict(mesh=11, num_point=4)),
                            (lobatto, models.two_stage_rocket, dict(mesh=5, num_point=3)),
                            (radau, models.derivative_model, {})):
        system, _, _ = builder(ns, **kw)
        plan = system.plan
        src = ModelSource(plan)
        for world in (1, 2, 3, 5):
            runs = [owned_runs(plan, Tables(plan, src, 1, tile_filter(r, world)), r == 0) for r in range(world)]
            Reassembler(torch, plan, runs, 0, world, torch.device("cpu"))


def test_contiguous_share_partitions():
    from pockit_amd.sharding import contiguous_share

    for n in (0, 1, 7, 8, 2001):
        for world in (1, 2, 3, 8):
            parts = [contiguous_share(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
