"""CPU (gloo, world_size 2): HostShardedEvaluator's set-up agrees on every rank's verdict BEFORE its first collective
(ADVICE r3): when the rank-local construction (code generation, hipcc, pk_load_model ...) fails on ONE rank, every rank
raises the same error naming that rank -- nobody is left in the broadcast / barrier of the set-up -- and the healthy rank
releases what it had built.  The GPU evaluator is replaced by a stand-in (no GPU here)."""
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import models


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeEvaluator:
    closed = 0

    def __init__(self, plan, **kw):
        if int(os.environ["RANK_FOR_TEST"]) in [int(v) for v in os.environ["BAD_RANKS"].split(",") if v]:
            raise RuntimeError("hipcc failed (stand-in)")

    def close(self):
        _FakeEvaluator.closed += 1


def _worker(rank, world, port, bad, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK_FOR_TEST=str(rank), BAD_RANKS=",".join(map(str, bad)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pockit_amd.radau as radau
        from pockit_amd import hostshard

        hostshard.Evaluator = _FakeEvaluator
        system, _, _ = models.brachistochrone(radau, 6, 4)
        try:
            hostshard.HostShardedEvaluator(system.plan, rank, world, dist)
            ret[rank] = ("constructed", _FakeEvaluator.closed)
        except RuntimeError as exc:
            ret[rank] = (str(exc), _FakeEvaluator.closed)
        dist.barrier()                      # (every rank got here: nobody hangs in the set-up's collectives)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad", [(1,), (0,), (0, 1)])
def test_a_rank_whose_local_set_up_fails_makes_every_rank_raise_the_same_error(bad):
    world, port = 2, _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_worker, args=(world, port, bad, ret), nprocs=world, join=True)
        got = dict(ret)
    msgs = {v[0] for v in got.values()}
    assert len(msgs) == 1, got
    msg = next(iter(msgs))
    assert "rank-local set-up failed" in msg and all(f"rank {r}" in msg for r in bad)
    for r in range(world):
        assert got[r][1] == (0 if r in bad else 1)       # the healthy rank closed its evaluator, the failed one had none
