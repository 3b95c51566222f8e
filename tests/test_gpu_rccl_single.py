"""ONE real RCCL execution on a single GPU (VERDICT r4 'Next round' 4).  The pool's boxes have one GPU, RCCL refuses two ranks
on a device -- but it accepts a process group of ONE rank.  ``ShardedEvaluator(..., reassemble_single=True)`` runs the
"gather" and "allgather" forms (the all-gatherv of north_star: run-copy pack, RCCL collective, run-copy unpack into the
reference's triplet order, ``sharding.Reassembler``) through backend ``nccl`` on device tensors: what the gloo rehearsals
cannot exercise is the ordering between the evaluator's own HIP stream and the stream RCCL enqueues its kernels on (gloo
stages through the host, which synchronizes where RCCL does not).  200 cycles back to back with a changing x must each
show that iterate's values, not the previous one's."""
import os
import socket

import numpy as np
import pytest

import models

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world_of_one():
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [("planar_quadrotor", "radau", dict(mesh=200, num_point=6)),
                                  ("two_stage_rocket", "lobatto", dict(mesh=60, num_point=4)),
                                  ("humanoid_wbc", "radau", dict(mesh=100, num_point=8))])
def test_reassembly_over_rccl_matches_the_oracle_and_keeps_its_stream_order(case, rccl_world_of_one):
    import importlib

    import torch

    from pockit_amd.sharding import ShardedEvaluator

    dist = rccl_world_of_one
    bname, scheme, kw = case
    system, _, guess = getattr(models, bname)(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
    ref, _, _ = getattr(models, bname)(importlib.import_module(f"oracle.{scheme}"), **kw)
    plan = system.plan
    x, lam, sigma = models.bench_inputs(system, guess)
    dev = torch.device("cuda", 0)
    sev = ShardedEvaluator(plan, 0, 1, device=0, reassemble_single=True)
    assert sev.re is not None and sev.re.world == 1
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    want = (ref.objective(x), ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma))
    try:
        for form in ("allgather", "gather"):
            sev.full.zero_()
            sev.re.recv.fill_(float("nan"))
            torch.cuda.synchronize()
            out = sev.cycle(dx, dlam, sigma, dist, root=(0 if form == "gather" else None), exchange=form)
            torch.cuda.synchronize()
            for name, b in zip(("f", "grad", "g", "J", "H"), want):
                a = out[name].cpu().numpy()
                a = a[0] if name == "f" else a
                err = np.max(np.abs(a - b)) if np.size(b) else 0.0
                assert err <= TOL * max(1.0, np.max(np.abs(b))), f"{bname} {form} {name}: err {err:.3e}"
            if form == "allgather":      # the receive buffer IS the rank's packed runs: the collective carried every owned value
                packed = torch.cat([sev.full[a:b] for a, b in sev.runs[0]])
                assert torch.equal(sev.re.recv[: packed.numel()], packed)
        # 200 cycles back to back, x changing: every iterate's checksums (taken on the CALLER's stream, which `cycle` joins)
        # must equal those of the same x evaluated without any exchange.  A missing dependency between the evaluator's stream
        # and RCCL's would pack before the launch has written, or unpack before the collective has landed: stale sums.
        n_it = 200
        scale = 1.0 + 1e-4 * torch.arange(n_it, dtype=torch.float64, device=dev)
        for form in ("allgather", "gather"):
            sums = torch.zeros(n_it, 4, dtype=torch.float64, device=dev)
            for it in range(n_it):
                xi = dx * scale[it]
                out = sev.cycle(xi, dlam, sigma, dist, root=(0 if form == "gather" else None), exchange=form)
                sums[it, 0], sums[it, 1] = out["f"][0], out["g"].sum()
                sums[it, 2], sums[it, 3] = out["J"].sum(), out["H"].sum()
            torch.cuda.synchronize()
            plain = torch.zeros_like(sums)
            for it in range(n_it):
                xi = dx * scale[it]
                out = sev.cycle(xi, dlam, sigma, None)
                torch.cuda.synchronize()
                plain[it, 0], plain[it, 1] = out["f"][0], out["g"].sum()
                plain[it, 2], plain[it, 3] = out["J"].sum(), out["H"].sum()
            torch.cuda.synchronize()
            assert torch.equal(sums, plain), f"{bname} {form}: a back-to-back iterate shows another iterate's values"
            assert len(torch.unique(sums[:, 2])) > n_it // 2          # (the iterates DO differ)
    finally:
        sev.close()
        system._invalidate()
