"""CPU (gloo, world_size 2) tests of what `bench.py --gpus N` says about a multi-GPU run and of how the N > 1 set-up
degrades -- no GPU needed: the line is assembled from a result record, the machine facts are gathered over gloo, and the
peer-mapped mailboxes are set up against a stand-in of the C ABI whose hipIpc mapping fails on ONE rank."""
import argparse
import ctypes as C
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import models  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _record(n_gpus):
    """A result record shaped like ``bench.measure`` returns it for N ranks (numbers are placeholders)."""
    rec = dict(name="planar_quadrotor", intervals=2000 * n_gpus, nodes=12000 * n_gpus, n=96008, m=96000, nnz_J=755925, nnz_H=359972,
               steps=20, batches=500, median_batch_ms=0.1, ms_per_step=0.005, event_group=10, batch_launch={"form": "x"},
               batch_ms_p10=0.1, batch_ms_p90=0.1, batch_ms_min=0.1, batch_ms_max=0.1, region_wall_s=0.05, wall_ms_per_step=0.005,
               untimed_launches=1500, setup_s=0.2, compile_s_in_setup=0.0, dominant="pk_cycle", exchange="sums", tiles=336, ipw=6,
               bytes={"cycle": 15071568, "cycle_x_once": 11999312}, no_exchange_ms_per_step=0.005,
               exchange_forms_ms_per_step={"sums": 0.006, "direct": 0.03, "gather": 3.0, "allgather": "RuntimeError('x')"},
               dispatch_isolated_us=4.4, dispatch_in_flight_us=6.0, dispatch_samples=[200, 40], kernel_us={"pk_cycle": 4.5},
               ranks=[{"rank": r} for r in range(n_gpus)], exchange_check={"finite": True}, finite=True, side={}, end_to_end=None)
    rec["multi_gpu"] = {"ranks": n_gpus, "ranks_seen_by_rccl": n_gpus, "backend": "nccl (= RCCL on ROCm)", "devices": [{"rank": r, "device": r} for r in range(n_gpus)],
                        "peer_access": [[True] * n_gpus for _ in range(n_gpus)], "headline_form": "host-landed", "device_resident_form": "sums",
                        "device_resident_form_fallback": None, "peer_exchange_error": None}
    rec["end_to_end_host_sharded"] = {"cycles_per_s": 3000.0, "ms_per_cycle": 1 / 3.0, "batches": 5, "steps": 20,
                                      "batch_ms_min_p10_p90_max": [6, 6, 7, 7], "ranks": n_gpus, "finite": True}
    return rec


@pytest.mark.parametrize("n_gpus", [2, 8])
def test_the_line_of_a_multi_gpu_run_carries_what_the_run_saw(n_gpus):
    from tools.benchlib import line as L

    args = argparse.Namespace(steps=20, warmup=5, workload="planar_quadrotor", gpus=n_gpus)
    rec = _record(n_gpus)
    e2e = rec["end_to_end_host_sharded"]
    strong = {f"planar_quadrotor_2000_strong_scaled_over_{n_gpus}": {
                  "device_resident": {"cycles_per_s": 250000.0, "ms_per_step": 0.004, "exchange": "sums",
                                      "exchange_forms_ms_per_step": {"sums": 0.004, "gather": 0.5}}, "end_to_end_cycles_per_s": 9000.0},
              f"two_stage_rocket_1000_strong_scaled_over_{n_gpus}": {"error": "RuntimeError('no')"},
              f"humanoid_wbc_5000_strong_scaled_over_{n_gpus}": {
                  "device_resident": {"cycles_per_s": 90000.0, "ms_per_step": 1 / 90.0, "exchange": "sums",
                                      "exchange_forms_ms_per_step": {"sums": 1 / 90.0, "gather": "failed"}}, "end_to_end_cycles_per_s": None}}
    line = L.short_line(args, rec, e2e, n_gpus, 2000 * n_gpus, rec["ms_per_step"], ROOT, strong=strong)
    text = L.dumps_line(line)                                   # one strict JSON line the driver can parse
    # fixed keys beside `value` -- the RCCL reassembly north_star names, the strong-scaled BASELINE
    # workloads in plain cycles/s, the host-landed cycle; a form that failed says so instead of vanishing
    mg = json.loads(text)["multi_gpu"]
    assert mg["gather_rccl"]["value"] == pytest.approx(n_gpus * 1e3 / 3.0, rel=1e-4) and mg["gather_rccl"]["ms_per_step"] == 3.0
    assert mg["gather_rccl"]["allgather"]["value"] is None and "RuntimeError" in mg["gather_rccl"]["allgather"]["error"]
    assert set(mg["strong"]) == {"C3_12k", "C4_2x1000x4", "C5_5000x8"}
    assert mg["strong"]["C3_12k"] == {"value": 250000.0, "ms_per_step": 0.004, "gather_rccl_value": 2000.0, "host_landed_value": 9000.0}
    assert mg["strong"]["C4_2x1000x4"]["value"] is None and "no" in mg["strong"]["C4_2x1000x4"]["error"]
    assert mg["strong"]["C5_5000x8"]["gather_rccl_value"] is None
    assert mg["host_landed"]["value"] == pytest.approx(n_gpus * 3000.0)
    # under the gloo rehearsal the RCCL key is null (the backend string says why)
    rec_g = _record(n_gpus)
    rec_g["multi_gpu"].update(ranks_seen_by_rccl=None, backend="gloo (rehearsal: NOT a measurement)")
    lg = L.short_line(args, rec_g, e2e, n_gpus, 2000 * n_gpus, rec["ms_per_step"], ROOT)
    assert lg["multi_gpu"]["ranks_seen_by_rccl"] is None and lg["multi_gpu"]["ranks"] == n_gpus and lg["multi_gpu"]["strong"] is None
    assert len(text) < 4096 and json.loads(text)["n_gpus"] == n_gpus
    assert line["scaling"] == "weak" and line["unit"] == "12k-node-equivalent cycles/s"
    # value = the device-resident cycle (inputs and outputs in HBM), N x the per-GPU launch rate; never the PCIe-inclusive one
    assert line["value_is"] == "device_resident"
    assert line["value"] == pytest.approx(n_gpus * 1e3 / rec["ms_per_step"]) and line["device_resident"]["value"] == line["value"]
    # the form that hands the solver the reassembled triplets stays beside it, named
    assert line["end_to_end"]["value"] == pytest.approx(n_gpus * 3000.0) and line["end_to_end"]["ms_per_step"] == pytest.approx(1 / 3.0)
    assert line["multi_gpu"]["ranks_seen_by_rccl"] == n_gpus and "backend" in line["multi_gpu"]
    assert line["roofline"]["kernel"] == "pk_cycle" and line["roofline"]["regime"] in ("latency", "mall", "hbm")
    # the detail record keeps what the run saw of the machine
    d = L.detail_record(args, rec, e2e, n_gpus, 2000 * n_gpus, rec["ms_per_step"], rec["wall_ms_per_step"], ROOT)
    mg = d["multi_gpu"]
    assert len(mg["peer_access"]) == n_gpus and len(mg["devices"]) == n_gpus
    for key in ("backend", "headline_form", "device_resident_form", "device_resident_form_fallback", "peer_exchange_error"):
        assert key in mg
    assert set(("sums", "direct", "gather")) <= set(d["exchange_forms"]["ms_per_step"])
    # a failed host-landed leg leaves the headline alone and says what failed
    line2 = L.short_line(args, rec, {"error": "x"}, n_gpus, 2000 * n_gpus, rec["ms_per_step"], ROOT)
    assert line2["value"] == line["value"] and line2["end_to_end"]["value"] is None and line2["end_to_end"]["error"] == "x"


class _FakeLib:
    """Stand-in of libpockit_hip.so for the mailbox set-up: allocation and export work, the mapping of a PEER's handle
    fails on the ranks in ``bad`` (what a pair of GPUs without peer access answers)."""

    def __init__(self, rank, bad):
        self.rank, self.bad, self.calls = rank, bad, []
        self._keep = []

    def pk_device_alloc(self, h, nbytes, fine, out):
        buf = C.create_string_buffer(int(nbytes))
        self._keep.append(buf)
        out._obj.value = C.addressof(buf)
        return 0

    def pk_ipc_export(self, h, ptr, handle):
        handle.raw = bytes([self.rank]) * 64
        return 0

    def pk_ipc_open(self, h, handle, out):
        self.calls.append("open")
        if self.rank in self.bad:
            return 7
        out._obj.value = 0x1000 + handle.raw[0]
        return 0

    def pk_ipc_close(self, h, p):
        self.calls.append("close")
        return 0

    def pk_device_free(self, h, p):
        self.calls.append("free")
        return 0

    def pk_set_exchange(self, *a):
        self.calls.append("set_exchange")
        return 0

    def pk_last_error(self, h):
        return b"hipIpcOpenMemHandle: invalid device ordinal"


class _FakeCtx:
    def __init__(self, lib):
        self.lib, self.handle = lib, C.c_void_p(1)

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libpockit_hip error {rc}: {self.lib.pk_last_error(self.handle).decode()}")


class _FakeEvaluator:
    def __init__(self, lib):
        self.ctx = _FakeCtx(lib)


def _mailbox_worker(rank, world, port, bad, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pockit_amd.radau as radau
    from pockit_amd.sharding import PeerMailboxes

    system, _, _ = models.brachistochrone(radau, 4, 3)
    lib = _FakeLib(rank, bad)
    outcome = "ok"
    try:
        PeerMailboxes(torch, _FakeEvaluator(lib), system.plan, rank, world, dist, torch.device("cpu"))
    except RuntimeError as exc:
        outcome = str(exc)
    ret[rank] = (outcome, "set_exchange" in lib.calls, lib.calls.count("free"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad", [(), (1,)])
def test_a_peer_mapping_that_fails_on_one_rank_degrades_every_rank_alike(bad):
    """hipIpc mapping refused on rank 1 only: BOTH ranks must give the mailboxes up with the same message (and free what
    they had allocated), so that every rank takes the same fallback; with no failure both set the exchange up."""
    world, port = 2, _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_mailbox_worker, args=(world, port, bad, ret), nprocs=world, join=True)
        got = dict(ret)
    if not bad:
        assert all(v[0] == "ok" and v[1] for v in got.values())
        return
    msgs = {v[0] for v in got.values()}
    assert len(msgs) == 1 and "peer-mapped mailboxes are not available" in next(iter(msgs)) and "rank 1" in next(iter(msgs))
    assert not any(v[1] for v in got.values())            # nobody configured the exchange
    assert all(v[2] == 1 for v in got.values())           # everybody released its own mailbox


def _facts_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.benchlib import timing

    w = argparse.Namespace(exchange="gather", exchange_fallback="sums: peers' flags did not arrive", peer_error="no peer access")
    ret[rank] = json.dumps(timing.multi_gpu_facts(torch, dist, rank, world, w), sort_keys=True)
    dist.barrier()
    dist.destroy_process_group()


def test_machine_facts_are_gathered_identically_on_every_rank():
    """multi_gpu_facts walks through its collectives whatever fails locally (here: there is no GPU at all) and every rank
    ends up with the same record: ranks seen by the process group, a device entry and a peer-access row per rank, the
    exchange form and its fallback reason."""
    world, port = 2, _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_facts_worker, args=(world, port, ret), nprocs=world, join=True)
        got = dict(ret)
    assert got[0] == got[1]
    facts = json.loads(got[0])
    assert facts["ranks"] == 2 and facts["ranks_seen_by_rccl"] is None      # (gloo: RCCL has seen nobody)
    assert len(facts["devices"]) == 2 and len(facts["peer_access"]) == 2
    assert "rehearsal" in facts["backend"] and facts["device_resident_form"] == "gather"
    assert facts["device_resident_form_fallback"] and facts["peer_exchange_error"] == "no peer access"


@pytest.mark.parametrize("case", [("planar_quadrotor", "radau", dict(mesh=40, num_point=6)),
                                  ("two_stage_rocket", "lobatto", dict(mesh=[0, 0.2, 0.5, 1.0], num_point=[4, 7, 3])),
                                  ("brachistochrone", "radau", dict(mesh=9, num_point=5))])
def test_every_rank_uploads_only_the_part_of_x_its_tiles_read(case):
    """needed_x_runs: one rank reads all of x; with several ranks every rank's runs hold its tiles' nodes of every variable
    (plus the slot behind them), the phase's t0 / tf and the static parameters, together they cover x, and the share of a
    rank shrinks with the number of ranks."""
    import importlib

    from pockit_amd.codegen import ModelSource
    from pockit_amd.evaluator import Tables
    from pockit_amd.sharding import needed_x_runs, tile_filter

    bname, scheme, kw = case
    system, _, _ = getattr(models, bname)(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
    plan = system.plan
    src = ModelSource(plan)
    assert needed_x_runs(plan, Tables(plan, src, 2), True) == [(0, plan.n)]
    for world in (2, 3):
        cover = np.zeros(plan.n, dtype=np.int64)
        sizes = []
        for r in range(world):
            tb = Tables(plan, src, 2, tile_filter(r, world, plan))
            runs = needed_x_runs(plan, tb, r == 0)
            assert all(0 <= a < b <= plan.n for a, b in runs) and all(b1 <= a2 for (_, b1), (a2, _) in zip(runs, runs[1:]))
            mask = np.zeros(plan.n, dtype=bool)
            for a, b in runs:
                mask[a:b] = True
            sizes.append(int(mask.sum()))
            cover += mask
            assert mask[plan.l_s: plan.r_s].all()
            for k, pp in enumerate(plan.phase_plans):
                lay, base = pp.layout, int(plan.l_p[k])
                assert mask[base + lay.L - 2: base + lay.L].all()
                mine = tb.tiles[(tb.tiles["phase"] == k) & (tb.tiles["nj"] > 0)]
                for t in mine:
                    q0, nq = int(t["q0"]), int(t["nj"]) * int(lay.stride[int(t["j0"])]) + 1
                    for i in range(pp.nx + pp.nu):
                        length = lay.state_len if i < pp.nx else lay.L_m
                        assert mask[base + int(lay.l_v[i]) + q0: base + int(lay.l_v[i]) + min(q0 + nq, length)].all()
        assert (cover >= 1).all()
        assert max(sizes) < plan.n
