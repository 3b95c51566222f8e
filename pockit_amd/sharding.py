"""Mesh-interval sharding of one NLP across the GPUs of a node (one process per GPU).

Every defect row, path row, Jacobian and Hessian triplet belongs to exactly one mesh interval of
one phase (block-diagonal integration/translation matrices), so the tiles of every phase are
split into ``world`` contiguous ranges; rank r evaluates only its tiles, writing into full-size
output arrays at the reference positions.  Reassembly (``Reassembler``) is an RCCL *all-gather* over
xGMI issued through ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests): every
rank packs the positions it owns (a few dozen contiguous runs: its slice of every segment), the padded
packs are all-gathered, and the other ranks' packs are scattered to their reference positions -- RCCL has
no gather-v, so the pack/unpack index maps play that role.  Only the handful of sums over all nodes (the
objective's integrals and the gradient entries of t0/tf/static parameters) need a reduction: one tiny
all-reduce.

The boundary-node and system-level scalars are computed once, by rank 0 (``pk_set_shard``).
Reference: the reference is single-process; this is the build's own
design for BASELINE.json's "mesh intervals shard across the GPUs" requirement.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def contiguous_share(n_items: int, rank: int, world: int):
    """[lo, hi) of rank's share when n_items are dealt in contiguous, balanced ranges."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def tile_weights(plan, phase_index, tiles):
    """Output doubles every tile of a phase writes (rows of ``MeshLayout.tiles``): its slice of every I-expanded
    segment of J and H (nj * nnzI each), of the translation piece of every state (nj * nnzT), and one value per node
    for every per-node run (N segments, gradient entries, defect rows, path rows).  On an hp mesh (K = 1 .. 12) the
    triplet work of a tile varies by two orders of magnitude, so shares are balanced by this, not by tile count."""
    pp = plan.phase_plans[phase_index]
    lay = pp.layout
    n_I = sum(1 for cb in ("jac", "hess") for sg in getattr(plan, cb).segs[phase_index] if sg.kind == "I")
    n_N = sum(1 for cb in ("jac", "hess") for sg in getattr(plan, cb).segs[phase_index] if sg.kind != "I")
    per_node = n_N + 2 * pp.nx + pp.nu + pp.phase.n_c          # N segments, grad entries, defect rows, path rows
    w = np.zeros(len(tiles), dtype=np.int64)
    for t, row in enumerate(tiles):
        j0, nj, kid = int(row[0]), int(row[1]), int(row[2])
        kd = lay.kinds[kid]
        w[t] = nj * (kd.nnzI * n_I + kd.nnzT * pp.nx) + nj * int(lay.stride[j0]) * per_node
    return w


def balanced_cuts(weights, world: int):
    """``world + 1`` cut indices splitting ``weights`` into contiguous ranges of nearly equal sums: cut r is placed
    where the running sum is closest to r / world of the total (never before the previous cut)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        i = int(np.searchsorted(cum, target))
        if i > 0 and abs(cum[i - 1] - target) <= abs(cum[min(i, len(w))] - target):
            i -= 1
        cuts.append(min(max(i, cuts[-1]), len(w)))
    cuts.append(len(w))
    return cuts


def tile_filter(rank: int, world: int, plan=None):
    """Rank's contiguous share of every phase's tiles: balanced by output doubles when the plan is given (what the
    evaluators use), by tile count otherwise."""
    def keep(phase_index, tiles):
        if plan is None:
            lo, hi = contiguous_share(len(tiles), rank, world)
        else:
            cuts = balanced_cuts(tile_weights(plan, phase_index, tiles), world)
            lo, hi = cuts[rank], cuts[rank + 1]
        return tiles[lo:hi]

    return keep


def owned_runs(plan, tables, primary):
    """Contiguous runs [(start, stop)] of the packed output layout ``[grad | g | J | H]`` written by the tiles
    in ``tables`` (one rank's shard), derived from the same tables the kernels consume.  ``primary`` adds
    the boundary-node / system-level scalars (computed by rank 0 only)."""
    off = {"grad": 0, "g": plan.n, "J": plan.n + plan.m, "H": plan.n + plan.m + plan.nnz_J}
    runs = []
    tiles = tables.tiles
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        mine = tiles[(tiles["phase"] == k) & (tiles["nj"] > 0)]
        if len(mine) == 0:
            continue
        first, last = mine[0], mine[-1]
        stride = int(lay.stride[int(last["j0"])])
        R = stride
        i_lo, i_hi = int(first["offI"]), int(last["offI"]) + int(last["nj"]) * int(last["nnzI"])
        t_lo, t_hi = int(first["offT"]), int(last["offT"]) + int(last["nj"]) * int(last["nnzT"])
        r_lo, r_hi = int(first["r0"]), int(last["r0"]) + int(last["nj"]) * R
        q_lo = int(first["q0"])
        nq = int(last["nj"]) * stride + (1 if lay.scheme == "lgl" else 0)
        q_hi = int(last["q0"]) + (nq - 1 if (lay.scheme == "lgl" and not last["last"]) else nq)
        m_lo, m_hi = max(q_lo, lay.mid_lo), min(q_hi, lay.mid_hi)          # owned middle nodes
        for cbname, key in (("jac", "J"), ("hess", "H")):
            for sg in getattr(plan, cbname).segs[k]:
                if sg.kind == "I":
                    runs.append((off[key] + sg.base + i_lo, off[key] + sg.base + i_hi))
                elif m_hi > m_lo:
                    runs.append((off[key] + sg.base + m_lo - lay.mid_lo, off[key] + sg.base + m_hi - lay.mid_lo))
        for base in plan.jac.tconst[k]:
            runs.append((off["J"] + base + t_lo, off["J"] + base + t_hi))
        for i in range(pp.nx):
            g0 = off["g"] + plan.g_off[k] + i * lay.L_d
            runs.append((g0 + r_lo, g0 + r_hi))
            v0 = off["grad"] + plan.l_p[k] + int(lay.l_v[i])
            runs.append((v0 + q_lo, v0 + q_hi))
        for i in range(pp.nu):
            v0 = off["grad"] + plan.l_p[k] + int(lay.l_v[pp.nx + i])
            runs.append((v0 + q_lo, v0 + q_hi))
        for j in range(pp.phase.n_c):
            p0 = off["g"] + plan.path_off[k] + j * lay.L_m
            runs.append((p0 + q_lo, p0 + q_hi))
    if primary:
        single = [off["J"] + it.pos for it in plan.jac.items] + [off["H"] + it.pos for it in plan.hess.items]
        single += [off["H"] + b.pos + t for b in plan.outer for t in range(b.count)]
        single += [off["g"] + c for c in range(plan.n_sys)]
        runs += [(p, p + 1) for p in single]
    return [(a, b) for a, b in runs if b > a]


def owned_aux_runs(plan, tables, primary):
    """Contiguous runs [(start, stop)] of the AUXILIARY buffer (the quadrature-weighted gradient entries of the integrals, one per
    middle node, that the outer-product Hessian blocks of a model nonlinear in the integrals are formed from; plan.aux) filled
    by the tiles in ``tables``; ``primary`` adds the boundary / system-level scalars (rank 0's)."""
    runs = []
    tiles = tables.tiles
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        mine = tiles[(tiles["phase"] == k) & (tiles["nj"] > 0)]
        if len(mine) == 0:
            continue
        first, last = mine[0], mine[-1]
        stride = int(lay.stride[int(last["j0"])])
        q_lo = int(first["q0"])
        nq = int(last["nj"]) * stride + (1 if lay.scheme == "lgl" else 0)
        q_hi = int(last["q0"]) + (nq - 1 if (lay.scheme == "lgl" and not last["last"]) else nq)
        m_lo, m_hi = max(q_lo, lay.mid_lo), min(q_hi, lay.mid_hi)
        if m_hi > m_lo:
            for sg in plan.aux.segs[k]:
                runs.append((sg.base + m_lo - lay.mid_lo, sg.base + m_hi - lay.mid_lo))
    if primary:
        runs += [(it.pos, it.pos + 1) for it in plan.aux.items]
    return [(a, b) for a, b in runs if b > a]


def needed_x_runs(plan, tables, primary):
    """Contiguous runs [(start, stop)] of the NLP vector x that the tiles in ``tables`` (one rank's shard) read: per phase
    and variable the nodes of its tiles (plus the LGR end slot behind the last one), the phase's t0 / tf and the static
    parameters; ``primary`` adds the boundary nodes of every phase (the boundary / system-level work of rank 0).  A rank
    uploads these over its own PCIe link instead of all of x: with N ranks every link carries 1 / N of x, not N copies."""
    runs = []
    tiles = tables.tiles
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        base = int(plan.l_p[k])
        runs.append((base + lay.L - 2, base + lay.L))                      # t0, tf
        spans = []
        mine = tiles[(tiles["phase"] == k) & (tiles["nj"] > 0)]
        if len(mine):
            first, last = mine[0], mine[-1]
            stride = int(lay.stride[int(last["j0"])])
            spans.append((int(first["q0"]), int(last["q0"]) + int(last["nj"]) * stride + 1))
        if primary:
            spans += [(0, 1), (lay.L_m - 1, lay.L_m + 1)]
        for i in range(pp.nx + pp.nu):
            length = lay.state_len if i < pp.nx else lay.L_m
            for lo, hi in spans:
                lo, hi = max(lo, 0), min(hi + 1, length)                   # (+1: a slot more never hurts)
                if hi > lo:
                    runs.append((base + int(lay.l_v[i]) + lo, base + int(lay.l_v[i]) + hi))
    if plan.n_s:
        runs.append((plan.l_s, plan.r_s))
    runs.sort()
    merged = []
    for a, b in runs:
        if merged and a <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    return merged


def shared_gradient_slots(plan):
    """Gradient entries that are sums over all nodes (t0/tf of every phase, static parameters) plus the
    slots no node writes (LGR state end points): every rank's kernels write their own partial there."""
    slots = []
    for k, pp in enumerate(plan.phase_plans):
        lay = pp.layout
        if lay.scheme == "lgr":
            slots += [plan.l_p[k] + int(lay.l_v[i]) + lay.L_m for i in range(pp.nx)]
        slots += [plan.l_p[k] + lay.L - 2, plan.l_p[k] + lay.L - 1]
    slots += list(range(plan.l_s, plan.r_s))
    return np.array(sorted(set(int(v) for v in slots)), dtype=np.int64)


CHUNK = 4096       # doubles per chunk of a run table (one workgroup of pk_runs per chunk: 6 MB of runs = 190 workgroups)


def run_table(runs, packed_first=0):
    """(runs as (start, stop) in the full layout) -> int64 array [chunks, 3] of (full offset, packed offset, length):
    the runs laid one after the other in a pack, cut into chunks of at most CHUNK doubles."""
    rows, pos = [], packed_first
    for a, b in runs:
        for c in range(a, b, CHUNK):
            ln = min(CHUNK, b - c)
            rows.append((c, pos, ln))
            pos += ln
    return np.array(rows, dtype=np.int64).reshape(-1, 3), pos - packed_first


class RunCopier:
    """dst[dst_off + i] = src[src_off + i] over a run table.  Device tensors: ONE launch of the HIP kernel pk_runs
    through the C ABI (``ctx`` = the rank's runtime.Context; 16 bytes of traffic per double -- the int64 index tensors of
    ``index_select`` / ``index_copy_`` it replaces cost 3x that and put torch kernels on the hot path).  CPU tensors
    (the gloo tests of the exchange logic): slice copies."""

    def __init__(self, torch, table, device, ctx=None, swap=False):
        t = np.ascontiguousarray(table[:, [1, 0, 2]] if swap else table, dtype=np.int64)
        self.rows = [(int(a), int(b), int(n)) for a, b, n in t]
        self.ctx, self.n = ctx, len(t)
        self.dev_table = torch.from_numpy(t.copy()).to(device) if device.type == "cuda" else None

    def __call__(self, src, dst, stream=None):
        if self.n == 0:
            return
        if self.dev_table is not None:
            if self.ctx is None:
                raise RuntimeError("RunCopier on device tensors needs the rank's HIP context (no CPU path)")
            st = C.c_void_p(stream) if stream else None
            self.ctx.check(self.ctx.lib.pk_copy_runs_dev(self.ctx.handle, C.c_void_p(self.dev_table.data_ptr()), self.n,
                                                         C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), st))
            return
        for so, do, n in self.rows:
            dst[do: do + n] = src[so: so + n]


class Reassembler:
    """RCCL reassembly of the packed outputs ``[grad | g | J | H]`` on one or on every rank (the A/B forms of the
    exchange; device agnostic: CUDA tensors with RCCL on the GPUs, CPU tensors with gloo in the tests).  A rank's owned
    positions are a few dozen contiguous runs: packing and unpacking are run copies (``RunCopier``), no index tensors."""

    def __init__(self, torch, plan, runs_per_rank, rank, world, device, ctx=None):
        self.torch, self.rank, self.world = torch, rank, world
        self.total = plan.n + plan.m + plan.nnz_J + plan.nnz_H
        shared = shared_gradient_slots(plan)
        cover = np.zeros(self.total + 1, dtype=np.int64)
        for runs in runs_per_rank:
            for a, b in runs:
                cover[a] += 1
                cover[b] -= 1
        covered = np.cumsum(cover)[:-1]
        covered[shared] += 1
        if not np.all(covered == 1):
            raise RuntimeError("internal error: the shards do not partition the output positions")
        tables = [run_table(runs) for runs in runs_per_rank]
        self.pad = max(n for _, n in tables)
        self.n_shared = len(shared)
        self.shared_runs = [(int(i), int(i) + 1) for i in shared]
        self.device, self.ctx = device, ctx
        self.pack = RunCopier(torch, tables[rank][0], device, ctx)                       # full -> my pack
        # every rank's pack (row r of the receive buffer starts at r * row) -> full
        self._tables = [t for t, _ in tables]
        self.unpack_all = self._unpacker(self.pad)
        sh_t, _ = run_table(self.shared_runs)
        self.pack_shared = RunCopier(torch, sh_t, device, ctx)
        self.unpack_shared = RunCopier(torch, sh_t, device, ctx, swap=True)
        self.recv = torch.empty(self.pad * world, dtype=torch.float64, device=device)
        self.send = torch.zeros(self.pad, dtype=torch.float64, device=device)

    def _unpacker(self, row, skip=None):
        rows = []
        for r, t in enumerate(self._tables):
            if r == skip or len(t) == 0:
                continue
            u = t.copy()
            u[:, 1] += r * row
            rows.append(u)
        tab = np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)
        return RunCopier(self.torch, tab, self.device, self.ctx, swap=True)

    def exchange(self, full, small, dist, root=None, stream=None):
        """``full``: packed buffer of length total + 1; ``small``: the small reduction buffer [integrals | shared
        gradient slots] (integrals already filled with this rank's partials).  ``root = None``: tiny all-reduce + ONE
        all-gather, afterwards ``full`` and ``small`` are complete on every rank.  ``root = r``: ONE gather to rank r;
        the partial sums travel at the end of every pack and are added on rank r in rank order; the other ranks keep
        their own slices and partial sums."""
        n_sh = self.n_shared
        if n_sh:
            self.pack_shared(full, small[small.numel() - n_sh:], stream)
        if root is None:
            dist.all_reduce(small)
            self.pack(full, self.send, stream)
            dist.all_gather_into_tensor(self.recv, self.send)
            self.unpack_all(self.recv, full, stream)
        else:
            self._gather_to(full, small, dist, root, stream)
            if self.rank != root:
                return
        if n_sh:
            self.unpack_shared(small[small.numel() - n_sh:], full, stream)

    def _gather_to(self, full, small, dist, root, stream):
        torch, n_sm, row = self.torch, small.numel(), self.pad + small.numel()
        if getattr(self, "_send", None) is None or self._send.numel() != row:
            self._send = torch.zeros(row, dtype=full.dtype, device=full.device)
            if self.rank == root:
                self._rows = torch.empty(self.world * row, dtype=full.dtype, device=full.device)
                self._unpack_rows = self._unpacker(row)
        self.pack(full, self._send, stream)
        self._send[self.pad:] = small
        if self.rank != root:
            dist.gather(self._send, None, dst=root)
            return
        dist.gather(self._send, [self._rows[r * row: (r + 1) * row] for r in range(self.world)], dst=root)
        self._unpack_rows(self._rows, full, stream)
        small.copy_(self._rows[self.pad: row])
        for r in range(1, self.world):                       # fixed order: reproducible sums
            small.add_(self._rows[r * row + self.pad: (r + 1) * row])


class _DeviceArray:
    """A float64 device allocation of the C ABI (pk_device_alloc / pk_ipc_open) as a torch tensor, without a copy."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class PeerMailboxes:
    """The peer-mapped mailboxes of pk_xchg (the exchange of the partial sums inside ONE launch per rank, no collective):
    every rank allocates its mailbox in fine-grained device memory, exports it (hipIpc) and maps everybody else's."""

    def __init__(self, torch, ev, plan, rank, world, dist, device):
        import os

        lib, h, chk = ev.ctx.lib, ev.ctx.handle, ev.ctx.check
        self.ev, self.rank, self.world = ev, rank, world
        shared = shared_gradient_slots(plan)
        n_small = max(len(plan.I_syms), 1) + len(shared)
        self.stride = -(-(1 + n_small) // 16) * 16
        words = 2 * world * self.stride + 16          # (+ PK_XC_STATE: the rank's cycle count and time-out count)
        self.own, self.mapped = C.c_void_p(), []
        # Every rank walks through ALL collective calls below whatever fails locally (a rank that raised early would
        # leave the others waiting); the local failures are gathered and every rank takes the same decision.
        problem, raw = None, None
        try:
            chk(lib.pk_device_alloc(h, 8 * words, 1, C.byref(self.own)))
            handle = C.create_string_buffer(64)
            chk(lib.pk_ipc_export(h, self.own, handle))
            raw = handle.raw
        except Exception as exc:  # noqa: BLE001
            problem = f"rank {rank}: {exc!r}"
        got = [None] * world
        dist.all_gather_object(got, (rank, raw, os.getpid(), problem))
        ptrs = []
        if all(g[3] is None for g in got):
            try:
                for r, peer_raw, _pid, _ in sorted(got):
                    if r == rank:
                        ptrs.append(self.own.value)
                        continue
                    p = C.c_void_p()
                    chk(lib.pk_ipc_open(h, C.create_string_buffer(peer_raw, 64), C.byref(p)))
                    self.mapped.append(p)
                    ptrs.append(p.value)
            except Exception as exc:  # noqa: BLE001
                problem = f"rank {rank}: {exc!r}"
        verdicts = [None] * world
        dist.all_gather_object(verdicts, problem if problem else next((g[3] for g in got if g[3]), None))
        failed = [v for v in verdicts if v]
        if failed:
            self.close()
            raise RuntimeError("peer-mapped mailboxes are not available: " + "; ".join(sorted(set(failed))))
        self.table = torch.tensor(ptrs, dtype=torch.int64, device=device)
        self.idx = torch.tensor(np.asarray(shared, dtype=np.int32), dtype=torch.int32, device=device)
        chk(lib.pk_set_exchange(h, world, rank, C.c_void_p(self.table.data_ptr()), C.c_void_p(self.idx.data_ptr()),
                                len(shared), self.stride))
        dist.barrier()                     # every rank has mapped every mailbox before the first flag is raised

    def status(self, stream=None):
        """(cycles exchanged so far, exchanges that gave up waiting for a peer) of this rank; synchronizes the stream."""
        cyc, late = C.c_int64(), C.c_int64()
        self.ev.ctx.check(self.ev.ctx.lib.pk_exchange_status(self.ev.ctx.handle, stream, C.byref(cyc), C.byref(late)))
        return cyc.value, late.value

    def close(self):
        lib, h = self.ev.ctx.lib, self.ev.ctx.handle
        if h:
            for p in self.mapped:
                lib.pk_ipc_close(h, p)
            if self.own:
                lib.pk_device_free(h, self.own)
        self.mapped, self.own = [], C.c_void_p()


class HostStagedCollectives:
    """The collectives of the sharded cycle on CUDA tensors through a CPU (gloo) process group.  RCCL refuses two ranks
    on one device; this adapter lets the N > 1 code path run with several ranks on ONE GPU (tests, rehearsals of
    ``bench.py --gpus N``) -- it is not a measurement path."""

    def __init__(self, dist):
        self.dist = dist

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self.dist.all_reduce(h) if op is None else self.dist.all_reduce(h, op=op)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp):
        ho = out.cpu()
        self.dist.all_gather_into_tensor(ho, inp.cpu())
        out.copy_(ho)

    def gather(self, inp, gather_list, dst):
        hl = [t.cpu() for t in gather_list] if gather_list is not None else None
        self.dist.gather(inp.cpu(), hl, dst=dst)
        for t, h in zip(gather_list or [], hl or []):
            t.copy_(h)

    def __getattr__(self, name):          # barrier, ReduceOp, destroy_process_group, ...
        return getattr(self.dist, name)


class ShardedEvaluator:
    """Rank-local evaluator of one shard + what couples the shards.  ``dist`` is an initialised torch.distributed module.

    Forms of the exchange after the shard's ONE launch (pk_cycle on its tiles), ``exchange=``:

    * ``"sums"``   (needs ``enable_peer_exchange``) -- no reassembly: every rank leaves its slices of grad f / g / J / H
      in its own HBM at the reference positions; only the sums over all nodes (integrals -> f, the gradient entries of
      t0 / tf / static parameters) are exchanged, through peer-mapped mailboxes inside ONE one-workgroup launch
      (pk_xchg): no collective call, no host round trip.  The weak-scaling form: J / H stay where a device-side KKT
      solve or the per-GPU copy to the host wants them.
    * ``"direct"`` (needs ``enable_peer_exchange(root=r)``) -- reassembly on rank r's GPU without pack, collective or
      unpack: the other ranks' kernels store their slices straight into rank r's buffer through a peer mapping (xGMI),
      pk_xchg's flags tell rank r when every peer's launch has finished.
    * ``"gather"`` / ``"allgather"`` -- RCCL reassembly on one / on every rank (``Reassembler``): run-copy pack, ONE
      collective, run-copy unpack.  (``cycle(..., root=r)`` / ``root=None`` select these when ``exchange`` is not given.)
    """

    def __init__(self, plan, rank, world, device=0, intervals_per_wave=None, reassemble_single=False):
        """``reassemble_single``: with ONE rank, still run the "gather" / "allgather" forms through the ``Reassembler`` (pack,
        collective over a process group of one, unpack) -- RCCL accepts one rank per device, so the stream ordering between
        this evaluator's launches and RCCL's kernels can be exercised on a single GPU (tests/test_gpu_rccl_single.py)."""
        import torch

        from .codegen import ModelSource
        from .evaluator import Evaluator, Tables

        self.torch, self.rank, self.world, self.plan = torch, rank, world, plan
        # (intervals_per_wave None: the evaluator sizes the tiling for ONE shard's share of the mesh, from output_share)
        self.ev = Evaluator(plan, device=device, intervals_per_wave=intervals_per_wave,
                            tile_filter=tile_filter(rank, world, plan) if world > 1 else None, sharded=world > 1,
                            output_share=1.0 / max(world, 1), host_helpers=False)
        self.dev = dev = torch.device("cuda", device)
        n_I = max(len(plan.I_syms), 1)
        # integrals that later kernels need (models nonlinear in I) must be global *before* those kernels
        self.early_I = bool(plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I)
        self.sizes = [("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H)]
        self.total = sum(n for _, n in self.sizes)
        self.full = torch.zeros(self.total + 1, dtype=torch.float64, device=dev)
        self.out = self._views(self.full)
        self.out["f"] = torch.zeros(1, dtype=torch.float64, device=dev)
        self.re, self.runs = None, None
        if world > 1 or reassemble_single:
            ipw = self.ev.tables.intervals_per_wave
            src = self.ev.src
            self.runs = [owned_runs(plan, Tables(plan, src, ipw, tile_filter(r, world, plan) if world > 1 else None), r == 0)
                         for r in range(world)]
            self.re = Reassembler(torch, plan, self.runs, rank, world, dev, ctx=self.ev.ctx)
        n_sh = self.re.n_shared if self.re else 0
        self.small = torch.zeros(n_I + n_sh, dtype=torch.float64, device=dev)
        self.I = self.small[:n_I]
        lib, h = self.ev.ctx.lib, self.ev.ctx.handle
        self.ev.ctx.check(lib.pk_set_shard(h, int(rank != 0), 1, C.c_void_p(self.I.data_ptr())))
        # objectives / system constraints nonlinear in the integrals (outer-product Hessian blocks, easyderiv.py:323-459):
        # every shard fills the auxiliary entries of its own nodes, the sum over the ranks goes to the primary rank's blocks
        self.aux_own = self.aux_sum = None
        if plan.outer:
            p_aux, n_aux = C.c_void_p(), C.c_int64()
            self.ev.ctx.check(lib.pk_aux_buffer(h, C.byref(p_aux), C.byref(n_aux)))
            self.aux_own = torch.as_tensor(_DeviceArray(p_aux.value, n_aux.value), device=dev)
            self.aux_sum = torch.zeros(n_aux.value, dtype=torch.float64, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.peers, self.root, self._root_alloc, self._root_map, self.target = None, None, None, None, None
        self.inline_exchange = True      # the sums over the ranks are exchanged inside pk_cycle's launch (False: pk_xchg behind it)

    def _views(self, full):
        out, off = {}, 0
        for name, n in self.sizes:
            out[name] = full[off: off + n]
            off += n
        return out

    # ------------------------------------------------------------------ peer-mapped exchange (no collective)
    def enable_peer_exchange(self, dist, root=None):
        """Set up the mailboxes of the partial-sum exchange (``exchange="sums"``); with ``root = r`` also the peer
        mapping of rank r's output buffer (``exchange="direct"``: the other ranks' kernels store into it)."""
        torch, lib, h, chk = self.torch, self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check
        if self.world == 1:
            return
        if self.early_I:
            raise NotImplementedError("models nonlinear in the integrals need the integrals before the other kernels: "
                                      "they use the all-reduce path")
        if self.peers is None:
            self.peers = PeerMailboxes(torch, self.ev, self.plan, self.rank, self.world, dist, self.dev)
        if root is not None and self.root is None:
            self.root = int(root)
            count = self.total + 1
            handle = C.create_string_buffer(64)
            problem, base = None, None
            if self.rank == root:
                try:
                    # fine-grained: other GPUs write into it while this GPU's caches may hold neighbouring lines
                    self._root_alloc = C.c_void_p()
                    chk(lib.pk_device_alloc(h, 8 * count, 1, C.byref(self._root_alloc)))
                    chk(lib.pk_ipc_export(h, self._root_alloc, handle))
                    base = self._root_alloc.value
                except Exception as exc:  # noqa: BLE001 -- every rank still walks through the collectives below
                    problem = f"rank {self.rank}: {exc!r}"
            box = [(handle.raw, problem) if self.rank == root else None]
            dist.broadcast_object_list(box, src=root)
            if self.rank != root and box[0][1] is None:
                try:
                    self._root_map = C.c_void_p()
                    chk(lib.pk_ipc_open(h, C.create_string_buffer(box[0][0], 64), C.byref(self._root_map)))
                    base = self._root_map.value
                except Exception as exc:  # noqa: BLE001
                    self._root_map, problem = None, f"rank {self.rank}: {exc!r}"
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, problem or box[0][1])
            failed = sorted(set(v for v in verdicts if v))
            if failed:
                self.root = None
                raise RuntimeError("the peer mapping of the root's output buffer is not available: " + "; ".join(failed))
            if self.rank == root:
                self.full = torch.as_tensor(_DeviceArray(base, count), device=self.dev)
                f = self.out["f"]
                self.out = self._views(self.full)
                self.out["f"] = f
            self.target, off = {}, 0
            for name, n in self.sizes:                         # where this rank's kernels store in "direct" mode
                self.target[name] = base + 8 * off
                off += n
            dist.barrier()

    def close(self):
        lib, h = self.ev.ctx.lib, self.ev.ctx.handle
        if h:
            self.torch.cuda.synchronize()
            if self.peers is not None:
                self.peers.close()
            if self._root_map is not None:
                lib.pk_ipc_close(h, self._root_map)
            if self._root_alloc is not None:
                f = self.out["f"]
                self.full = self.torch.zeros(1, dtype=self.torch.float64, device=self.dev)   # drop the views first
                self.out = {"f": f}
                lib.pk_device_free(h, self._root_alloc)
        self.peers = self._root_map = self._root_alloc = None
        self.ev.close()

    def _inline(self, exchange):
        """Whether the sums are exchanged inside pk_cycle's launch.  Never for ``"direct"``: the finalize workgroup would
        raise the completion flags while the other workgroups of the same launch are still streaming their slices into the
        root's buffer over xGMI -- the root could read J / H before they have landed.  With pk_xchg as a launch of its own
        behind pk_cycle on the same stream, the kernel boundary orders every peer store before the flag."""
        return bool(self.inline_exchange) and exchange != "direct"

    def fast_step(self, x, lam, sigma, exchange="sums"):
        """A closure running one "sums" / "direct" cycle on this evaluator's stream with pre-built ctypes arguments and no
        torch call (bench.py: the loop is launch-bound, a cycle is two ~5 us launches)."""
        if self.peers is None or (exchange == "direct" and self.target is None):
            raise RuntimeError(f'exchange="{exchange}" needs enable_peer_exchange() first')
        lib, h, chk, o = self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check, self.out
        st = C.c_void_p(self.stream.cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        remote = exchange == "direct" and self.rank != self.root
        if exchange == "direct":
            tg = {k: C.c_void_p(v) for k, v in self.target.items()}
        else:
            tg = {k: ptr(o[k]) for k in ("grad", "g", "J", "H")}
        chk(lib.pk_set_shared_grad_target(h, ptr(o["grad"]) if remote else None))
        cyc = (h, C.c_void_p(x.data_ptr()), ptr(lam), C.c_double(float(sigma)), ptr(o["f"]), tg["grad"], tg["g"], tg["J"],
               tg["H"], st)
        px, pgrad, pf = C.c_void_p(x.data_ptr()), (ptr(o["grad"]) if remote else tg["grad"]), ptr(o["f"])
        cycle_fn, xchg_fn = lib.pk_eval_cycle_dev, lib.pk_exchange_sums_dev
        inline = self._inline(exchange)
        chk(lib.pk_set_exchange_inline(h, int(inline)))     # (stays set: the caller's loop owns the context)
        if inline:
            def step():
                rc = cycle_fn(*cyc)
                if rc:
                    chk(rc)
        else:
            def step():
                rc = cycle_fn(*cyc) or xchg_fn(h, px, pgrad, pf, 0, 1, st)
                if rc:
                    chk(rc)
        repeat_fn, two = lib.pk_eval_cycle_dev_repeat, 0 if inline else 1

        def many(count):          # `count` cycles enqueued by the library itself (no interpreter between the launches)
            rc = repeat_fn(*cyc, count, two, pgrad)
            if rc:
                chk(rc)

        step.many = many
        return step

    def cycle(self, x, lam, sigma, dist=None, root=None, exchange=None):
        """One f, grad f, g, J, H cycle on device tensors; results are left in ``self.out`` (reference order; complete
        on every rank, on rank ``root`` only, or -- ``exchange="sums"`` -- every rank's own slices plus the complete
        shared gradient slots, integrals and f).  Ordered after / before the work of torch's current stream."""
        torch = self.torch
        # Kernels and collectives are ordered on ONE stream of our own (torch's default stream has the null handle,
        # which the C ABI reads as "the context's stream" -- a stream torch's operations are not ordered with); the
        # caller's current stream is joined before and after.
        caller = torch.cuda.current_stream()
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            self._cycle_on_stream(x, lam, sigma, dist, root, exchange)
        caller.wait_stream(self.stream)
        return self.out

    def _cycle_on_stream(self, x, lam, sigma, dist, root, exchange):
        lib, h, chk = self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check
        st = C.c_void_p(self.stream.cuda_stream)
        o = self.out
        sharded = dist is not None and (self.world > 1 or self.re is not None)
        px = C.c_void_p(x.data_ptr())
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        if exchange is None:
            exchange = "allgather" if root is None else "gather"
        if sharded and self.world > 1 and exchange in ("sums", "direct"):
            if self.peers is None or (exchange == "direct" and self.target is None):
                raise RuntimeError(f'exchange="{exchange}" needs enable_peer_exchange() first')
            remote = exchange == "direct" and self.rank != self.root
            if exchange == "direct":
                tg = {k: C.c_void_p(v) for k, v in self.target.items()}
                # a peer keeps its partial sums of the shared gradient slots in its own buffer
                chk(lib.pk_set_shared_grad_target(h, ptr(o["grad"]) if remote else None))
            else:
                tg = {k: ptr(o[k]) for k in ("grad", "g", "J", "H")}
                chk(lib.pk_set_shared_grad_target(h, None))
            inline = self._inline(exchange)
            chk(lib.pk_set_exchange_inline(h, int(inline)))
            chk(lib.pk_eval_cycle_dev(h, px, ptr(lam), float(sigma), ptr(o["f"]), tg["grad"], tg["g"], tg["J"], tg["H"], st))
            if not inline:
                chk(lib.pk_exchange_sums_dev(h, px, tg["grad"] if not remote else ptr(o["grad"]), ptr(o["f"]), 0, 1, st))
            chk(lib.pk_set_exchange_inline(h, 0))
            return o
        if not self.early_I:
            # ONE launch per rank (pk_cycle): this shard's tiles of all five outputs, its share of the integrals (-> self.I)
            # and of the shared gradient slots; f is recomputed from the reduced integrals below
            chk(lib.pk_set_shared_grad_target(h, None))
            chk(lib.pk_set_exchange_inline(h, 0))
            chk(lib.pk_eval_cycle_dev(h, px, ptr(lam), float(sigma), ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]), ptr(o["J"]),
                                      ptr(o["H"]), st))
        else:
            chk(lib.pk_eval_integrals_dev(h, px, st))                  # this shard's share of every integral
            if sharded:
                dist.all_reduce(self.I)
            # (grad f, g, J: one kernel each, or -- meshes with workgroup-wide intervals -- the fused x-kernel)
            chk(lib.pk_eval_xpart_dev(h, px, ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]), ptr(o["J"]), st))
            chk(lib.pk_eval_hess_dev(h, px, ptr(lam), float(sigma), ptr(o["H"]), st))
            if self.aux_sum is not None:
                self.aux_sum.copy_(self.aux_own)
                if sharded:
                    dist.all_reduce(self.aux_sum)          # (a sum with zeros: every entry is filled by exactly one rank)
                if self.rank == 0:
                    chk(lib.pk_eval_outer_dev(h, ptr(self.aux_sum), ptr(o["H"]), st))
        if sharded:
            root = None if exchange == "allgather" else root
            if self.early_I:           # integrals are already global: keep them out of the second reduction
                keep = self.I.clone()
                self.re.exchange(self.full, self.small, dist, root, self.stream.cuda_stream)
                self.I.copy_(keep)
            else:
                self.re.exchange(self.full, self.small, dist, root, self.stream.cuda_stream)
        if root is None or root == self.rank or not sharded:     # (gather mode: only the root holds the reduced integrals)
            chk(lib.pk_eval_f_from_integrals_dev(h, px, C.c_void_p(o["f"].data_ptr()), st))
        return o
