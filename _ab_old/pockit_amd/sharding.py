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


def tile_filter(rank: int, world: int):
    def keep(phase_index, tiles):
        lo, hi = contiguous_share(len(tiles), rank, world)
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


class Reassembler:
    """All-gather based reassembly of the packed outputs ``[grad | g | J | H]`` (device agnostic: CUDA
    tensors with RCCL on the GPUs, CPU tensors with gloo in the tests)."""

    def __init__(self, torch, plan, runs_per_rank, rank, world, device):
        self.torch, self.rank, self.world = torch, rank, world
        idx = [np.concatenate([np.arange(a, b, dtype=np.int64) for a, b in runs]) if runs else np.zeros(0, np.int64)
               for runs in runs_per_rank]
        self.total = plan.n + plan.m + plan.nnz_J + plan.nnz_H
        shared = shared_gradient_slots(plan)
        covered = np.bincount(np.concatenate(idx + [shared]), minlength=self.total)
        if not np.all(covered == 1):
            raise RuntimeError("internal error: the shards do not partition the output positions")
        self.pad = max(len(ix) for ix in idx)
        dummy = self.total                      # padded tail of a pack lands in one scratch element
        padded = [np.concatenate([ix, np.full(self.pad - len(ix), dummy, dtype=np.int64)]) for ix in idx]
        self.own_idx = torch.from_numpy(padded[rank]).to(device)
        self.all_idx = torch.from_numpy(np.concatenate(padded)).to(device)
        self.shared_idx = torch.from_numpy(shared).to(device)
        self.recv = torch.empty(self.pad * world, dtype=torch.float64, device=device)

    def exchange(self, full, small, dist, root=None):
        """``full``: packed buffer of length total + 1 (last element is scratch); ``small``: the small
        reduction buffer [integrals | shared gradient slots] (already filled with this rank's partials).
        One tiny all-reduce + one all-gather; afterwards ``full`` and ``small`` are complete on every rank.

        ``root = r``: ONE gather to rank r instead -- the triplets are reassembled where the (host-side) NLP solver
        runs.  Rank r receives (N - 1) packs over its N - 1 direct xGMI links at once, whereas the all-gather moves
        N (N - 1) packs through the same links; the small partial sums travel at the end of every pack and are added
        on rank r in rank order (no second, latency-bound collective).  The other ranks keep their own slices and
        their own partial sums."""
        n_sh = self.shared_idx.numel()
        if n_sh:
            small[-n_sh:] = full.index_select(0, self.shared_idx)
        if root is None:
            dist.all_reduce(small)
            send = full.index_select(0, self.own_idx)
            dist.all_gather_into_tensor(self.recv, send)
            full.index_copy_(0, self.all_idx, self.recv)
        else:
            self._gather_to(full, small, dist, root)
            if self.rank != root:
                return
        if n_sh:
            full.index_copy_(0, self.shared_idx, small[-n_sh:])

    def _gather_to(self, full, small, dist, root):
        torch, n_sm, row = self.torch, small.numel(), self.pad + small.numel()
        if getattr(self, "_send", None) is None or self._send.numel() != row:
            self._send = torch.empty(row, dtype=full.dtype, device=full.device)
            if self.rank == root:
                self._rows = torch.empty(self.world * row, dtype=full.dtype, device=full.device)
                tail = torch.full((n_sm,), self.total, dtype=torch.int64, device=full.device)   # -> the scratch element
                self._rows_idx = torch.cat([torch.cat([self.all_idx[r * self.pad: (r + 1) * self.pad], tail])
                                            for r in range(self.world)])
        torch.index_select(full, 0, self.own_idx, out=self._send[: self.pad])
        self._send[self.pad:] = small
        if self.rank != root:
            dist.gather(self._send, None, dst=root)
            return
        dist.gather(self._send, [self._rows[r * row: (r + 1) * row] for r in range(self.world)], dst=root)
        full.index_copy_(0, self._rows_idx, self._rows)
        small.copy_(self._rows[self.pad: row])
        for r in range(1, self.world):                       # fixed order: reproducible sums
            small.add_(self._rows[r * row + self.pad: (r + 1) * row])


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
    """Rank-local evaluator + collectives.  ``dist`` is an initialised torch.distributed module."""

    def __init__(self, plan, rank, world, device=0, intervals_per_wave=None):
        import torch

        from .codegen import ModelSource
        from .evaluator import Evaluator, Tables, _intervals_per_wave

        self.torch, self.rank, self.world, self.plan = torch, rank, world, plan
        if world > 1 and plan.outer:
            raise NotImplementedError("objectives / system constraints nonlinear in the integrals (outer-product "
                                      "Hessian blocks) are evaluated on one GPU only; they are O(n^2) and small")
        if intervals_per_wave is None:          # the tiling is sized for ONE shard's share of the mesh
            intervals_per_wave = _intervals_per_wave(plan, shards=world)
        self.ev = Evaluator(plan, device=device, intervals_per_wave=intervals_per_wave,
                            tile_filter=tile_filter(rank, world) if world > 1 else None)
        dev = torch.device("cuda", device)
        n_I = max(len(plan.I_syms), 1)
        # integrals that later kernels need (models nonlinear in I) must be global *before* those kernels
        self.early_I = bool(plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I)
        sizes = [("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H)]
        self.full = torch.zeros(sum(n for _, n in sizes) + 1, dtype=torch.float64, device=dev)
        self.out, off = {}, 0
        for name, n in sizes:
            self.out[name] = self.full[off: off + n]
            off += n
        self.out["f"] = torch.zeros(1, dtype=torch.float64, device=dev)
        self.re = None
        if world > 1:
            ipw = self.ev.tables.intervals_per_wave
            src = self.ev.src
            runs = [owned_runs(plan, Tables(plan, src, ipw, tile_filter(r, world)), r == 0) for r in range(world)]
            self.re = Reassembler(torch, plan, runs, rank, world, dev)
        n_sh = self.re.shared_idx.numel() if self.re else 0
        self.small = torch.zeros(n_I + n_sh, dtype=torch.float64, device=dev)
        self.I = self.small[:n_I]
        lib, h = self.ev.ctx.lib, self.ev.ctx.handle
        self.ev.ctx.check(lib.pk_set_shard(h, int(rank != 0), 1, C.c_void_p(self.I.data_ptr())))
        self.stream = torch.cuda.Stream(device=dev)

    def cycle(self, x, lam, sigma, dist=None, root=None):
        """One f, grad f, g, J, H cycle on device tensors; results (reference order, complete on
        every rank, or -- ``root = r`` -- on rank r only) are left in ``self.out``.  Ordered after / before the
        work of torch's current stream."""
        torch = self.torch
        # Kernels and collectives are ordered on ONE stream of our own (torch's default stream has the null handle,
        # which the C ABI reads as "the context's stream" -- a stream torch's operations are not ordered with); the
        # caller's current stream is joined before and after.
        caller = torch.cuda.current_stream()
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            self._cycle_on_stream(x, lam, sigma, dist, root)
        caller.wait_stream(self.stream)
        return self.out

    def _cycle_on_stream(self, x, lam, sigma, dist, root):
        torch = self.torch
        lib, h, chk = self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check
        st = C.c_void_p(self.stream.cuda_stream)
        o = self.out
        sharded = dist is not None and self.world > 1
        px = C.c_void_p(x.data_ptr())
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        if not self.early_I:
            # ONE launch per rank (pk_cycle): this shard's tiles of all five outputs, its share of the integrals (-> self.I)
            # and of the shared gradient slots; f is recomputed from the reduced integrals below
            chk(lib.pk_eval_cycle_dev(h, px, ptr(lam), float(sigma), ptr(o["f"]), ptr(o["grad"]), ptr(o["g"]), ptr(o["J"]),
                                      ptr(o["H"]), st))
        else:
            chk(lib.pk_eval_integrals_dev(h, px, st))                  # this shard's share of every integral
            if sharded:
                dist.all_reduce(self.I)
            chk(lib.pk_eval_grad_dev(h, px, ptr(o["grad"]), st))
            chk(lib.pk_eval_g_dev(h, px, ptr(o["g"]), st))
            chk(lib.pk_eval_jac_dev(h, px, ptr(o["J"]), st))
            chk(lib.pk_eval_hess_dev(h, px, ptr(lam), float(sigma), ptr(o["H"]), st))
        if sharded:
            if self.early_I:           # integrals are already global: keep them out of the second reduction
                keep = self.I.clone()
                self.re.exchange(self.full, self.small, dist, root)
                self.I.copy_(keep)
            else:
                self.re.exchange(self.full, self.small, dist, root)
        if root is None or root == self.rank or not sharded:     # (gather mode: only the root holds the reduced integrals)
            chk(lib.pk_eval_f_from_integrals_dev(h, px, C.c_void_p(o["f"].data_ptr()), st))
        return o
