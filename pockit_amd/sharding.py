"""Mesh-interval sharding of one NLP across the GPUs of a node (one process per GPU).

Every defect row, path row, Jacobian and Hessian triplet belongs to exactly one mesh interval of
one phase (block-diagonal integration/translation matrices), so the tiles of every phase are
split into ``world`` contiguous ranges; rank r evaluates only its tiles, writing into full-size
output arrays at the reference positions (all other positions stay zero).  Reassembly is a sum
over ranks of arrays with disjoint support -- an RCCL all-reduce over xGMI issued through
``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  The objective's
integrals are partial sums per rank and are all-reduced before F_o(I, s) is evaluated.

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


class ShardedEvaluator:
    """Rank-local evaluator + collectives.  ``dist`` is an initialised torch.distributed module.

    All rank-local results live in ONE packed device buffer ``[I | grad | g | J | H]`` so that the
    reassembly is a single collective per cycle (xGMI collectives are latency-bound at these sizes:
    five ~MB-sized all-reduces cost several times one packed all-reduce)."""

    def __init__(self, plan, rank, world, device=0, intervals_per_wave=None):
        import torch

        from .evaluator import Evaluator

        self.torch, self.rank, self.world, self.plan = torch, rank, world, plan
        self.ev = Evaluator(plan, device=device, intervals_per_wave=intervals_per_wave,
                            tile_filter=tile_filter(rank, world) if world > 1 else None)
        dev = torch.device("cuda", device)
        n_I = max(len(plan.I_syms), 1)
        # integrals that later kernels need (models nonlinear in I) must be global *before* those kernels
        self.early_I = bool(plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I)
        sizes = [("I", n_I), ("grad", plan.n), ("g", plan.m), ("J", plan.nnz_J), ("H", plan.nnz_H)]
        self.pack = torch.zeros(sum(n for _, n in sizes), dtype=torch.float64, device=dev)
        self.out, off = {}, 0
        for name, n in sizes:
            self.out[name] = self.pack[off: off + n]
            off += n
        self.out["f"] = torch.zeros(1, dtype=torch.float64, device=dev)
        self.I = self.out["I"]
        lib, h = self.ev.ctx.lib, self.ev.ctx.handle
        self.ev.ctx.check(lib.pk_set_shard(h, int(rank != 0), 1, C.c_void_p(self.I.data_ptr())))

    def cycle(self, x, lam, sigma, dist=None):
        """One f, grad f, g, J, H cycle on device tensors; results (reference order, complete on
        every rank) are left in ``self.out``.  Enqueues on torch's current stream."""
        torch = self.torch
        lib, h, chk = self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        o = self.out
        sharded = dist is not None and self.world > 1
        if sharded:
            self.pack.zero_()
        px = C.c_void_p(x.data_ptr())
        chk(lib.pk_eval_integrals_dev(h, px, st))                      # this shard's share of every integral
        if sharded and self.early_I:
            dist.all_reduce(self.I)
        chk(lib.pk_eval_grad_dev(h, px, C.c_void_p(o["grad"].data_ptr()), st))
        chk(lib.pk_eval_g_dev(h, px, C.c_void_p(o["g"].data_ptr()), st))
        chk(lib.pk_eval_jac_dev(h, px, C.c_void_p(o["J"].data_ptr()), st))
        chk(lib.pk_eval_hess_dev(h, px, C.c_void_p(lam.data_ptr()), float(sigma), C.c_void_p(o["H"].data_ptr()), st))
        if sharded:
            if self.early_I:
                dist.all_reduce(self.pack[self.I.numel():])
            else:
                dist.all_reduce(self.pack)                             # integrals ride along
        chk(lib.pk_eval_f_from_integrals_dev(h, px, C.c_void_p(o["f"].data_ptr()), st))
        return o
