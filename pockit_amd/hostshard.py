"""Host-landed sharded cycle: N processes (one per GPU) serve the five NLP callbacks of ONE system, every GPU landing its
slices of grad f / g / J / H in one pinned host array over its OWN PCIe link.

Why: with a host-side solver the cycle is bound by the PCIe link, not by the kernels (12 MB per
cycle at 12k nodes, 121 MB at 40k nodes against 5 / 22 us of compute).  Reassembling the triplets on one GPU first would
add an xGMI hop and still leave ONE link to the host; here rank r copies its owned runs (a few dozen contiguous runs,
``sharding.owned_runs``) straight into the solver's arrays, N links in parallel ("each GPU D2H's
its own slices straight into the pinned host array at precomputed offsets (no collective)").

Mechanics.  Rank 0 is the solver's process (``HostShardedEvaluator`` with the callback methods); the other ranks call
``serve()``.  One POSIX shared-memory segment holds a control block, x, lambda, sigma, every rank's partial sums and the
packed outputs ``[grad | g | J | H]``; every process page-locks it for its GPU (``pk_host_register``).  A callback on a
new x writes x, bumps the sequence number and evaluates rank 0's own shard; every other rank sees the number, uploads x
from the segment over its link, evaluates its shard (``pk_eval_xpart_dev``: the fused x-kernel on its tiles), lets the
run-copy kernel (``pk_runs``) store its owned runs into the segment and publishes its partial sums [integrals | shared
gradient slots]; rank 0 adds them in rank order and evaluates f = F_o(I, s) (systembase.py:592-605) on the host.  The
Hessian works the same way with lambda and sigma.  No collective, no device-to-device traffic.

Only models whose system-level functions are linear in the integrals (none of BASELINE's configs is otherwise).
torch.distributed is used at set-up only (the segment's name).
"""
from __future__ import annotations

import ctypes as C
import time
from multiprocessing import shared_memory

import numpy as np
import sympy as sp

from .evaluator import Evaluator, _intervals_per_wave
from .sharding import owned_runs, run_table, shared_gradient_slots, tile_filter

CMD_EXIT, CMD_X, CMD_HESS = 0, 1, 2
CTRL_WORDS = 64          # control block: [0] sequence number, [1] command, [8 + r] last sequence rank r completed


class HostShardedEvaluator:
    """See the module docstring.  ``dist``: an initialised torch.distributed module (any backend; used once)."""

    def __init__(self, plan, rank, world, dist, device=0, intervals_per_wave=None, timeout_s=120.0):
        if plan.outer or plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I:
            raise NotImplementedError("the host-landed sharded cycle needs system functions that are linear in the integrals")
        self.plan, self.rank, self.world, self.timeout_s = plan, rank, world, float(timeout_s)
        if intervals_per_wave is None:
            intervals_per_wave = _intervals_per_wave(plan, shards=world)
        self.ev = Evaluator(plan, device=device, intervals_per_wave=intervals_per_wave,
                            tile_filter=tile_filter(rank, world, plan) if world > 1 else None)
        lib, h, chk = self.ev.ctx.lib, self.ev.ctx.handle, self.ev.ctx.check
        self.lib, self.h, self.chk = lib, h, chk
        n, m = plan.n, plan.m
        self.n_I = max(len(plan.I_syms), 1)
        self.shared = shared_gradient_slots(plan)
        self.n_small = self.n_I + len(self.shared)
        self.total = n + m + plan.nnz_J + plan.nnz_H
        # ---- layout of the segment, in 8-byte words
        off = {"ctrl": 0}
        off["sigma"] = CTRL_WORDS
        off["x"] = off["sigma"] + 8
        off["lam"] = off["x"] + n
        off["part"] = off["lam"] + m
        off["out"] = -(-(off["part"] + world * self.n_small) // 16) * 16       # 128-byte aligned
        off["f"] = off["out"] + self.total
        words = off["f"] + 8
        self.off = off
        name = [None]
        if rank == 0:
            self.shm = shared_memory.SharedMemory(create=True, size=8 * words)
            name[0] = self.shm.name
        if world > 1:
            dist.broadcast_object_list(name, src=0)
        if rank != 0:
            self.shm = shared_memory.SharedMemory(name=name[0])
            try:        # (Python < 3.13 registers an ATTACHED segment with the resource tracker too, which then tries to
                from multiprocessing import resource_tracker      # unlink it a second time at exit: rank 0 owns it)

                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:  # noqa: BLE001
                pass
        self.words = np.ndarray((words,), dtype=np.float64, buffer=self.shm.buf)
        self.ctrl = np.ndarray((CTRL_WORDS,), dtype=np.int64, buffer=self.shm.buf)
        if rank == 0:
            self.words[:] = 0.0
        if world > 1:
            dist.barrier()
        self._host_base = C.c_void_p(self.words.ctypes.data)
        self._dev_base = C.c_void_p()
        chk(lib.pk_host_register(h, self._host_base, 8 * words, C.byref(self._dev_base)))
        view = lambda key, cnt: self.words[off[key]: off[key] + cnt]  # noqa: E731
        self.h_x, self.h_lam, self.h_sigma = view("x", n), view("lam", m), view("sigma", 1)
        self.h_part = view("part", world * self.n_small).reshape(world, self.n_small)
        self.h_out, self.h_f = view("out", self.total), view("f", 1)
        self.out = {"grad": self.h_out[:n], "g": self.h_out[n: n + m], "J": self.h_out[n + m: n + m + plan.nnz_J],
                    "H": self.h_out[n + m + plan.nnz_J:]}
        self._dev = lambda key: C.c_void_p(self._dev_base.value + 8 * off[key])  # noqa: E731
        # ---- device side of this rank: a full-size local buffer [grad | g | J | H], x, lambda, the partial vector
        self._alloc = []

        def dalloc(count):
            p = C.c_void_p()
            chk(lib.pk_device_alloc(h, 8 * max(int(count), 1), 0, C.byref(p)))
            self._alloc.append(p)
            return p

        self.d_full, self.d_x, self.d_lam = dalloc(self.total + 1), dalloc(n), dalloc(m)
        self.d_small, self.d_f = dalloc(self.n_small), dalloc(1)
        at = lambda base, words_: C.c_void_p(base.value + 8 * int(words_))  # noqa: E731
        self.d_out = {"grad": self.d_full, "g": at(self.d_full, n), "J": at(self.d_full, n + m),
                      "H": at(self.d_full, n + m + plan.nnz_J)}
        chk(lib.pk_set_shard(h, int(rank != 0), 1, self.d_small))          # the integrals land at the head of d_small
        # ---- run tables: this rank's owned runs of the x-part and of H (same offsets in the local buffer and in the
        # segment's packed outputs), and the shared gradient slots -> tail of the partial vector
        runs = owned_runs(plan, self.ev.tables, rank == 0)
        split = n + m + plan.nnz_J
        rx = [(a, b) for a, b in runs if a < split]
        rh = [(a, b) for a, b in runs if a >= split]

        def table(rr, shift_dst=0):
            t, _ = run_table(rr)
            t = t.copy()
            t[:, 1] = t[:, 0] + shift_dst          # destination offset = position in the packed layout (+ shift)
            return self._upload_table(t)

        self.tab_x, self.tab_h = table(rx), table(rh)
        sh = np.zeros((len(self.shared), 3), dtype=np.int64)
        sh[:, 0], sh[:, 1], sh[:, 2] = self.shared, self.n_I + np.arange(len(self.shared)), 1
        self.tab_sh = self._upload_table(sh)                # local grad -> d_small[n_I + i]
        self._seq = 0
        self._x_seq = -1          # sequence number whose x-part results the segment holds
        # f = F_o(I, s) on the host (rank 0): the objective as a function of the integrals and static parameters
        if rank == 0:
            syms = list(plan.I_syms) + list(plan.s_syms)
            self._F_o = sp.lambdify(syms, sp.sympify(plan.system._expr_objective), modules="math") if syms else None
            self._F_const = float(sp.sympify(plan.system._expr_objective)) if not syms else None

    # ------------------------------------------------------------------ helpers
    def _upload_table(self, t):
        t = np.ascontiguousarray(t, dtype=np.int64)
        p = C.c_void_p()
        self.chk(self.lib.pk_device_alloc(self.h, max(t.nbytes, 8), 0, C.byref(p)))
        self._alloc.append(p)
        if len(t):
            self.chk(self.lib.pk_copy_dev(self.h, p, C.c_void_p(t.ctypes.data), t.nbytes, None))
            self.chk(self.lib.pk_sync(self.h, None))
        return (p, len(t))

    def _runs(self, tab, src, dst):
        p, cnt = tab
        if cnt:
            self.chk(self.lib.pk_copy_runs_dev(self.h, p, cnt, src, dst, None))

    def _do_x_part(self):
        lib, h, chk, n, m = self.lib, self.h, self.chk, self.plan.n, self.plan.m
        chk(lib.pk_copy_dev(h, self.d_x, self._dev("x"), 8 * n, None))                 # x over THIS rank's link
        # the fused x-kernel on this shard's tiles: its slices of grad f / g / J, its share of the integrals (-> d_small)
        # and its partial sums of the shared gradient slots (f is rank 0's to compute from the summed integrals)
        chk(lib.pk_eval_xpart_dev(h, self.d_x, self.d_f, self.d_out["grad"], self.d_out["g"], self.d_out["J"], None))
        self._runs(self.tab_x, self.d_full, self._dev("out"))                            # owned runs -> the solver's arrays
        self._runs(self.tab_sh, self.d_out["grad"], self.d_small)                        # shared slots behind the integrals
        chk(lib.pk_copy_dev(h, C.c_void_p(self._dev("part").value + 8 * self.rank * self.n_small), self.d_small,
                            8 * self.n_small, None))
        chk(lib.pk_sync(h, None))

    def _do_hess(self):
        lib, h, chk, m = self.lib, self.h, self.chk, self.plan.m
        chk(lib.pk_copy_dev(h, self.d_lam, self._dev("lam"), 8 * m, None))
        chk(lib.pk_eval_hess_dev(h, self.d_x, self.d_lam, float(self.h_sigma[0]), self.d_out["H"], None))
        self._runs(self.tab_h, self.d_full, self._dev("out"))
        chk(lib.pk_sync(h, None))

    def _wait_all(self, seq):
        t0 = time.perf_counter()
        done = self.ctrl[8: 8 + self.world]
        while True:
            if np.all(done[1:] >= seq):
                return
            if time.perf_counter() - t0 > self.timeout_s:
                raise RuntimeError(f"host-sharded cycle: ranks {np.nonzero(done < seq)[0].tolist()} did not finish sequence {seq}")

    # ------------------------------------------------------------------ the other ranks
    def serve(self):
        """Ranks != 0: evaluate this rank's shard whenever rank 0 posts a command; returns on CMD_EXIT (or when nothing
        arrives for ``timeout_s``)."""
        last, t_idle = 0, time.perf_counter()
        while True:
            seq = int(self.ctrl[0])
            if seq == last:
                if time.perf_counter() - t_idle > self.timeout_s:
                    return False
                continue
            cmd = int(self.ctrl[1])
            if cmd == CMD_EXIT:
                self.ctrl[8 + self.rank] = seq
                return True
            if cmd == CMD_X:
                self._do_x_part()
            elif cmd == CMD_HESS:
                self._do_hess()
            self.ctrl[8 + self.rank] = seq
            last, t_idle = seq, time.perf_counter()

    # ------------------------------------------------------------------ rank 0: the callbacks
    def _post(self, cmd):
        self._seq += 1
        self.ctrl[1] = cmd
        self.ctrl[0] = self._seq            # (x86: stores are not reordered with older stores)
        return self._seq

    def _prepare(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.plan.n,):
            raise ValueError(f"x must have shape ({self.plan.n},)")
        if self._x_seq >= 0 and np.array_equal(x, self.h_x):
            return
        self.h_x[:] = x
        seq = self._post(CMD_X)
        self._do_x_part()
        self._wait_all(seq)
        small = self.h_part[0].copy()
        for r in range(1, self.world):        # rank order: reproducible sums
            small += self.h_part[r]
        I = small[: self.n_I]
        s = x[self.plan.l_s: self.plan.r_s]
        args = [float(v) for v in I[: len(self.plan.I_syms)]] + [float(v) for v in s]
        self.h_f[0] = self._F_o(*args) if self._F_o is not None else self._F_const
        self.out["grad"][self.shared] = small[self.n_I:]
        self._x_seq = seq

    # (the structures complete the cyipopt ``problem_obj`` protocol: systembase.py:671-674, 811-818)
    def jacobianstructure(self):
        return self.plan.jac_row, self.plan.jac_col

    def hessianstructure(self):
        return self.plan.hess_row, self.plan.hess_col

    def objective(self, x):
        self._prepare(x)
        return np.float64(self.h_f[0])

    def gradient(self, x):
        self._prepare(x)
        return self.out["grad"]

    def constraints(self, x):
        self._prepare(x)
        return self.out["g"]

    def jacobian(self, x):
        self._prepare(x)
        return self.out["J"]

    def hessian(self, x, lagrange, obj_factor):
        lam = np.ascontiguousarray(lagrange, dtype=np.float64)
        if lam.shape != (self.plan.m,):
            raise ValueError(f"lagrange must have shape ({self.plan.m},)")
        self._prepare(x)
        self.h_lam[:] = lam
        self.h_sigma[0] = float(obj_factor)
        seq = self._post(CMD_HESS)
        self._do_hess()
        self._wait_all(seq)
        return self.out["H"]

    # ------------------------------------------------------------------ shutdown
    def close(self):
        if getattr(self, "shm", None) is None:
            return
        if self.rank == 0 and self.world > 1:
            seq = self._post(CMD_EXIT)
            try:
                self._wait_all(seq)
            except RuntimeError:
                pass
        lib, h = self.lib, self.h
        if h:
            lib.pk_sync(h, None)
            lib.pk_host_unregister(h, self._host_base)
            for p in self._alloc:
                lib.pk_device_free(h, p)
        self._alloc = []
        self.out = self.h_x = self.h_lam = self.h_sigma = self.h_part = self.h_out = self.h_f = None
        self.words = self.ctrl = None
        self.ev.close()
        try:
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass
        self.shm = None
