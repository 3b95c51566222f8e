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

Progress is reported by the GPUs themselves: behind its copies every rank's stream stores the sequence number into a mark
word of the segment (``pk_store_word_dev``), first when the partial sums and the grad f / g slices have landed, then when
the J slices have, and when the H slices have.  Rank 0 polls those words -- it never waits for another PROCESS to notice
that its GPU has finished -- and returns objective / gradient / constraints while J is still on the links (``jacobian``
waits for the second mark).  ``hessian`` launches on the prepared x before it compares x with it.

Models whose objective or system constraints are nonlinear in the integrals (easyderiv.py:323-459; none of BASELINE's configs)
take one more round trip per x: every rank's share of the integrals first (``CMD_INT``), rank 0 adds them up and hands the totals
back, then the x-part proper; the outer-product blocks of their Hessian are formed by rank 0's GPU from all ranks' auxiliary
entries, which travel through the segment too.
torch.distributed is used at set-up only (the segment's name).
"""
from __future__ import annotations

import ctypes as C
import os
import time
from multiprocessing import shared_memory

import numpy as np
import sympy as sp

from .evaluator import Evaluator
from .sharding import needed_x_runs, owned_runs, run_table, shared_gradient_slots, tile_filter

CMD_EXIT, CMD_X, CMD_HESS, CMD_X_TRIAL, CMD_J, CMD_INT = 0, 1, 2, 3, 4, 5
# (CMD_X_TRIAL: an x-part whose J slices stay on the GPUs until CMD_J; CMD_INT: models nonlinear in the integrals -- every rank's
#  share of the integrals first, rank 0 adds them up and hands the totals back before the x-part proper)
MAX_RANKS = 56
CTRL_WORDS = 384         # control block: [0] sequence number, [1] command, [2] pid of rank 0 (liveness), [8 + r] last sequence
                         # rank r's HOST loop has drained; progress marks written by rank r's GPU behind its copies (a one-word
                         # kernel, pk_store_word_dev): [MARK_EARLY + r] sequence whose partial sums, grad f and g slices have
MARK_EARLY, MARK_X, MARK_H = 64, 128, 192      # landed, [MARK_X + r] ... whose J slices have landed, [MARK_H + r] ... H slices
MARK_INT, MARK_AUX = 256, 320                  # (models nonlinear in the integrals: rank r's integrals / auxiliary entries are in)


class _AttachedSegment:
    """Rank 0's POSIX shared-memory segment mapped into another rank: plain shm_open + mmap.  (``SharedMemory(name=...)`` of
    Python < 3.13 registers an ATTACHED segment with the resource tracker as if this process owned it -- a second unlink at
    exit, or, with a tracker shared by spawned processes, rank 0's own registration removed.  Rank 0 owns the segment.)"""

    def __init__(self, name, size):
        import mmap

        try:                                   # (what SharedMemory itself calls; no resource-tracker registration)
            import _posixshmem

            fd = _posixshmem.shm_open("/" + name.lstrip("/"), os.O_RDWR, mode=0o600)
        except ImportError:
            fd = os.open("/dev/shm/" + name.lstrip("/"), os.O_RDWR)
        try:
            self._mmap = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self.name, self.buf = name, memoryview(self._mmap)

    def close(self):
        if self.buf is not None:
            self.buf.release()
            self.buf = None
            self._mmap.close()

    def unlink(self):          # (never: rank 0 does)
        pass


class HostShardedEvaluator:
    """See the module docstring.  ``dist``: an initialised torch.distributed module (any backend; used once)."""

    def __init__(self, plan, rank, world, dist, device=0, intervals_per_wave=None, timeout_s=120.0):
        # objective / system constraints nonlinear in the integrals (easyderiv.py:323-459): the integrals must be complete before
        # every other kernel, and the outer-product Hessian blocks are formed by rank 0 from all ranks' auxiliary entries
        self.needs_I = bool(plan.outer or plan.needs_I_grad or plan.needs_I_con or plan.jac.needs_I or plan.hess.needs_I)
        self.n_aux = int(plan.n_aux) if plan.outer else 0
        if world > MAX_RANKS:
            raise ValueError(f"at most {MAX_RANKS} ranks")
        self.plan, self.rank, self.world, self.timeout_s = plan, rank, world, float(timeout_s)
        self.zero_copy = False      # True: callbacks return live views of the shared segment (valid until the next iterate)
        self.ev = None
        # Rank-local work that can fail alone (code generation, hipcc, pk_load_model, pk_set_problem): every rank reports its
        # verdict BEFORE the first collective of the set-up, and all raise the same error if any failed -- a rank that raised
        # alone would leave the others in the broadcast / barrier below until the backend's time-out.
        problem = None
        try:
            self.ev = Evaluator(plan, device=device, intervals_per_wave=intervals_per_wave,
                                tile_filter=tile_filter(rank, world, plan) if world > 1 else None,
                                output_share=1.0 / max(world, 1), host_helpers=False)
        except Exception as exc:  # noqa: BLE001 -- reported to every rank below
            problem = f"rank {rank}: {exc!r}"
        if world > 1:
            verdicts = [None] * world
            dist.all_gather_object(verdicts, problem)
        else:
            verdicts = [problem]
        failed = sorted(v for v in verdicts if v)
        if failed:
            if self.ev is not None:
                self.ev.close()
                self.ev = None
            raise RuntimeError("HostShardedEvaluator: the rank-local set-up failed: " + "; ".join(failed))
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
        off["I"] = off["part"] + world * self.n_small            # (the summed integrals, handed back to every rank)
        off["aux"] = off["I"] + self.n_I                         # (all ranks' auxiliary entries, for rank 0's outer-product blocks)
        off["out"] = -(-(off["aux"] + self.n_aux) // 16) * 16    # 128-byte aligned
        off["f"] = off["out"] + self.total
        words = off["f"] + 8
        self.off = off
        name = [None]
        self.shm = None
        if rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=8 * words)
                name[0] = self.shm.name
            except Exception as exc:  # noqa: BLE001 -- the other ranks learn it from the broadcast (name None)
                name[0] = None
                self._shm_error = repr(exc)
        try:
            self._set_up(plan, rank, world, dist, name, words, off, n, m)
        except BaseException:
            # (a failure after the segment exists must not leave it behind in /dev/shm: it is held in memory)
            self._registered = getattr(self, "_registered", False)
            self.close()
            raise

    def _set_up(self, plan, rank, world, dist, name, words, off, n, m):
        lib, h, chk = self.lib, self.h, self.chk
        if world > 1:
            dist.broadcast_object_list(name, src=0)
        if name[0] is None:
            raise RuntimeError("HostShardedEvaluator: rank 0 could not create the shared segment"
                               + (f" ({self._shm_error})" if rank == 0 else ""))
        if rank != 0:
            self.shm = _AttachedSegment(name[0], 8 * words)
        self.words = np.ndarray((words,), dtype=np.float64, buffer=self.shm.buf)
        self.ctrl = np.ndarray((CTRL_WORDS,), dtype=np.int64, buffer=self.shm.buf)
        if rank == 0:
            self.words[:] = 0.0
            self.ctrl[2] = os.getpid()
        if world > 1:
            dist.barrier()
        self._host_base = C.c_void_p(self.words.ctypes.data)
        self._dev_base = C.c_void_p()
        self._registered = False
        chk(lib.pk_host_register(h, self._host_base, 8 * words, C.byref(self._dev_base)))
        self._registered = True
        view = lambda key, cnt: self.words[off[key]: off[key] + cnt]  # noqa: E731
        self.h_x, self.h_lam, self.h_sigma = view("x", n), view("lam", m), view("sigma", 1)
        self.h_part = view("part", world * self.n_small).reshape(world, self.n_small)
        self.h_I = view("I", self.n_I)
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

        # (grad f and g slices travel first: objective / gradient / constraints return while J is still on the link)
        r_early = [(a, b) for a, b in rx if a < n + m]
        rj = [(a, b) for a, b in rx if a >= n + m]
        self.tab_early, self.tab_j, self.tab_h = table(r_early), table(rj), table(rh)
        # the +-1 translation entries of J never change (phasebase.py:1071-1081): after this rank's first x-part they are in
        # the solver's array for good, and the per-iterate run copy leaves them out (19 % of J at 12k quadrotor nodes)
        const = [(split_j0 + a, split_j0 + b) for split_j0 in (n + m,) for a, b in plan.jac_constant_runs() if b - a >= 4096]

        def minus(runs, holes):
            out = []
            for a, b in runs:
                cur = a
                for ha, hb in holes:
                    if hb <= cur or ha >= b:
                        continue
                    if ha > cur:
                        out.append((cur, ha))
                    cur = max(cur, hb)
                if cur < b:
                    out.append((cur, b))
            return out

        # every rank's Hessian kernel writes only the entries of its own tiles (rank 0: also the boundary / system items), at
        # the reference positions: a share of at most 8 MiB is stored straight into the segment (as the single-GPU shim does)
        self.h_direct = 8 * sum(b - a for a, b in rh) <= (8 << 20)
        self._h_host = C.c_void_p(self._dev_base.value + 8 * (off["out"] + split))
        self.tab_j_changing = table(minus(rj, const)) if const else self.tab_j
        self._x_filled = False
        word = lambda i: C.c_void_p(self._dev_base.value + 8 * int(i))  # noqa: E731
        bases = (("early", MARK_EARLY), ("x", MARK_X), ("h", MARK_H), ("int", MARK_INT), ("aux", MARK_AUX))
        self._mark_dev = {k: word(base + rank) for k, base in bases}
        self._marks = {k: self.ctrl[base: base + world] for k, base in bases}
        if self.needs_I:
            self.tab_int = self._upload_table(np.array([[0, rank * self.n_small, self.n_I]], dtype=np.int64))   # d_small -> part
            self.tab_I_in = self._upload_table(np.array([[0, 0, self.n_I]], dtype=np.int64))                    # h_I -> d_small
        if self.n_aux:
            from .sharding import owned_aux_runs

            p_aux, cnt = C.c_void_p(), C.c_int64()
            chk(lib.pk_aux_buffer(h, C.byref(p_aux), C.byref(cnt)))
            self.d_aux = p_aux
            self.tab_aux = table(owned_aux_runs(plan, self.ev.tables, rank == 0))
        # what this rank reads of x: uploaded by a run copy over its own link (1 / N of x per link, not N copies of x)
        xr = needed_x_runs(plan, self.ev.tables, rank == 0)
        self.tab_xin = table(xr)
        self.x_upload_doubles = int(sum(b - a for a, b in xr))
        sh = np.zeros((len(self.shared), 3), dtype=np.int64)
        sh[:, 0], sh[:, 1], sh[:, 2] = self.shared, self.n_I + np.arange(len(self.shared)), 1
        self.tab_sh = self._upload_table(sh)                # local grad -> d_small[n_I + i]
        self.tab_part = self._upload_table(np.array([[0, rank * self.n_small, self.n_small]], dtype=np.int64))
        self._seq = 0
        self._x_seq = -1          # sequence number whose x-part results the segment holds
        self._j_asked, self._j_seq = True, -1      # (grad f / J asked for at the prepared iterate; sequence whose mark says J is in)
        # rank 0's own passes over x and lambda (staging them into the segment, the bitwise compare of every callback) grow
        # with the whole system, not with a rank's share: from 1 MB on they are cut into slices for a few helper threads
        self.helper_threads = self._start_helpers(world) if rank == 0 else 0
        # f = F_o(I, s) on the host (rank 0): the objective as a function of the integrals and static parameters
        if rank == 0:
            syms = list(plan.I_syms) + list(plan.s_syms)
            self._F_o = sp.lambdify(syms, sp.sympify(plan.system._expr_objective), modules="math") if syms else None
            self._F_const = float(sp.sympify(plan.system._expr_objective)) if not syms else None

    # ------------------------------------------------------------------ helpers
    def _start_helpers(self, world):
        """Helper threads of the library for rank 0's passes over x / lambda (pk_host_threads): as many as this rank's share
        of the host's cores allows (at most 6; POCKIT_AMD_HOST_THREADS=k overrides, 0 = none), kept only if a measured pass
        over x is at least a quarter faster with them (a container whose CPUs are time slices of one core gains nothing)."""
        n = self.plan.n
        env = os.environ.get("POCKIT_AMD_HOST_THREADS", "auto")
        if 8 * n < (1 << 20) or env == "0":
            return 0
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        k = int(env) if env.isdigit() else max(0, min(6, cores // max(world, 1) - 2))
        if k < 1:
            return 0
        lib = self.lib
        mine = np.array(self.h_x)
        a, b = mine.ctypes.data, self.h_x.ctypes.data

        def pass_us():
            lib.pk_same_bits(a, b, n)
            ts = []
            for _ in range(9):
                t = time.perf_counter()
                lib.pk_same_bits(a, b, n)
                ts.append(time.perf_counter() - t)
            return sorted(ts)[len(ts) // 2]

        from . import runtime

        # The pool is process-wide (pk_host_threads): one that another evaluator of this process started is shared, and never
        # stopped from here -- that evaluator may be in a pass on another thread right now.
        self._own_helpers = False
        running = runtime._HOST_HELPERS["k"]
        if running:
            return int(running)
        alone = pass_us()
        if lib.pk_host_threads(k):
            return 0
        helped = pass_us()
        if helped > 0.75 * alone:
            lib.pk_host_threads(0)
            return 0
        self._own_helpers = True
        runtime._HOST_HELPERS["k"] = k
        return k

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

    def _do_x_part(self, seq, with_j=True):
        """Enqueue this rank's share of an x-part; nothing here waits (the marks tell rank 0 what has landed).  ``with_j`` False:
        the J slices stay in device memory (a line search's trial point asks for f and g only; ``_do_j`` sends them if the
        point is accepted)."""
        lib, h, chk = self.lib, self.h, self.chk
        if self.needs_I:       # (x came up with CMD_INT; now the integrals summed over the ranks, where the kernels read them)
            self._runs(self.tab_I_in, self._dev("I"), self.d_small)
        else:
            self._runs(self.tab_xin, self._dev("x"), self.d_x)                           # this rank's part of x, over ITS link
        # the fused x-kernel on this shard's tiles: its slices of grad f / g / J, its share of the integrals (-> d_small)
        # and its partial sums of the shared gradient slots (f is rank 0's to compute from the summed integrals)
        chk(lib.pk_eval_xpart_dev(h, self.d_x, self.d_f, self.d_out["grad"], self.d_out["g"], self.d_out["J"], None))
        self._runs(self.tab_sh, self.d_out["grad"], self.d_small)                        # shared slots behind the integrals
        self._runs(self.tab_part, self.d_small, self._dev("part"))       # (kernel copies: a DMA behind the kernels costs a
        self._runs(self.tab_early, self.d_full, self._dev("out"))        #  cross-engine hand-off, ~10 us)
        chk(lib.pk_store_word_dev(h, self._mark_dev["early"], seq, None))
        if with_j:
            self._do_j(seq)

    def _do_int(self, seq):
        """Models nonlinear in the integrals, first half of an x-part: x up, this shard's share of every integral, into the segment."""
        self._runs(self.tab_xin, self._dev("x"), self.d_x)
        self.chk(self.lib.pk_eval_integrals_dev(self.h, self.d_x, None))
        self._runs(self.tab_int, self.d_small, self._dev("part"))
        self.chk(self.lib.pk_store_word_dev(self.h, self._mark_dev["int"], seq, None))

    def _do_j(self, seq):
        # owned runs of J -> the solver's array (the first time all of them, then only what changes with x)
        self._runs(self.tab_j_changing if self._x_filled else self.tab_j, self.d_full, self._dev("out"))
        self._x_filled = True
        self.chk(self.lib.pk_store_word_dev(self.h, self._mark_dev["x"], seq, None))

    def _do_hess(self, seq):
        lib, h, chk = self.lib, self.h, self.chk
        # the Hessian kernel reads the multipliers from the (page-locked) segment itself: every rank then moves only the rows
        # of ITS tiles over its link, and there is no upload in front of the kernel
        target = self._h_host if self.h_direct else self.d_out["H"]      # (a small share of H: the kernel's own stores go over
        chk(lib.pk_eval_hess_dev(h, self.d_x, self._dev("lam"), float(self.h_sigma[0]), target, None))      # the link, no copy)
        if self.n_aux:
            # outer-product blocks: every rank's auxiliary entries (of ITS nodes) into the segment; rank 0 forms the blocks from
            # all of them (the only place a rank waits for the others inside a command)
            self._runs(self.tab_aux, self.d_aux, self._dev("aux"))
            chk(lib.pk_store_word_dev(h, self._mark_dev["aux"], seq, None))
            if self.rank == 0:
                self._wait_marks("aux", seq)
                chk(lib.pk_eval_outer_dev(h, self._dev("aux"), target, None))
        if not self.h_direct:
            self._runs(self.tab_h, self.d_full, self._dev("out"))
        chk(lib.pk_store_word_dev(h, self._mark_dev["h"], seq, None))

    def _wait_marks(self, which, seq):
        """Rank 0 waits until every rank's GPU has stored the mark ``which`` of sequence ``seq`` (its own critical path: a
        spin, the clock looked at every 4096 polls)."""
        marks = self._marks[which]
        t0, polls = None, 0
        while True:
            if marks.min() >= seq:
                return
            polls += 1
            if polls & 0xFFF == 0:
                now = time.perf_counter()
                t0 = t0 or now
                if now - t0 > self.timeout_s:
                    raise RuntimeError(f"host-sharded cycle: ranks {np.nonzero(marks < seq)[0].tolist()} did not finish "
                                       f"'{which}' of sequence {seq} within {self.timeout_s:.0f} s")

    def _wait_all(self, seq):
        """Rank 0 waits until every other rank's host loop has drained ``seq`` (shutdown)."""
        done = self.ctrl[8: 8 + self.world]
        t0, polls = None, 0
        while True:
            if np.all(done[1:] >= seq):
                return
            polls += 1
            if polls & 0xFFF == 0:
                now = time.perf_counter()
                t0 = t0 or now
                if now - t0 > self.timeout_s:
                    raise RuntimeError(f"host-sharded cycle: ranks {np.nonzero(done < seq)[0].tolist()} did not finish "
                                       f"sequence {seq} within {self.timeout_s:.0f} s")

    # ------------------------------------------------------------------ the other ranks
    def _solver_alive(self):
        pid = int(self.ctrl[2])
        if pid <= 0:
            return True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        return True

    def serve(self):
        """Ranks != 0: evaluate this rank's shard whenever rank 0 posts a command; returns True on CMD_EXIT, False when
        the solver's process has gone away.  A long linear solve or line search on rank 0 between two callbacks does not
        strand the workers: there is no idle time-out, only the liveness of rank 0 (checked twice a second while idle).
        Polling backs off -- a busy spin for the first 2 ms after a command (the next callback of an iterate follows within
        microseconds), then 50 us sleeps, 1 ms sleeps after a second of silence."""
        last, t_last = 0, time.perf_counter()
        next_check = t_last + 0.5
        drained = True
        while True:
            seq = int(self.ctrl[0])
            if seq == last:
                idle = time.perf_counter() - t_last
                if idle > 2e-3:
                    if not drained:          # (off rank 0's critical path: it reads the GPU's own marks)
                        self.chk(self.lib.pk_wait_idle(self.h, None))
                        self.ctrl[8 + self.rank] = last
                        drained = True
                    time.sleep(1e-3 if idle > 1.0 else 5e-5)
                    now = time.perf_counter()
                    if now >= next_check:
                        next_check = now + 0.5
                        if not self._solver_alive():
                            return False
                continue
            cmd = int(self.ctrl[1])
            if cmd == CMD_EXIT:
                self.chk(self.lib.pk_wait_idle(self.h, None))
                self.ctrl[8 + self.rank] = seq
                return True
            if cmd == CMD_X:
                self._do_x_part(seq)
            elif cmd == CMD_X_TRIAL:
                self._do_x_part(seq, with_j=False)
            elif cmd == CMD_J:
                self._do_j(seq)
            elif cmd == CMD_INT:
                self._do_int(seq)
            elif cmd == CMD_HESS:
                self._do_hess(seq)
            drained = False
            last, t_last = seq, time.perf_counter()

    # ------------------------------------------------------------------ rank 0: the callbacks
    def _post(self, cmd):
        self._seq += 1
        self.ctrl[1] = cmd
        self.ctrl[0] = self._seq            # (x86: stores are not reordered with older stores)
        return self._seq

    def _as_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.plan.n,):
            raise ValueError(f"x must have shape ({self.plan.n},)")
        return x

    def _is_prepared(self, x):
        return self._x_seq >= 0 and bool(self.lib.pk_same_bits(x.ctypes.data, self.h_x.ctypes.data, self.plan.n))

    def _prepare(self, x):
        """A new x: post it, evaluate rank 0's own shard, wait until every rank's partial sums, grad f and g slices are in
        the segment (J is still landing: ``jacobian`` waits for it), finish f and the shared gradient slots."""
        x = self._as_x(x)
        if self._is_prepared(x):
            return
        self.lib.pk_copy_bits(self.h_x.ctypes.data, x.ctypes.data, self.plan.n)
        # J slices go ahead only while the solver keeps asking for grad f / J: an iterate it did not ask them for was a rejected
        # trial point of a line search, and its J on the links stood in the way of the next trial point's x
        if self.needs_I:
            seq = self._post(CMD_INT)
            self._do_int(seq)
            self._wait_marks("int", seq)
            total = self.h_part[0, : self.n_I].copy()
            for r in range(1, self.world):    # rank order: reproducible sums
                total += self.h_part[r, : self.n_I]
            self.h_I[:] = total
        ahead = self._j_asked
        self._j_asked = False
        seq = self._post(CMD_X if ahead else CMD_X_TRIAL)
        self._do_x_part(seq, with_j=ahead)
        self._j_seq = seq if ahead else -1       # (-1: the J slices of this iterate are still on the GPUs)
        self._wait_marks("early", seq)
        small = self.h_part[0].copy()
        for r in range(1, self.world):        # rank order: reproducible sums
            small += self.h_part[r]
        I = self.h_I if self.needs_I else small[: self.n_I]      # (needs_I: every rank's d_small[:n_I] held the TOTALS)
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

    def _result(self, key):
        """What a callback returns: an array the caller owns (the reference's semantics), or -- ``zero_copy``, which a solver
        adapter sets when the solver copies at once, as cyipopt does -- the live view of the shared segment, valid until the
        next iterate overwrites it."""
        view = self.out[key]
        return view if self.zero_copy else view.copy()

    def _alive(self):
        if self.shm is None or self.out is None:
            raise RuntimeError("HostShardedEvaluator is closed")

    def objective(self, x):
        self._alive()
        self._prepare(x)
        return np.float64(self.h_f[0])

    def gradient(self, x):
        self._alive()
        self._prepare(x)
        self._j_asked = True
        return self._result("grad")

    def constraints(self, x):
        self._alive()
        self._prepare(x)
        return self._result("g")

    def jacobian(self, x):
        self._alive()
        self._prepare(x)
        self._j_asked = True
        if self._j_seq == -1:                  # (the prepared iterate was taken for a trial point: its J is still on the GPUs)
            self._j_seq = self._post(CMD_J)
            self._do_j(self._j_seq)
        self._wait_marks("x", self._j_seq)
        return self._result("J")

    def hessian(self, x, lagrange, obj_factor):
        self._alive()
        lam = np.ascontiguousarray(lagrange, dtype=np.float64)
        if lam.shape != (self.plan.m,):
            raise ValueError(f"lagrange must have shape ({self.plan.m},)")
        x = self._as_x(x)
        self.lib.pk_copy_bits(self.h_lam.ctypes.data, lam.ctypes.data, self.plan.m)
        self.h_sigma[0] = float(obj_factor)
        if self._x_seq >= 0:
            # a solver asks for H at the x it has just evaluated: launch on the prepared x BEFORE comparing (the compare of
            # n doubles then runs while the GPUs work); in the rare other case drain and start over on the new x
            seq = self._post(CMD_HESS)
            self._do_hess(seq)
            same = self._is_prepared(x)
            self._wait_marks("h", seq)
            if same:
                return self._result("H")
        self._prepare(x)
        seq = self._post(CMD_HESS)
        self._do_hess(seq)
        self._wait_marks("h", seq)
        return self._result("H")

    # ------------------------------------------------------------------ shutdown
    def close(self):
        if getattr(self, "shm", None) is None:
            return
        if self.rank == 0 and self.world > 1 and getattr(self, "ctrl", None) is not None and getattr(self, "_seq", None) is not None:
            seq = self._post(CMD_EXIT)
            try:
                keep, self.timeout_s = self.timeout_s, min(self.timeout_s, 10.0)
                self._wait_all(seq)
                self.timeout_s = keep
            except RuntimeError:
                pass
        lib, h = self.lib, self.h
        if getattr(self, "helper_threads", 0) and getattr(self, "_own_helpers", False):      # (only the pool this object started)
            from . import runtime

            lib.pk_host_threads(0)
            runtime.host_helpers_stopped()
        self.helper_threads = 0
        if h:
            lib.pk_sync(h, None)
            if getattr(self, "_registered", False):
                lib.pk_host_unregister(h, self._host_base)
                self._registered = False
            for p in getattr(self, "_alloc", []):
                lib.pk_device_free(h, p)
        self._alloc = []
        self.out = self.h_x = self.h_lam = self.h_sigma = self.h_part = self.h_out = self.h_f = None
        self.words = self.ctrl = self.h_I = self._marks = None
        if self.ev is not None:
            self.ev.close()
        try:
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass
        self.shm = None
