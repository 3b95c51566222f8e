"""GPU evaluator of one transcribed system: plan -> code object + device tables -> callbacks.

``Evaluator(plan)`` generates the model's HIP source, compiles (or fetches from cache) its gfx950
code object, flattens the mesh-dependent tables into the blobs of csrc/pk_abi.h, uploads
everything through the C ABI and then serves the five NLP callbacks with host NumPy arrays
(pk_eval_*), or with device pointers and a stream for callers that keep data resident (pk_eval_*_dev).

Mirrors the callback semantics of /root/reference/pockit/base/systembase.py:602-835: callbacks
return freshly allocated float64 arrays; ``x`` is borrowed and never written.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import hipbuild, runtime
from .codegen import ModelSource
from .transcription import SystemPlan

N_CU = 256      # compute units of an MI355X (one workgroup of the cycle per CU is the sweet spot, see _intervals_per_wave)


def _intervals_per_wave(plan, override=None, shards=1, subs=0, want_workgroups=False):
    """Intervals per wavefront, up to 64 nodes per wave (Layout.tiles caps it).  Measured on MI355X with pk_cycle
    (tools/ipw_sweep.sh): at 12k nodes the cycle is bound by the number of vector-memory
    instructions a CU has to issue and by the time the dispatcher needs to start the waves, so fuller waves
    (18-42 nodes: 126-133k cycles/s) beat many small ones (12 nodes: 114k, 6 nodes: 81k) as long as every CU
    still gets work; the 40k-node humanoid runs best with full 64-node waves."""
    forced = override or os.environ.get("POCKIT_AMD_IPW")        # (tests / sweeps: tools/ipw_sweep.sh)
    if forced and not want_workgroups:
        return int(forced)
    # Candidates from ~16 nodes per wave (on small meshes a wave of 6-8 nodes pays its fixed work -- tile record, table
    # staging, one streaming iteration per segment -- for a quarter of the entries: tools/ipw_small_sweep.sh, humanoid
    # 100 x 8 10.7 -> 8.2 us per cycle, quadrotor 100 x 6 4.79 -> 4.14) up to full waves; the choice minimizes the work of
    # the busiest CU.  A launch is as slow as its most loaded CU: with 303 workgroups on 256 CUs 47 CUs hold two and the
    # launch takes as long as with 504 (tools/mid_size_sweep.sh: quadrotor 3000 x 6 with 380 tiles 6.09 us, with 304 tiles
    # 5.21 us).  Cost of a candidate = workgroups per CU x (intervals per wave [x 1.5 where one wave does the Jacobian and
    # the values] + 4 for a workgroup's fixed work); ties go to the fuller waves.  The model reproduces the optimum of the
    # sweeps at 100 ... 10 000 intervals (profiles/r02_g_ipw_small.txt, r02_g_mid_size.txt, r02_z2_tile_sweep.txt).
    shards = max(1, int(shards))
    best = None
    caps = []
    for pp in plan.phase_plans:
        lay = pp.layout
        kmed = max(1, int(np.median(np.asarray(lay.K, dtype=np.int64))))
        nodes = kmed if lay.scheme == "lgr" else max(kmed - 1, 1)
        caps.append((int(lay.N), max(1, (runtime.WAVE if lay.scheme == "lgr" else runtime.WAVE - 1) // nodes), kmed))
    if not caps:
        return 1
    lo = max(1, math.ceil(16 / max(1, int(np.median([k for _, _, k in caps])))))
    hi = max(lo, max(c for _, c, _ in caps))
    if forced:      # (the workgroups of the forced tiling, not the model's choice)
        lo = hi = int(forced)
    for ipw in range(lo, hi + 1):
        tiles = 0
        for n_p, cap, _ in caps:
            t = math.ceil(max(n_p / shards - 1, 0) / min(ipw, cap)) + 1        # (the first interval is a kind of its own)
            tiles += -(-t // runtime.WAVES_PER_BLOCK) * runtime.WAVES_PER_BLOCK
        roles = 3 if tiles <= 1024 else 2          # (pk_set_problem: the x-part is split into two roles up to 1024 tiles)
        if subs:
            roles = int(subs)
        per_cu = math.ceil((roles * tiles // runtime.WAVES_PER_BLOCK + 3) / N_CU)
        cost = per_cu * (ipw * (1.5 if roles == 2 else 1.0) + 4.0)      # (+4: a workgroup's fixed work, in intervals)
        if best is None or cost <= best[0]:
            best = (cost, ipw, roles * tiles // runtime.WAVES_PER_BLOCK + 3)
    return best[2] if want_workgroups else best[1]


def _launch_underfills_the_chip(plan, shards=1):
    """True when the one-launch cycle of this mesh, tiled by default, has no more workgroups than the GPU has CUs: the
    cycle's time is then one wave's chain, not throughput (see compile_plan)."""
    return bool(plan.phase_plans) and _intervals_per_wave(plan, shards=shards, want_workgroups=True) <= N_CU


def magic_number(d: int) -> int:
    """32-bit magic of the divisor ``d`` for the kernels' ``magic_div`` (pk_kernels.hip.h):
    ``p // d == (p * magic) >> 32`` for ``p < 2**16`` with ``magic = ceil(2**32 / d)``.  ``d == 1`` has no 32-bit magic
    (2**32): it is encoded as 0 and the kernel returns ``p`` itself; ``d == 0`` (nothing to divide) is 0 as well."""
    if d <= 1:
        return 0
    return (0xFFFFFFFF // d + 1) & 0xFFFFFFFF


def magic_div(p: int, magic: int) -> int:
    """Host model of the device function of the same name (used by the tests)."""
    return p if magic == 0 else (p * magic) >> 32


class Tables:
    """Mesh-dependent tables of a plan in the layout of csrc/pk_abi.h (host NumPy arrays).

    ``tile_filter(phase_index, tiles) -> tiles`` lets a rank keep only its shard of the tiles."""

    def __init__(self, plan: SystemPlan, src: ModelSource, intervals_per_wave=None, tile_filter=None, shards=1):
        ib, db, lb = [], [], []

        def put(store, arr, dtype):
            arr = np.asarray(arr, dtype=dtype).ravel()
            off = sum(len(a) for a in store)
            store.append(arr)
            return off

        nP = len(plan.phase_plans)
        phases = np.zeros(nP, dtype=runtime.PHASE_DTYPE)
        kinds, tiles = [], []
        ipw = _intervals_per_wave(plan, intervals_per_wave, shards=shards, subs=getattr(src, "cycle_subs", 0))
        for k, pp in enumerate(plan.phase_plans):
            lay = pp.layout
            kind0 = len(kinds)
            for kd in lay.kinds:
                rec = np.zeros((), dtype=runtime.KIND_DTYPE)
                rec["K"], rec["R"], rec["nnzI"], rec["nnzT"] = kd.K, kd.R, kd.nnzI, kd.nnzT
                rec["irc_off"] = put(ib, np.stack([kd.I_r, kd.I_c], axis=1) if kd.nnzI else np.zeros((0, 2)), np.int32)
                rec["iv_off"] = put(db, kd.I_v, np.float64)
                rec["tv_off"] = put(db, kd.T_v, np.float64)
                rec["full_off"] = put(db, kd.full, np.float64)
                kinds.append(rec)
            tl = lay.tiles(ipw)
            if tile_filter is not None:
                tl = tile_filter(k, tl)
            ph = phases[k]
            ph["scheme"] = 0 if lay.scheme == "lgr" else 1
            ph["n_x"], ph["n_u"], ph["n_c"] = pp.nx, pp.nu, pp.phase.n_c
            ph["L_m"], ph["L_d"], ph["state_len"], ph["L"] = lay.L_m, lay.L_d, lay.state_len, lay.L
            ph["x_off"], ph["g_off"], ph["path_off"] = plan.l_p[k], plan.g_off[k], plan.path_off[k]
            ph["mid_lo"], ph["mid_hi"] = lay.mid_lo, lay.mid_hi
            ph["tile_lo"], ph["tile_hi"] = len(tiles), len(tiles) + len(tl)
            ph["tau_off"] = put(db, lay.tau, np.float64)
            ph["w_off"] = put(db, lay.w, np.float64)
            ph["width_off"] = put(db, lay.width, np.float64)
            ph["n_int"] = lay.N
            ph["ivK_off"] = put(ib, lay.K, np.int32)
            ph["ivld_off"] = put(ib, lay.ld, np.int32)
            full_pos = [int(kinds[kind0 + int(kf)]["full_off"]) for kf in lay.kid_full]
            ph["ivfull_off"] = put(ib, full_pos, np.int32)
            cbs = [("jac", "jseg_off"), ("hess", "hseg_off"), ("aux", "aseg_off")]
            if src.compact:
                cbs.append(("hessc", "hcseg_off"))
            if src.compact_j:
                cbs.append(("jacc", "jcseg_off"))
            for cbname, field in cbs:
                segs = getattr(plan, cbname).segs[k]
                bases = [s.base for s in segs if s.kind == "I"] + [s.base for s in segs if s.kind == "D"] + \
                        [s.base for s in segs if s.kind == "N"]
                ph[field] = put(lb, bases, np.int64)
            ph["jt_off"] = put(lb, plan.jac.tconst[k], np.int64)
            if src.compact_j:
                ph["jct_off"] = put(lb, plan.jacc.tconst[k], np.int64)
            red = [plan.l_p[k] + s if s >= 0 else plan.r_s + s for s in plan.grad_red_slots[k]]
            ph["red_off"] = put(ib, red, np.int32)
            def empty_tile():
                e = np.zeros((), dtype=runtime.TILE_DTYPE)
                e["phase"], e["K"] = k, 1
                return e

            for row in tl:
                j0, nj, kid, kidf, q0, r0, offI, offT = (int(v) for v in row)
                big = int(lay.K[j0]) > runtime.WAVE       # more points than a wave has lanes: the interval takes a whole
                while big and len(tiles) % runtime.WAVES_PER_BLOCK:       # workgroup (first slot of a tile block)
                    tiles.append(empty_tile())
                rec = np.zeros((), dtype=runtime.TILE_DTYPE)
                rec["phase"], rec["j0"], rec["nj"] = k, j0, nj
                rec["kid"], rec["kidf"] = kind0 + kid, kind0 + kidf
                rec["q0"], rec["r0"], rec["offI"], rec["offT"] = q0, r0, offI, offT
                rec["K"] = int(lay.K[j0])
                rec["last"] = 1 if j0 + nj == lay.N else 0
                for f in ("nnzI", "nnzT", "irc_off", "iv_off", "tv_off"):
                    rec[f] = kinds[kind0 + kid][f]
                rec["full_off"] = kinds[kind0 + kidf]["full_off"]
                R = int(kinds[kind0 + kidf]["R"])
                for field, d in (("magicI", int(rec["nnzI"])), ("magicR", R), ("magicT", int(rec["nnzT"]))):
                    rec[field] = magic_number(d)
                tiles.append(rec)
                if big:
                    tiles.extend(empty_tile() for _ in range(runtime.WAVES_PER_BLOCK - 1))
            while len(tiles) % runtime.WAVES_PER_BLOCK:      # a workgroup never mixes phases: pad with empty tiles
                rec = np.zeros((), dtype=runtime.TILE_DTYPE)
                rec["phase"], rec["K"] = k, 1
                tiles.append(rec)
            ph["tile_hi"] = len(tiles)
        # gradient slots no tile writes: state end slots (LGR), t0/tf, static parameters
        gz = []
        for k, pp in enumerate(plan.phase_plans):
            lay = pp.layout
            if lay.scheme == "lgr":
                gz += [plan.l_p[k] + lay.l_v[i] + lay.L_m for i in range(pp.nx)]
            gz += [plan.l_p[k] + lay.L - 2, plan.l_p[k] + lay.L - 1]
        gz += list(range(plan.l_s, plan.r_s))
        self.gz_off, self.n_gz = put(ib, gz, np.int32), len(gz)

        def items(cbname):
            cb = getattr(plan, cbname)
            off = src.list_off[cbname]
            arr = np.zeros(len(cb.items), dtype=runtime.ITEM_DTYPE)
            for i, it in enumerate(cb.items):
                arr[i] = (it.pos, it.coef, off[it.lst] + it.eid, it.lam)
            return arr

        self.items_jac, self.items_hess, self.items_aux = items("jac"), items("hess"), items("aux")
        self.items_hessc = items("hessc") if src.compact else np.zeros(0, dtype=runtime.ITEM_DTYPE)
        self.items_jacc = items("jacc") if src.compact_j else np.zeros(0, dtype=runtime.ITEM_DTYPE)
        self.outer = np.zeros(len(plan.outer), dtype=runtime.OUTER_DTYPE)
        for i, b in enumerate(plan.outer):
            flags = (1 if b.tril else 0) | (2 if b.collapseA else 0) | (4 if b.collapseB else 0) | (8 if b.second else 0)
            self.outer[i] = (b.pos, b.offA, b.lenA, b.offB, b.lenB, b.offM, flags, b.count, 0)
        self.phases = phases
        self.kinds = np.array(kinds, dtype=runtime.KIND_DTYPE) if kinds else np.zeros(0, runtime.KIND_DTYPE)
        self.tiles = np.array(tiles, dtype=runtime.TILE_DTYPE) if tiles else np.zeros(0, runtime.TILE_DTYPE)
        cat = lambda store, dt: np.concatenate(store).astype(dt) if store else np.zeros(0, dt)  # noqa: E731
        self.ib, self.db, self.lb = cat(ib, np.int32), cat(db, np.float64), cat(lb, np.int64)
        self.intervals_per_wave = ipw


def compile_plan(plan: SystemPlan, sharded=False, output_share=1.0, extra_flags=(), fixed=None):
    """Generated source + gfx950 code object of a plan.  The group size (codegen.split_groups: how many derivative entries a
    pass evaluates, stages and streams) is searched under two criteria, in this order:

    * **LDS** -- every launch of the code object must fit the 160 KiB of a workgroup *including* the table blocks the
      runtime adds (``ModelSource.launch_lds_bytes``, the bytes ``pk_runtime.cpp::launch_raw`` asks for): the size is halved
      until it does.  Nothing a wave stages grows with the number of states (WIDE phases, codegen.py), so a fitting size
      exists for every model; if none did, this raises instead of handing the library a model it must reject.
    * **registers** -- a model whose kernels would spill vector registers to scratch memory is generated again with smaller
      groups (every size down to 4, the fewest spilled registers win).  What is left is reported: ``ModelSource.spilling_kernels``
      and a ``RuntimeWarning`` (the kernels are correct, a spilling one is slow).

    Every size tried stays cached, so a model pays its extra compiles once.  Returns (ModelSource, code object)."""
    fast = plan.system._fastmath

    generated = {}

    def generate(cap, wide_nx=None):
        if (cap, wide_nx) not in generated:
            generated[(cap, wide_nx)] = ModelSource(plan, sharded=sharded, output_share=output_share, group_cap=cap, wide_nx=wide_nx)
        return generated[(cap, wide_nx)]

    def build(cap, src=None):
        src = src or generate(cap)
        code = hipbuild.compile_model(src.source, fastmath=fast, extra_flags=extra_flags)
        src.spilling_kernels = hipbuild.spills(hipbuild.resource_usage(src.source, fastmath=fast, extra_flags=extra_flags))
        return src, code, sum(v[0] for v in src.spilling_kernels.values())

    if fixed is not None:      # (Evaluator.checked's rebuild: the group size and wide threshold of the build it replaces, ONE compile
        best = build(int(fixed[0]), generate(int(fixed[0]), fixed[1]))      #  instead of the search -- with the flags of that rebuild
        return best[0], best[1]                                              #  the search took more than half an hour on a two-phase model)
    fixed_cap = bool(os.environ.get("POCKIT_AMD_GROUP_CAP"))
    cap0 = generate(None).group_cap
    while not fixed_cap and cap0 > 1 and not generate(cap0).fits_lds():
        cap0 //= 2
    if not generate(cap0).fits_lds():
        worst = max(generate(cap0).launch_lds_bytes().items(), key=lambda kv: kv[1])
        raise ValueError(f"the model does not fit a workgroup's LDS at any group size: {worst[0]} needs {worst[1]} bytes with "
                         f"groups of {cap0} (limit {ModelSource.LDS_LIMIT})")

    # A launch that underfills the chip (small and medium meshes: what the reference's example programs use) is as slow as
    # one wave's chain of evaluation, staging and streaming: groups of 16 instead of 32 split the roles of a moderately
    # large model into passes that run as workgroups of their own (humanoid 25 ... 500 x 8: 8.3 -> 7.3 ... 6.6 us per cycle);
    # with the chip full (humanoid 5000 x 8) the single pass is 4 % faster (profiles/r04_ze_*.txt).  The choice is a fact of
    # the mesh like PK_TAB_CAP: a refinement that crosses the line costs one more (cached) compile.
    free = not fixed_cap and os.environ.get("POCKIT_AMD_PASS_PARALLEL", "auto") == "auto"
    settled = None                    # (the underfill rule's choice: the group-size searches below are skipped, not the rest)
    if free and cap0 > ModelSource.GROUP_CAP // 2 and _launch_underfills_the_chip(plan, max(1, int(round(1.0 / output_share)))):
        probe = generate(ModelSource.GROUP_CAP // 2)
        if probe.grouped and probe.cycle_subs and probe.fits_lds():
            trial = build(probe.group_cap, probe)
            if trial[2] == 0:
                settled = trial
    best = settled or build(cap0)
    if settled is None and best[0].grouped and not fixed_cap:
        cap = best[0].group_cap      # (every size down to 4 while registers still spill: the count is not monotonic in the size.
        while best[2] > 0 and cap > 4:      # Some models keep spills at EVERY size -- DESIGN.md section 11 lists the measured ones)
            cap //= 2
            trial = build(cap)
            if trial[2] < best[2]:
                best = trial
    # The cycle of a model evaluated in groups is fastest with every pass as a workgroup of its own (ModelSource.cycle_subs,
    # DESIGN.md section 3c), which needs two workgroups of the launch on a CU: when the groups need too much LDS for that,
    # half the size is tried (rocket_powered_descent at 2000 x 4: 22.6 -> 11.3 us per cycle, drone_stabilization 15.2 -> 13.6;
    # profiles/r04_y_fat_pp_sweep.txt) -- unless it brings spills back.
    src = best[0]
    if free and src.grouped and not src.cycle_subs:
        cap = src.group_cap
        while cap > 8:
            cap //= 2
            probe = generate(cap)
            if not probe.cycle_subs:
                continue                     # (generated only: the rows of this size still leave no room for two workgroups)
            trial = build(cap, probe)
            if trial[2] <= best[2]:
                best = trial
            break
    # A pass-parallel model whose cycle kernel leaves room for ONE wave per SIMD runs its workgroups in two rounds (one
    # workgroup per CU at a time): the values workgroups of the second round publish their partial sums late and the launch
    # ends with the finalize workgroup (drone_stabilization at 2000 x 4: 480 workgroups, values waves entering up to 5.4 us
    # into a 13.3 us launch, profiles/r05_m_drone_wave_timeline.txt).  The register hog is the values role of a phase with
    # 9 ... 16 states (all states' node values, end slots and row sums at once: 250 VGPRs); evaluated the WIDE way -- chunks of
    # 8 states -- it needs far fewer.  Tried only there, kept only if it raises the occupancy without spills.
    src = best[0]
    if free and src.cycle_subs and best[2] == 0 and src.wide_nx == ModelSource.WIDE_NX and \
            any(ModelSource.WIDE_NX_LOW < pp.nx <= ModelSource.WIDE_NX for pp in plan.phase_plans):
        occ = lambda s_: ((hipbuild.resource_usage(s_.source, fastmath=fast, extra_flags=extra_flags) or {}).get("pk_cycle") or {}).get("occupancy", 0)  # noqa: E731
        if occ(src) == 1:
            probe = generate(src.group_cap, ModelSource.WIDE_NX_LOW)
            if probe.cycle_subs and probe.fits_lds():
                trial = build(src.group_cap, probe)
                if trial[2] == 0 and occ(trial[0]) > 1:
                    best = trial
    if best[2] > 0:
        import warnings

        warnings.warn("pockit_amd: kernels of this model spill vector registers to scratch memory at the best group size found "
                      f"({best[0].group_cap}): " + ", ".join(f"{k} {v[0]} VGPRs / {v[1]} B" for k, v in best[0].spilling_kernels.items()),
                      RuntimeWarning, stacklevel=2)
    return best[0], best[1]


class Evaluator:
    #: the one build change with evidence behind it for the round-5 defect: SGPR spills to scratch memory
    SGPR_TO_SCRATCH = ("-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0")

    @classmethod
    def checked(cls, plan: SystemPlan, **kw):
        """An evaluator whose FUSED kernel (pk_cycle: the x-callbacks and the one-launch cycle) has been verified against its
        own stand-alone kernels at a probe point (``self_check``).  Round 5 found code objects -- models wide in several
        directions, four cases, cause open -- whose fused kernel returned wrong f / grad / g while
        every stand-alone kernel was exact, with no error.  On a mismatch the model is built once more with the compiler's SGPR
        spills sent to scratch memory (the change that made every affected build tested exact) and checked again; if that
        fails too, this raises instead of handing out an evaluator that answers wrongly.  POCKIT_AMD_SELF_CHECK=0 skips it."""
        ev = cls(plan, **kw)
        if os.environ.get("POCKIT_AMD_SELF_CHECK", "1") == "0":
            return ev
        ok, worst = ev.self_check()
        if ok:
            return ev
        import warnings

        warnings.warn(f"pockit_amd: the fused kernel of this model's code object failed its self-check against the stand-alone "
                      f"kernels ({worst}); rebuilding with SGPR spills in scratch memory (DESIGN.md section 11)", RuntimeWarning, stacklevel=3)
        fixed = (ev.src.group_cap, ev.src.wide_nx)
        ev.close()
        ev = cls(plan, hipcc_flags=cls.SGPR_TO_SCRATCH, _fixed=fixed, **kw)
        ok, worst2 = ev.self_check()
        if ok:
            return ev
        # Both builds disagree with their own stand-alone kernels, which have been exact in every case seen: serve the model
        # through THEM (five launches per iterate instead of one -- what a model that needs the integrals first gets anyway).
        ev.close()
        ev = cls(plan, **kw)
        if ev.src.big:      # (a mesh with intervals of more than 64 points has the fused x-kernel only)
            ev.close()
            raise RuntimeError("pockit_amd: the fused kernel of this model's code object fails its self-check against the stand-alone "
                               f"kernels in both builds tried ({worst}; with SGPR spills in scratch memory: {worst2}) -- an open defect "
                               "(DESIGN.md section 11) -- and this mesh (an interval of more than 64 points) has no stand-alone "
                               "path.  No evaluator is handed out rather than one that answers wrongly")
        ev.ctx.check(ev.ctx.lib.pk_set_host_option(ev.ctx.handle, b"separate_x", 1))
        ev.separate_x = True
        warnings.warn(f"pockit_amd: both builds of this model fail the self-check of their fused kernel ({worst}; {worst2}): the "
                      "callbacks are served by the stand-alone kernels, one launch each (slower, exact in every case seen; DESIGN.md "
                      "section 11)", RuntimeWarning, stacklevel=3)
        return ev

    def self_check(self, tol=1.0e-9):
        """(ok, description of the worst disagreement).  One probe point -- x in [0.7, 1.3], lambda ~ N(0, 1), sigma 1 -- through
        the fused launch and through pk_int + pk_fin, pk_grad, pk_g, pk_jac, pk_hess; entries that are not finite must be so in
        both.  A consistency check, not a physical point.  (Meshes with intervals of more than 64 points and models that need
        the integrals first serve both sides from the same kernels: the check is vacuous there.)"""
        rng = np.random.default_rng(20240531)
        x = 0.7 + 0.6 * rng.uniform(size=self.plan.n)
        lam = rng.standard_normal(self.plan.m)
        with np.errstate(all="ignore"):
            fused = [np.array(v, dtype=np.float64, ndmin=1) for v in self.cycle(x, lam, 1.0)]
            alone = [np.array(v, dtype=np.float64, ndmin=1) for v in (self.objective_direct(x), self.gradient_direct(x), self.constraints_direct(x),
                                                                       self.jacobian_direct(x), self.hessian_direct(x, lam, 1.0))]
        self._invalidate_x()
        worst, ok = "", True
        for name, a, b in zip(("f", "grad f", "g", "J", "H"), fused, alone):
            fa, fb = np.isfinite(a), np.isfinite(b)
            if a.shape != b.shape or not np.array_equal(fa, fb):
                ok, worst = False, f"{name}: shapes / finite entries differ"
                break
            if fa.any():
                err = float(np.max(np.abs(a[fa] - b[fa]))) / max(1.0, float(np.max(np.abs(b[fb]))))
                if err > tol:
                    ok, worst = False, f"{name}: relative difference {err:.2e}"
                    break
        return ok, worst

    def __init__(self, plan: SystemPlan, device: int = 0, intervals_per_wave=None, tile_filter=None, sharded=False,
                 output_share=1.0, host_helpers=True, hipcc_flags=(), _fixed=None):
        self.plan = plan
        self.hipcc_flags = tuple(hipcc_flags)
        self.separate_x = False      # (True: Evaluator.checked switched this context to the stand-alone kernels)
        self._want_host_helpers = bool(host_helpers) and tile_filter is None
        # (compiled before the context is created: a box without a GPU -- the build container -- can still fill the
        # code-object cache by constructing evaluators, tools/warm_cache.sh)
        self.src, code = compile_plan(plan, sharded=sharded, output_share=output_share, extra_flags=self.hipcc_flags, fixed=_fixed)
        self.ctx = runtime.Context(device)            # raises RuntimeError without a GPU
        lib, h = self.ctx.lib, self.ctx.handle
        md = runtime.ModelDesc()
        md.n_phase, md.n_I, md.nred = self.src.nphase, max(len(plan.I_syms), 1), self.src.nred
        md.lds_g, md.lds_j, md.lds_h, md.lds_x = self.src.lds_g, self.src.lds_j, self.src.lds_h, self.src.lds_x
        md.ne_j, md.ne_h = self.src.list_off["jac"]["total"], self.src.list_off["hess"]["total"]
        md.ne_a = self.src.list_off["aux"]["total"]
        md.ne_hc = self.src.list_off["hessc"]["total"] if self.src.compact else 0
        md.lds_e = self.src.lds_e
        md.lds_jc = self.src.lds_jc
        md.ne_jc = self.src.list_off["jacc"]["total"] if self.src.compact_j else 0
        md.tab_cap = self.src.tab_cap
        md.sharded = int(self.src.sharded)
        md.max_phases = self.src.max_phases
        md.cycle_subs = self.src.cycle_subs
        md.hess_subs = self.src.h_ngmax if self.src.cycle_subs else 0
        # (the stand-alone compact kernels of such a model: a workgroup per pass too)
        md.hessc_subs = self.src.hc_ngmax if (self.src.cycle_subs and self.src.hc_ngmax > 1) else 0
        md.jacc_subs = self.src.jc_ngmax if (self.src.cycle_subs and self.src.jc_ngmax > 1) else 0
        md.big_global, md.big_rows = int(self.src.big_global), int(self.src.big_rows)
        md.wide = int(any(self.src.wide))      # (the library then refuses pk_xall: _refuse_sequential_values_role)
        self._err_views = None
        self._csr = {}
        md.prepass_f = 1
        md.prepass_grad = int(plan.needs_I_grad)
        md.prepass_g = int(plan.needs_I_con)
        md.prepass_jac = int(plan.jac.needs_I)
        md.prepass_hess = int(plan.hess.needs_I)
        self._code = code
        self._views = {}
        self.zero_copy = False   # True: callbacks return views of pinned buffers (set by the IPOPT adapter)
        self.writable_results = os.environ.get("POCKIT_AMD_WRITABLE_RESULTS", "0") == "1"    # (see _result)
        if (md.prepass_grad or md.prepass_g or md.prepass_jac or md.prepass_hess) and self.src.big:
            # (integrals first AND an interval of more than 64 points: the x-callbacks then run pk_xall behind the prepass)
            self._refuse_sequential_values_role("a model that needs the integrals first, on a mesh with an interval of more than 64 points,")
        self.ctx.check(lib.pk_load_model(h, code, len(code), C.byref(md)))
        self.model_desc = md
        # (the tiling is sized for ONE shard's share of the mesh: output_share = 1 / number of shards)
        self.set_tables(Tables(plan, self.src, intervals_per_wave, tile_filter, shards=max(1, int(round(1.0 / output_share)))))

    def set_tables(self, tb: Tables):
        plan, lib, h = self.plan, self.ctx.lib, self.ctx.handle
        self.tables = tb
        self._views = {}
        pd = runtime.ProblemDesc()
        pd.n, pd.m, pd.n_sys, pd.n_s, pd.l_s = plan.n, plan.m, plan.n_sys, plan.n_s, plan.l_s
        pd.n_phase, pd.n_tiles, pd.n_kinds = len(tb.phases), len(tb.tiles), len(tb.kinds)
        pd.nnz_J, pd.nnz_H = plan.nnz_J, plan.nnz_H
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        pd.phases, pd.tiles, pd.kinds = vp(tb.phases), vp(tb.tiles), vp(tb.kinds)
        pd.items_jac, pd.n_items_jac = vp(tb.items_jac), len(tb.items_jac)
        pd.items_hess, pd.n_items_hess = vp(tb.items_hess), len(tb.items_hess)
        pd.ib, pd.n_ib = tb.ib.ctypes.data_as(runtime.c_int32_p), len(tb.ib)
        pd.db, pd.n_db = tb.db.ctypes.data_as(runtime.c_double_p), len(tb.db)
        pd.lb, pd.n_lb = tb.lb.ctypes.data_as(C.POINTER(C.c_int64)), len(tb.lb)
        pd.gz_off, pd.n_gz = tb.gz_off, tb.n_gz
        pd.items_aux, pd.n_items_aux = vp(tb.items_aux), len(tb.items_aux)
        pd.outer, pd.n_outer, pd.n_aux = vp(tb.outer), len(tb.outer), plan.n_aux
        pd.items_hessc, pd.n_items_hessc = vp(tb.items_hessc), len(tb.items_hessc)
        pd.nnz_Hc = plan.nnz_Hc if self.src.compact else 0
        pd.items_jacc, pd.n_items_jacc = vp(tb.items_jacc), len(tb.items_jacc)
        pd.nnz_Jc = plan.nnz_Jc if self.src.compact_j else 0
        self._struct = [np.ascontiguousarray(a, dtype=np.int32) for a in
                        (plan.jac_row, plan.jac_col, plan.hess_row, plan.hess_col)]
        pd.jac_row, pd.jac_col, pd.hess_row, pd.hess_col = (a.ctypes.data_as(runtime.c_int32_p) for a in self._struct)
        self.ctx.check(lib.pk_set_problem(h, C.byref(pd)))
        # (pk_set_problem starts every problem in the reference layouts: what set_cycle_layout chose is applied again, so that
        #  a caller's compact-sized buffers never receive reference-layout writes after a change of tables)
        jc, hc = getattr(self, "_cycle_layout", (False, False))
        if jc or hc:
            self.set_cycle_layout(jc, hc)
        self._host_setup()

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    # ------------------------------------------------------------------ host-array callbacks
    def _x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.plan.n,):
            raise ValueError(f"x must have shape ({self.plan.n},)")
        return x

    # IPOPT calls objective / gradient / constraints / jacobian one after the other on the same iterate and then
    # hessian with new multipliers (the five methods optimizer/ipopt.py hands to cyipopt; reference ipopt.py:41-53).
    # Every callback is ONE call into the library (pk_callback_x / pk_callback_hess): a bitwise compare decides whether x is
    # the prepared iterate; a new x is staged, uploaded, evaluated by the fused x-kernel and all of its results are on their
    # way to pinned host memory behind it; the callback then only waits for its own result.
    #
    # Where results land: ``zero_copy`` (set by the IPOPT adapter: cyipopt copies a result at once) hands out views of
    # the context's pinned buffers, valid until the next iterate.  Otherwise the callbacks return arrays the caller
    # owns, as the reference's do -- but over pinned memory the DMA writes directly (runtime.PinnedRing): J, grad f and g of
    # one iterate are three views of ONE block laid out like the device's buffer, [J | grad f | g], so that they arrive in a
    # single DMA; the x-independent entries of J (the translation part, ``SystemPlan.jac_constant_runs``) were put into the
    # block when it was allocated and never cross PCIe again.  A block is recycled only when the caller no longer refers
    # to any of its arrays.
    HEAD_RUN_MIN = 1024         # constant runs worth leaving out of the copy: at the head of J (the copy just starts later)
    INNER_RUN_MIN = 80_000      # ... and inside it (the copy splits in two: ~10 us per extra DMA = 0.5 MB of link time)

    def _host_setup(self):
        p = self.plan
        self._jac_compact = False
        self._nj = p.nnz_J          # Jacobian values of a landing block (the layout the Jacobian callback serves)
        self._blocks = runtime.PinnedRing(p.nnz_J + p.n + p.m)
        self._blocks_ref = self._blocks
        self._ring_h = runtime.PinnedRing(p.nnz_H)
        self._ring_hc = None
        self._next = self._cur = None
        self._cur_views, self._handed = {}, set()
        self._c_f, self._c_fresh = C.c_double(), C.c_int()
        self._a_f, self._a_fresh = C.addressof(self._c_f), C.addressof(self._c_fresh)
        self._const_runs = {}
        self.jac_constant_runs = self._register_constant_runs(False)
        # large systems: the solver thread's passes over x and lambda (a compare per callback, the staging copies) get helpers
        self.host_helper_threads = runtime.host_helpers(self.ctx.lib, p.n) if self._want_host_helpers else 0

    def _register_constant_runs(self, compact):
        """The x-independent runs of the Jacobian (in the layout the shim serves right now) worth leaving out of the copy."""
        if compact not in self._const_runs:
            runs = [(a, b) for a, b in self.plan.jac_constant_runs(compact)
                    if b - a >= (self.HEAD_RUN_MIN if a == 0 else self.INNER_RUN_MIN)]
            if runs:
                lo = (C.c_int64 * len(runs))(*[a for a, _ in runs])
                hi = (C.c_int64 * len(runs))(*[b for _, b in runs])
                self.ctx.check(self.ctx.lib.pk_set_jac_constant_runs(self.ctx.handle, len(runs), lo, hi))
            self._const_runs[compact] = runs
        return self._const_runs[compact]

    def set_jacobian_layout(self, compact):
        """The layout ``jacobian()`` and the J part of the landing blocks have: the reference's triplets or the compact
        layout (``plan.jacc_row/col``; an extra launch of pk_jacc behind the fused x-kernel, fewer values over PCIe)."""
        compact = bool(compact)
        if compact == self._jac_compact:
            return
        self.ctx.check(self.ctx.lib.pk_set_jacobian_layout(self.ctx.handle, int(compact)))
        p = self.plan
        self._jac_compact = compact
        self._nj = p.nnz_Jc if compact else p.nnz_J
        if compact:
            if getattr(self, "_blocks_compact", None) is None:
                self._blocks_compact = runtime.PinnedRing(p.nnz_Jc + p.n + p.m)
            self._blocks = self._blocks_compact
        else:
            self._blocks = self._blocks_ref
        self.jac_constant_runs = self._register_constant_runs(compact)
        self._next = self._cur = None
        self._cur_views, self._handed = {}, set()

    def _next_block(self):
        """The landing block [J | grad f | g] of the next new iterate (None: zero-copy mode, or the caller holds on to every
        block of the ring -- the results then land in the context's buffers and are copied)."""
        if self.zero_copy:
            return None
        it = self._next
        if it is None:
            it = self._blocks.take_item()
            if it is None:
                return None
            if not it.ready:
                if self.jac_constant_runs:
                    self.ctx.check(self.ctx.lib.pk_fill_jac_constants(self.ctx.handle, it.address))
                it.ready = True
            self._next = it
        return it

    def _became_current(self, it):
        """A new iterate was prepared; its x-results land in ``it`` (None: in the context's buffers)."""
        self._cur, self._next, self._handed = it, None, set()
        if it is None:
            self._cur_views = {}
        else:
            p, root, nj = self.plan, it.root, self._nj
            self._cur_views = {3: root[: nj], 1: root[nj: nj + p.n], 2: root[nj + p.n: nj + p.n + p.m]}

    def _callback_x(self, what, x):
        x = self._x(x)
        it = self._next_block()
        rc = self.ctx.lib.pk_callback_x(self.ctx.handle, what, x.ctypes.data, it.address if it is not None else None,
                                        self._a_f, self._a_fresh)
        if rc:
            self.ctx.check(rc)
        if self._c_fresh.value:
            self._became_current(it)

    def _invalidate_x(self):
        """The context's x / result buffers are about to be used by an entry point outside the prepared-x protocol."""
        if self.ctx is not None:
            self.ctx.lib.pk_invalidate_x(self.ctx.handle)

    def _pinned(self, what):
        """NumPy view of the context's pinned result buffer ``what`` (0 f, 1 grad, 2 g, 3 jac, 4 hess)."""
        if what not in self._views:
            ptr, cnt = runtime.c_double_p(), C.c_int64()
            self.ctx.check(self.ctx.lib.pk_host_buffer(self.ctx.handle, what, C.byref(ptr), C.byref(cnt)))
            self._views[what] = np.ctypeslib.as_array(ptr, shape=(max(cnt.value, 1),))[: cnt.value]
        return self._views[what]

    def _result(self, what, count):
        """The array result ``what`` of the current iterate landed in, as the callback's return value."""
        own = self._cur_views.get(what)
        if own is None:                      # landed in the context's buffer: zero-copy mode, or the ring is exhausted
            view = self._pinned(what)[:count]           # (the caller keeps many results): plain array + host copy
            return view if self.zero_copy else view.copy()
        if self.zero_copy:
            return own
        if what in self._handed:             # asked twice for the same iterate: the first array is the caller's
            return own.copy()
        self._handed.add(what)
        if what == 3 and self.jac_constant_runs:
            # The landing block keeps the x-independent entries of J (the +-1 translation part) from iterate to iterate: they
            # were filled in once and never cross PCIe again.  A caller that scaled this array in place would corrupt them for
            # every later Jacobian served from the block, so by default the array is handed out READ-ONLY (the reference
            # returns a fresh writable array; ``.copy()`` gives one).  ``writable_results = True`` (System.writable_results):
            # the array is writable and its block is marked to have the constant entries filled in again before it is
            # reused -- the reference's semantics at the price of one host pass over those entries per iterate.
            if self.writable_results:
                if self._cur is not None:
                    self._cur.ready = False
            else:
                own.flags.writeable = False
        return own

    def objective(self, x):
        self._callback_x(0, x)
        return np.float64(self._c_f.value)

    def gradient(self, x):
        self._callback_x(1, x)
        return self._result(1, self.plan.n)

    def constraints(self, x):
        self._callback_x(2, x)
        return self._result(2, self.plan.m)

    def jacobian(self, x):
        self._callback_x(3, x)
        return self._result(3, self._nj)

    def _lam(self, lagrange):
        lam = np.ascontiguousarray(lagrange, dtype=np.float64)
        if lam.shape != (self.plan.m,):
            raise ValueError(f"lagrange must have shape ({self.plan.m},)")
        return lam

    def _callback_hess(self, x, lam, obj_factor, target, compact):
        x = self._x(x)
        it = self._next_block()
        rc = self.ctx.lib.pk_callback_hess(self.ctx.handle, x.ctypes.data, lam.ctypes.data, float(obj_factor),
                                           it.address if it is not None else None, target, compact, self._a_fresh)
        if rc:
            self.ctx.check(rc)
        if self._c_fresh.value:
            self._became_current(it)

    def hessian(self, x, lagrange, obj_factor):
        lam = self._lam(lagrange)
        hit = None if self.zero_copy else self._ring_h.take_item()
        self._callback_hess(x, lam, obj_factor, hit.address if hit is not None else None, 0)
        if hit is None:                      # the context's buffer: zero-copy mode, or the caller keeps every Hessian
            view = self._pinned(4)[: self.plan.nnz_H]
            return view if self.zero_copy else view.copy()
        return hit.array

    def set_host_mode(self, prefetch=True, host_direct=False):
        """``prefetch`` (default True): all four x-results are copied to the host right behind the kernel; False: f and
        g always, grad f and J when first asked for.  ``host_direct``: the kernels store into pinned host memory
        themselves instead of device memory + DMA (A/B switch)."""
        self.ctx.check(self.ctx.lib.pk_set_host_mode(self.ctx.handle, int(bool(prefetch)), int(bool(host_direct))))

    # one-shot variants without the x cache (each uploads x and runs only its own kernels)
    def objective_direct(self, x):
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        out = np.empty(1)
        self.ctx.check(self.ctx.lib.pk_eval_f(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return np.float64(out[0])

    def gradient_direct(self, x):
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        out = np.empty(self.plan.n)
        self.ctx.check(self.ctx.lib.pk_eval_grad(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return out

    def constraints_direct(self, x):
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        out = np.empty(self.plan.m)
        self.ctx.check(self.ctx.lib.pk_eval_g(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return out

    def jacobian_direct(self, x):
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        out = np.empty(self.plan.nnz_J)
        self.ctx.check(self.ctx.lib.pk_eval_jac(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return out

    def hessian_direct(self, x, lagrange, obj_factor):
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        lam = np.ascontiguousarray(lagrange, dtype=np.float64)
        out = np.empty(self.plan.nnz_H)
        self.ctx.check(self.ctx.lib.pk_eval_hess(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(lam),
                                                 float(obj_factor), runtime.as_dp(out)))
        return out

    def hessian_compact(self, x, lagrange, obj_factor):
        """Values of the compact (coalesced) Hessian layout ``plan.hessc_row/col``."""
        if not self.src.compact:
            raise NotImplementedError("compact Hessian layout is not available for this model")
        lam = self._lam(lagrange)
        # the prepared-x protocol, like hessian(): the x of the iterate is already on the device (objective ... jacobian ran
        # on it), the values come back by DMA into a pinned array of the caller's
        if self._ring_hc is None:
            self._ring_hc = runtime.PinnedRing(self.plan.nnz_Hc)
        hit = self._ring_hc.take_item()
        if hit is not None:
            self._callback_hess(x, lam, obj_factor, hit.address, 1)
            return hit.array
        # (the caller keeps every array of the ring: a plain array, filled through the context's pinned buffer)
        lib, h = self.ctx.lib, self.ctx.handle
        self._callback_x(2, x)
        out = np.empty(self.plan.nnz_Hc)
        self.ctx.check(lib.pk_eval_hessc_prepared(h, runtime.as_dp(lam), float(obj_factor), runtime.as_dp(out), 0))
        return out

    def jacobian_compact(self, x):
        """Values of the compact (coalesced) Jacobian layout ``plan.jacc_row/col`` (one pk_jacc launch; the one-shot entry
        point -- a solver loop uses ``set_jacobian_layout(True)`` and ``jacobian()``)."""
        x = self._x(x)
        self._invalidate_x()        # the context's x buffer is about to hold another iterate
        out = np.empty(self.plan.nnz_Jc)
        self.ctx.check(self.ctx.lib.pk_eval_jacc(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return out

    def mesh_error(self, x):
        """Mesh error estimation data of every phase at the NLP point ``x`` (one pk_err launch): a list of
        ``(T, I)`` with shape (n_x, rows) each -- the two sides of the collocation equation on every interval
        re-collocated with one more point (reference: phasebase.py:1339-1372)."""
        from . import refine

        lib, h = self.ctx.lib, self.ctx.handle
        if self._err_views is None:
            recs, tables, n_out, views, groups = refine.error_tables(self.plan)
            self.ctx.check(lib.pk_set_mesh_error_tables(h, recs.ctypes.data, len(recs), groups.ctypes.data, len(groups),
                                                        runtime.as_dp(tables),
                                                        len(tables), n_out))
            self._err_views, self._err_len = views, n_out
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        T, I = np.empty(self._err_len), np.empty(self._err_len)
        self.ctx.check(lib.pk_eval_mesh_error(h, runtime.as_dp(x), runtime.as_dp(T), runtime.as_dp(I)))
        return [(T[o: o + nx * rows].reshape(nx, rows), I[o: o + nx * rows].reshape(nx, rows))
                for o, nx, rows in self._err_views]

    # ------------------------------------------------------------------ device-resident CSR hand-off
    def csr_map(self, which):
        """``CsrMap`` of the Jacobian (``"jac"``) or of the lower triangle of the Hessian of the Lagrangian
        (``"hess"``, reference layout); built and uploaded on first use."""
        from .csr import CsrMap

        if which not in ("jac", "hess"):
            raise ValueError('which must be "jac" or "hess"')
        if which not in self._csr:
            plan = self.plan
            if which == "jac":
                m = CsrMap(plan.jac_row, plan.jac_col, (plan.m, plan.n))
            else:
                m = CsrMap(plan.hess_row, plan.hess_col, (plan.n, plan.n))
            seg = None if m.seg is None else m.seg.ctypes.data_as(runtime.c_int32_p)
            self.ctx.check(self.ctx.lib.pk_set_csr_map(self.ctx.handle, 0 if which == "jac" else 1, seg,
                                                       m.perm.ctypes.data_as(runtime.c_int32_p), m.nnz, m.n_triplets))
            self._csr[which] = m
            if which == "jac" and self.src.compact_j:
                # the CSR values of J from the compact evaluation: its few repeated positions are summed by the gather
                plan.jacc  # noqa: B018  (builds the compact plan)
                mc = CsrMap(plan.jacc_row, plan.jacc_col, (plan.m, plan.n))
                if mc.n_triplets < m.n_triplets and mc.nnz == m.nnz and np.array_equal(mc.indices, m.indices) \
                        and np.array_equal(mc.indptr, m.indptr):      # (only where the compact layout writes fewer values)
                    segc = None if mc.seg is None else mc.seg.ctypes.data_as(runtime.c_int32_p)
                    self.ctx.check(self.ctx.lib.pk_set_csr_map(self.ctx.handle, 3, segc,
                                                               mc.perm.ctypes.data_as(runtime.c_int32_p), mc.nnz, mc.n_triplets))
                    self._csr["jacc"] = mc
            if which == "hess" and self.src.compact:
                # the compact Hessian has one value per distinct (row, col): if its pattern is the full pattern's set of
                # entries, the CSR values are a permutation of it (pk_eval_hess_csr then never writes the repeats)
                mc = CsrMap(plan.hessc_row, plan.hessc_col, (plan.n, plan.n))
                if mc.seg is None and mc.nnz == m.nnz and np.array_equal(mc.indices, m.indices) \
                        and np.array_equal(mc.indptr, m.indptr):
                    self.ctx.check(self.ctx.lib.pk_set_csr_map(self.ctx.handle, 2, None,
                                                               mc.perm.ctypes.data_as(runtime.c_int32_p), mc.nnz, mc.n_triplets))
                    self._csr["hessc"] = mc
        return self._csr[which]

    def jacobian_csr(self, x):
        """CSR values of the constraint Jacobian (structure: ``csr_map("jac")``), gathered on the device."""
        m = self.csr_map("jac")
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        out = np.empty(m.nnz)
        self.ctx.check(self.ctx.lib.pk_eval_jac_csr(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(out)))
        return out

    def hessian_csr(self, x, lagrange, obj_factor):
        """CSR values of the lower triangle of the Hessian of the Lagrangian (``csr_map("hess")``)."""
        m = self.csr_map("hess")
        x = self._x(x)
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        lam = np.ascontiguousarray(lagrange, dtype=np.float64)
        out = np.empty(m.nnz)
        self.ctx.check(self.ctx.lib.pk_eval_hess_csr(self.ctx.handle, runtime.as_dp(x), runtime.as_dp(lam),
                                                     float(obj_factor), runtime.as_dp(out)))
        return out

    def jacobian_csr_dev(self, d_x, d_out, stream=None):
        self.csr_map("jac")
        self._invalidate_x()        # (the triplets pass through the context's J buffer)
        self.ctx.check(self.ctx.lib.pk_eval_jac_csr_dev(self.ctx.handle, d_x, d_out, stream))

    def hessian_csr_dev(self, d_x, d_lam, sigma, d_out, stream=None):
        self.csr_map("hess")
        self._invalidate_x()
        self.ctx.check(self.ctx.lib.pk_eval_hess_csr_dev(self.ctx.handle, d_x, d_lam, float(sigma), d_out, stream))

    def gather_csr_dev(self, which, d_triplets, d_out, stream=None):
        """Triplet values already on the device (e.g. from ``cycle_dev``) -> CSR values."""
        self.csr_map(which)
        self.ctx.check(self.ctx.lib.pk_gather_csr_dev(self.ctx.handle, 0 if which == "jac" else 1, d_triplets, d_out,
                                                      stream))

    def cycle(self, x, lagrange, obj_factor):
        """All five outputs on the same x from ONE call and ONE launch (pk_cycle): returns (f, grad, g, J, H).  Staging and
        landing are the callbacks' (pk_callback_cycle): x and lambda through the pinned staging buffers, [J | grad f | g] into a
        landing block whose x-independent entries are already there, H into a pinned array of the Hessian ring.  Afterwards
        the iterate is the prepared one: the callbacks on the same x are served from what has landed."""
        x = self._x(x)
        lam = self._lam(lagrange)
        p = self.plan
        it = None if self._jac_compact else self._next_block()      # (the one-launch cycle writes the reference layout)
        hit = None if self.zero_copy else self._ring_h.take_item()
        if it is not None and hit is not None:
            rc = self.ctx.lib.pk_callback_cycle(self.ctx.handle, x.ctypes.data, lam.ctypes.data, float(obj_factor), it.address,
                                                hit.address, self._a_f)
            if rc:
                self.ctx.check(rc)
            self._became_current(it)
            v = self._cur_views
            self._handed = {1, 2, 3}
            return np.float64(self._c_f.value), v[1], v[2], v[3], hit.array
        # no landing block to be had (zero-copy mode, the caller holds on to every block, compact Jacobian layout): plain
        # arrays, one copy per output
        self._invalidate_x()        # the context's x / result buffers are about to hold another iterate
        J, grad, g, H = np.empty(p.nnz_J), np.empty(p.n), np.empty(p.m), np.empty(p.nnz_H)
        f = np.empty(1)
        dp = runtime.as_dp
        self.ctx.check(self.ctx.lib.pk_eval_cycle(self.ctx.handle, dp(x), dp(lam), float(obj_factor), dp(f), dp(grad),
                                                  dp(g), dp(J), dp(H)))
        return np.float64(f[0]), grad, g, J, H

    # ------------------------------------------------------------------ device-pointer API
    def cycle_dev(self, d_x, d_lam, sigma, d_f, d_grad, d_g, d_jac, d_hess, stream=None):
        """Enqueue one full callback cycle on device pointers (ints); no synchronization."""
        self.ctx.check(self.ctx.lib.pk_eval_cycle_dev(self.ctx.handle, d_x, d_lam, float(sigma), d_f, d_grad, d_g,
                                                      d_jac, d_hess, stream))

    def sync(self, stream=None):
        self.ctx.check(self.ctx.lib.pk_sync(self.ctx.handle, stream))

    def profile(self, enable=True, period=1):
        """Time the kernels whose bit is set in ``enable`` with HIP events; only every ``period``-th launch."""
        # (the x-CALLBACKS of a profiled context run pk_xall, which the library refuses for a model with a wide phase -- error 27,
        #  DESIGN.md section 11; profiling around device-resident cycles, what bench.py does, never reaches that kernel)
        self.ctx.check(self.ctx.lib.pk_profile_sampling(self.ctx.handle, int(period)))
        self.ctx.check(self.ctx.lib.pk_profile(self.ctx.handle, int(enable)))

    def set_cycle_graph(self, enable=True):
        """Replay the fused cycle from a cached hipGraph (measured slower than plain launches on MI355X /
        ROCm 7.2 -- DESIGN.md section 5 -- so it is off by default)."""
        self.ctx.check(self.ctx.lib.pk_set_cycle_graph(self.ctx.handle, int(bool(enable))))

    #: what the guarded entry points say (the reason in full: DESIGN.md section 11; the library's own guard is error 27)
    _WIDE_SEQUENTIAL = ("is not available for a model with a wide phase (more than {nx} states): it runs the values role of such a "
                        "phase with its dynamics passes inside one wave (pk_xall), a form that returned wrong f / grad / g for "
                        "some models and raised GPU memory faults -- an open defect (DESIGN.md section 11).  The default "
                        "one-launch cycle and the five callbacks do not use it")

    def _refuse_sequential_values_role(self, what):
        if any(self.src.wide):
            raise NotImplementedError(f"pockit_amd: {what} " + self._WIDE_SEQUENTIAL.format(nx=self.src.wide_nx))

    def set_cycle_mode(self, single_launch=True):
        """True (default): one cycle = one launch (pk_cycle).  False: pk_xall, then pk_hess with the reductions."""
        if not single_launch:
            self._refuse_sequential_values_role("the two-launch form of the cycle")
        self.ctx.check(self.ctx.lib.pk_set_cycle_mode(self.ctx.handle, int(bool(single_launch))))

    def set_cycle_layout(self, jacobian_compact=False, hessian_compact=False):
        """What ``cycle_dev`` writes into its J / H buffers: the reference's triplet lists (default) or the compact layouts
        (``plan.nnz_Jc`` / ``plan.nnz_Hc`` values) -- from the same single launch."""
        if hessian_compact and not self.src.compact:
            raise NotImplementedError("compact Hessian layout is not available for this model")
        if jacobian_compact:
            self.plan.jacc  # noqa: B018
        self.ctx.check(self.ctx.lib.pk_set_cycle_layout(self.ctx.handle, int(bool(jacobian_compact)), int(bool(hessian_compact))))
        self._cycle_layout = (bool(jacobian_compact), bool(hessian_compact))

    def profile_read(self):
        """{kernel name: (launches, total_ms)} accumulated while profiling was enabled."""
        out = {}
        for k, name in enumerate(runtime.KERNELS):
            n, ms = C.c_int64(), C.c_double()
            self.ctx.check(self.ctx.lib.pk_profile_read(self.ctx.handle, k, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out
