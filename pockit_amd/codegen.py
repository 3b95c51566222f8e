"""HIP code generator: evaluation plan -> gfx950 device code for one model.

The generated translation unit contains only *model* code -- straight-line fp64 expression
evaluation with joint common-subexpression elimination per function group -- wrapped in one
``struct`` per phase, plus thin ``extern "C" __global__`` wrappers that instantiate the
hand-written kernel templates of ``csrc/pk_kernels.hip.h`` (tiling, LDS staging, coalesced
streaming, reductions) for those structs.  It is the MI355X counterpart of the reference's
function compiler (/root/reference/pockit/base/fastfunc.py:271-308: SymPy -> CSE -> NumPy source
-> numba.njit), with two differences: all functions of a phase share one CSE per callback
(the reference compiles each F/G/H separately), and the chain rule is already folded into the
emitted expressions (pockit_amd/transcription.py), so no per-callback list plumbing remains.

The generated source does not depend on the mesh (only on the model structure), so a compiled
code object is reused across mesh refinements; it is cached by source hash (hipbuild.py).
"""
from __future__ import annotations

import hashlib
import os

import sympy as sp
from sympy.printing.c import C99CodePrinter
from sympy.printing.repr import ReprPrinter

from .model import FIXED, FREE, FUNC
from .transcription import DT, SIG, TAU, WQ, SystemPlan, lam_path, lam_sys, ltb_sym, ltf_sym, mu_sym


class _CPrinter(C99CodePrinter):
    """fp64 C with small integer powers expanded (as the reference does, fastfunc.py:180)."""

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        bs = self._print(b)
        if e.is_Integer:
            n = int(e)
            if 1 <= abs(n) <= 3:
                prod = "*".join([f"({bs})"] * abs(n))
                return f"({prod})" if n > 0 else f"(1.0/({prod}))"
        if e == sp.S.Half:
            return f"sqrt({bs})"
        if e == -sp.S.Half:
            return f"(1.0/sqrt({bs}))"
        return f"pow({bs}, {self._print(e)})"

    def _print_Integer(self, expr):
        return f"{int(expr)}.0"

    def _print_Rational(self, expr):
        return f"({int(expr.p)}.0/{int(expr.q)}.0)"


_printer = _CPrinter({"precision": 17})


def ccode(expr):
    return _printer.doprint(sp.sympify(expr))


class _Names:
    """SymPy symbol -> C identifier / lvalue expression."""

    def __init__(self):
        self.map = {}

    def add(self, sym, cname):
        self.map[sym] = sp.Symbol(cname)

    def apply(self, expr):
        return sp.sympify(expr).xreplace(self.map)


def _fuse_sincos(exprs):
    """Replace sin(u) / cos(u) of the same argument by a pair of symbols filled by ONE ``sincos`` call (one
    argument reduction instead of two).  Returns (rewritten exprs, [(u, sin symbol, cos symbol)])."""
    sines, cosines = set(), set()
    for e in exprs:
        sines |= {f.args[0] for f in e.atoms(sp.sin)}
        cosines |= {f.args[0] for f in e.atoms(sp.cos)}
    both = sorted(sines & cosines, key=sp.default_sort_key)
    if not both:
        return exprs, []
    pairs, repl = [], {}
    for k, u in enumerate(both):
        sn, cs = sp.Symbol(f"pk_sn{k}"), sp.Symbol(f"pk_cs{k}")
        pairs.append((u, sn, cs))
        repl[sp.sin(u)], repl[sp.cos(u)] = sn, cs
    return [e.xreplace(repl) for e in exprs], pairs


class _MemoRepr(ReprPrinter):
    """``sympy.srepr`` with every sub-expression printed once per process: the same strings (the emit keys on disk stay
    valid), but the derivative entries of a model share most of their sub-trees, and ordering an Add's terms for printing
    is what a warm model set-up spent its time in (humanoid: 22 of 25 s of compile_plan before)."""

    def __init__(self):
        super().__init__()
        self.memo = {}

    def _print(self, expr, **kwargs):
        if kwargs or not isinstance(expr, sp.Basic):
            return super()._print(expr, **kwargs)
        got = self.memo.get(expr)
        if got is None:
            got = self.memo[expr] = super()._print(expr)
            if len(self.memo) > 400_000:
                self.memo.clear()
        return got


_SREPR = _MemoRepr()
_EMIT_MEMO = {}
_EMIT_DIR = os.path.join(os.environ.get("POCKIT_AMD_CACHE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")),
                         "emit")


def _emit_body(outputs, base, names, indent="    "):
    """C statements computing ``outputs`` = [(lvalue, expr)], expanding placeholder symbols through
    one joint CSE of their defining expressions ``base`` (placeholder -> expr).

    The result depends only on the expressions (not on the mesh), and the CSE + printing is the expensive part of
    setting a model up again after a mesh refinement (humanoid: 3.4 s of 5.5 s): bodies are memoised by the
    structural representation of their inputs, in memory and under ``_cache/emit/``."""
    outs = [(lv, sp.sympify(e)) for lv, e in outputs]
    needed, frontier = [], set()
    for _, e in outs:
        frontier |= {s for s in e.free_symbols if s in base}
    needed = sorted(frontier, key=lambda s: s.name)
    defs = [names.apply(base[k]) for k in needed]
    finals = [names.apply(e) for _, e in outs]
    key = hashlib.sha256("\x1f".join(
        ["emit-v1", indent] + [k.name for k in needed] + [_SREPR.doprint(e) for e in defs] +
        [lv for lv, _ in outs] + [_SREPR.doprint(e) for e in finals]).encode()).hexdigest()[:32]
    if key in _EMIT_MEMO:
        return _EMIT_MEMO[key]
    path = os.path.join(_EMIT_DIR, key + ".c")
    if os.path.exists(path):
        with open(path) as fh:
            _EMIT_MEMO[key] = fh.read()
        try:
            os.utime(path)                   # (tools/prune_cache.py drops the bodies nothing has asked for lately)
        except OSError:
            pass
        return _EMIT_MEMO[key]
    text = _emit_body_uncached(outs, needed, defs, finals, indent)
    _EMIT_MEMO[key] = text
    try:
        os.makedirs(_EMIT_DIR, exist_ok=True)
        staged = f"{path}.{os.getpid()}.part"     # (ranks that start cold together write the same bodies)
        with open(staged, "w") as fh:
            fh.write(text)
        os.replace(staged, path)
    except OSError:
        pass                                 # (read-only install: the in-memory memo still serves this process)
    return text


def _emit_body_uncached(outs, needed, defs, finals, indent):
    fused, pairs = _fuse_sincos(defs + finals)
    defs, finals = fused[: len(defs)], fused[len(defs):]
    lines = []
    for u, sn, cs in pairs:
        lines.append(f"{indent}double {sn.name}, {cs.name};")
        lines.append(f"{indent}sincos({ccode(u)}, &{sn.name}, &{cs.name});")
    if needed:
        repl, red = sp.cse(defs, optimizations="basic", symbols=sp.numbered_symbols("c_"))
        for sym, e in repl:
            lines.append(f"{indent}const double {sym.name} = {ccode(e)};")
        for k, e in zip(needed, red):
            lines.append(f"{indent}const double {k.name} = {ccode(e)};")
    for (lv, _), e in zip(outs, finals):
        lines.append(f"{indent}{lv} = {ccode(e)};")
    return "\n".join(lines)


def ctable_(name, values):
    vals = ", ".join(str(v) for v in values) or "0"
    return f"  __host__ __device__ static constexpr int {name}(int g) {{ constexpr int t[] = {{{vals}}}; return t[g]; }}"


def split_groups(n_i, n_n, cap):
    """Partition the derivative entries a role evaluates per node -- ``n_i`` I-expanded segments (staged in LDS, streamed
    K^2-fold) and ``n_n`` per-node segments (stored directly) -- into groups a wave evaluates, stages and streams one after
    the other: [(i0, ni, n0, nn)].  One group while the set is small (the kernels then run exactly the single-pass code of
    rounds 1-3); otherwise balanced runs of at most ``cap`` I segments, then runs of at most ``cap`` per-node segments.  The
    reference has no such notion (phasebase.py:1211-1337 loops over any number of entries): the groups bound what ONE pass
    keeps in LDS (64 doubles per staged segment and wave), in VGPRs (two values per segment in the streaming loop) and in
    SGPRs (one run pointer per segment), so a model's size no longer meets a wall of the hardware."""
    if n_i <= cap and n_i + n_n <= cap + cap // 2:
        return [(0, n_i, 0, n_n)]
    out = []
    for total, kind in ((n_i, 0), (n_n, 1)):
        if total == 0:
            continue
        ng = -(-total // cap)
        lo = 0
        for g in range(ng):
            cnt = total // ng + (1 if g < total % ng else 0)
            out.append((lo, cnt, 0, 0) if kind == 0 else (0, 0, lo, cnt))
            lo += cnt
    return out


def split_chunks(n, cap):
    """Balanced runs of at most ``cap`` per-node outputs: [(lo, count)]; one run while n <= cap + cap / 2."""
    if n <= cap + cap // 2:
        return [(0, n)]
    ng = -(-n // cap)
    out, lo = [], 0
    for g in range(ng):
        cnt = n // ng + (1 if g < n % ng else 0)
        out.append((lo, cnt))
        lo += cnt
    return out


def split_even(n, cap):
    """Balanced runs of AT MOST ``cap`` items: [(lo, count)]."""
    ng = max(1, -(-n // cap))
    out, lo = [], 0
    for g in range(ng):
        cnt = n // ng + (1 if g < n % ng else 0)
        out.append((lo, cnt))
        lo += cnt
    return out


def mu_chunks(needs, cap, mu_cap, mu_max):
    """Passes of a WIDE model's compact Hessian: consecutive runs of per-node outputs, each with at most ``cap`` outputs whose
    expressions refer to at most ``mu_cap`` distinct contracted multipliers mu_i (``needs[e]`` = the states output e refers
    to) -- a single output may refer to up to ``mu_max``.  Returns [(lo, count, [states])] or None when an output refers to
    more than ``mu_max`` states (the compact layout is then not offered for the model)."""
    out, lo, cur = [], 0, set()
    for e, need in enumerate(needs):
        if len(need) > mu_max:
            return None
        if e > lo and (e - lo >= cap or len(cur | need) > mu_cap):
            out.append((lo, e - lo, sorted(cur)))
            lo, cur = e, set()
        cur = cur | need
    if len(needs) > lo or not out:
        out.append((lo, len(needs) - lo, sorted(cur)))
    return out


class ModelSource:
    """Generates the HIP source of one SystemPlan; exposes the compile-time counts the runtime
    tables must agree with."""

    # A phase with more states than this is WIDE (= PK_WIDE_NX of pk_kernels.hip.h): nothing a wave keeps in LDS or in
    # registers may then grow with the number of states -- the dynamics values and defect rows are produced in passes over
    # chunks of at most WIDE_CHUNK states, the Hessian passes stage the multiplier rows of their own states only, the compact
    # Hessian contracts the multipliers its pass refers to, and every pass fetches the node arguments it reads itself.
    WIDE_NX = 16
    WIDE_CHUNK = 16
    WIDE_NX_LOW = 8        # (see __init__: wide_nx)
    MAX_ROWS = 256         # sums over all nodes pk_cycle's finalize workgroup takes: one thread each (fin_handoff)
    MU_MAX = 64            # most states ONE entry of the compact Hessian may refer to (their multiplier rows are staged together)
    LDS_LIMIT = 160 * 1024

    # most segments ONE pass of a tile wave evaluates, stages and streams (split_groups); POCKIT_AMD_GROUP_CAP overrides,
    # and the evaluator halves it for a model whose kernels would still spill registers (hipbuild.resource_usage)
    GROUP_CAP = 32

    # outputs of one launch beyond this many bytes do not stay in the 256 MiB (268 MB) Infinity Cache (MALL) any more: x,
    # lambda and the tables live there too, and the step was measured between 209 and 262 MB of outputs
    MALL_BYTES = 240_000_000

    def __init__(self, plan: SystemPlan, sharded: bool = False, output_share: float = 1.0, group_cap=None, wide_nx=None):
        """``wide_nx``: a phase with more states than this is evaluated the WIDE way (default WIDE_NX = 16; evaluator.compile_plan
        lowers it to WIDE_NX_LOW for a pass-parallel model whose cycle kernel would otherwise leave room for one wave per SIMD
        only: the state-chunked values role needs far fewer registers)."""
        self.plan = plan
        self.wide_nx = int(wide_nx or self.WIDE_NX)
        self.group_cap = int(group_cap or os.environ.get("POCKIT_AMD_GROUP_CAP") or self.GROUP_CAP)
        self.groups = {}          # (callback, phase) -> [(i0, ni, n0, nn)]
        # Bytes one launch of the cycle writes on THIS evaluator (a shard of a multi-GPU run writes its share): decides the
        # cache policy of the streaming stores, the third mesh fact the code object depends on (after PK_TAB_CAP, PK_BIG).
        self.output_bytes = 8.0 * (1 + plan.n + plan.m + plan.nnz_J + plan.nnz_H) * float(output_share)
        # sharded: the finalize workgroup of pk_cycle carries the in-launch exchange of the partial sums between the GPUs.
        # Single-GPU code objects are compiled without it (its mere presence cost the 12k-node cycle 3 %).
        self.sharded = bool(sharded)
        self.nphase = len(plan.phase_plans)
        nI = len(plan.I_syms)
        # integrals evaluated by the pre-pass: those any system-level function references
        self.int_needed = sorted(set(plan.which_o) | set(plan.which_c))
        self.int_local = [[a for a in self.int_needed if plan.I_owner[a][0] == k] for k in range(self.nphase)]
        self.nred = max([1] + [len(v) for v in self.int_local] + [len(v) for v in plan.grad_red_slots])
        # edge lists: fixed order [phase0 front, phase0 back, phase1 front, ..., system]
        self.list_keys = []
        for k in range(self.nphase):
            self.list_keys += [("f", k), ("b", k)]
        self.list_keys.append(("s",))
        self.list_off = {}
        # intervals with more points than a wavefront has lanes (64 < K <= 256) are evaluated by a whole workgroup: code
        # compiled into the object only when the mesh has such an interval (PK_BIG)
        self.big = any(int(pp.layout.K.max()) > 64 for pp in plan.phase_plans)
        self.compact = not plan.outer
        self.compact_j = True
        self.wide = [pp.nx > self.wide_nx for pp in plan.phase_plans]
        chunk = self.WIDE_CHUNK if self.wide_nx >= self.WIDE_NX else self.WIDE_NX_LOW
        self.dyn_chunks = [split_even(pp.nx, min(chunk, self.group_cap)) if self.wide[k] else []
                           for k, pp in enumerate(plan.phase_plans)]
        self.hc_passes = {}
        if self.compact:
            for k, pp in enumerate(plan.phase_plans):
                if not self.wide[k]:
                    continue
                mus = {mu_sym(i): i for i in range(pp.nx)}
                needs = [{mus[sy] for sy in sp.sympify(sg.expr).free_symbols if sy in mus} for sg in plan.hessc.segs[k]]
                passes = mu_chunks(needs, self.group_cap, self.group_cap, self.MU_MAX)
                if passes is None:        # an entry that couples more than MU_MAX states: reference layout only
                    self.compact = False
                    break
                self.hc_passes[k] = passes
        for cbname in ("jac", "hess", "aux") + (("hessc",) if self.compact else ()) + (("jacc",) if self.compact_j else ()):
            cb = getattr(plan, cbname)
            off, table = 0, {}
            for key in self.list_keys:
                table[key] = off
                off += len(cb.lists.get(key, []))
            table["total"] = off
            self.list_off[cbname] = table
        self.source = self._generate()
        self.hash = hashlib.sha256(self.source.encode()).hexdigest()[:24]

    def launch_lds_bytes(self):
        """Dynamic LDS per workgroup of every tile kernel's launch, exactly as csrc/pk_runtime.cpp sizes it (launch_raw adds
        the table blocks of the workgroup's four waves to what the entry point asks for): {kernel: bytes}.  What
        evaluator.compile_plan holds against LDS_LIMIT, and pk_load_model checks again."""
        w = 8 * 4
        tab = w * (2 * self.tab_cap + 2 * 64 + self.tab_cap // 2)
        ne = lambda cb: 8 * self.list_off[cb]["total"] if cb in self.list_off else 0  # noqa: E731
        cyc = max(w * max(self.lds_x, self.lds_h), ne("jac"), ne("hess"))
        out = {
            "pk_g": w * self.lds_g + tab,
            "pk_jac": max(w * self.lds_j, ne("jac")) + tab,
            "pk_hess": max(w * self.lds_h, ne("hess")) + tab,
            "pk_xall": max(w * self.lds_x, ne("jac")) + tab,
            "pk_cycle": max(cyc, 8 * 1024 if self.sharded else 0) + tab,
            "pk_jacc": max(w * self.lds_jc, ne("jacc")) + tab,
        }
        cycc = max(cyc, ne("jacc"))
        if self.compact:
            out["pk_hessc"] = max(w * self.lds_g, ne("hessc")) + tab
            cycc = max(cycc, w * self.lds_g, ne("hessc"))
        out["pk_cyclec"] = max(cycc, 8 * 1024 if self.sharded else 0) + tab
        return out

    def fits_lds(self):
        return max(self.launch_lds_bytes().values()) <= self.LDS_LIMIT

    # ------------------------------------------------------------------ names
    def _phase_names(self, k):
        pp = self.plan.phase_plans[k]
        p = pp.phase
        nm = _Names()
        for i, sym in enumerate(p.x + p.u + [p.t] + p.s):
            nm.add(sym, f"a[{i}]")
        self._sys_names(nm)
        nm.add(DT, "pk_dt")
        nm.add(TAU, "pk_tau")
        nm.add(WQ, "pk_w")
        for j in range(p.n_c):
            nm.add(lam_path(j), f"lp[{j}]")
        for i in range(p.n_x):
            nm.add(mu_sym(i), f"mu[{i}]")
            nm.add(ltf_sym(i), f"ltf[{i}]")
            nm.add(ltb_sym(i), f"ltb[{i}]")
        return nm

    def _sys_names(self, nm, with_s=False):
        plan = self.plan
        if with_s:
            for i, sym in enumerate(plan.s_syms):
                nm.add(sym, f"sy.s[{i}]")
        for a, sym in enumerate(plan.I_syms):
            nm.add(sym, f"sy.I[{a}]")
        nm.add(SIG, "sy.sigma")
        for c in range(plan.n_sys):
            nm.add(lam_sys(c), f"sy.lams[{c}]")
        return nm

    # ------------------------------------------------------------------ per-phase struct
    def _phase_struct(self, k):
        plan = self.plan
        pp = plan.phase_plans[k]
        p, lay = pp.phase, pp.layout
        nm = self._phase_names(k)
        base = pp.base()
        nx, nu, ns, nc = pp.nx, pp.nu, pp.ns, p.n_c
        S = []
        S.append(f"struct P{k} {{")
        S.append(f"  static constexpr int NX = {nx}, NU = {nu}, NS = {ns}, NC = {nc};")
        S.append(f"  static constexpr int NARG = {nx + nu + 1 + ns};")
        S.append(f"  static constexpr int SCHEME = {0 if lay.scheme == 'lgr' else 1};")
        S.append(f"  static constexpr int INDEX = {k};")
        wide = self.wide[k]
        dch = self.dyn_chunks[k]
        S.append(f"  static constexpr bool WIDE = {'true' if wide else 'false'};")
        S.append(f"  static constexpr int XROWS = {max(c[1] for c in dch) if wide else nx}, D_NG = {len(dch)};")

        # ---- boundary substitution (reference: phasebase.py:830-847) ----
        snm = _Names()
        for i, sym in enumerate(p.s):
            snm.add(sym, f"s[{i}]")

        def bc_expr(info, cur):
            if info.t == FREE:
                return cur
            if info.t == FIXED:
                return ccode(sp.Float(info.v))
            return ccode(snm.apply(info.v.expr))

        S.append("  __device__ static __forceinline__ double t0(const double* __restrict__ xp, int L, const double* s) {")
        S.append(f"    return {bc_expr(p.info_t_0, 'xp[L - 2]')};\n  }}")
        S.append("  __device__ static __forceinline__ double tf(const double* __restrict__ xp, int L, const double* s) {")
        S.append(f"    return {bc_expr(p.info_t_f, 'xp[L - 1]')};\n  }}")
        for which, infos in (("front", p.info_bc_0), ("back", p.info_bc_f)):
            S.append(f"  __device__ static __forceinline__ void fix_{which}(double* a, const double* s) {{")
            for i, info in enumerate(infos):
                if info.t != FREE:
                    S.append(f"    a[{i}] = {bc_expr(info, '')};")
            S.append("  }")
            S.append(f"  __device__ static __forceinline__ double {which}_value(int i, double cur, const double* s) {{")
            S.append("    switch (i) {")
            for i, info in enumerate(infos):
                if info.t != FREE:
                    S.append(f"      case {i}: return {bc_expr(info, '')};")
            S.append("      default: return cur;\n    }\n  }")

        sig_node = ("const double* __restrict__ a, double pk_tau, double pk_dt, double pk_w, "
                    "const PkSys& sy, const double* __restrict__ lp")

        # ---- values: dynamics + path (eval_g), integrands (pre-pass) ----
        outs = [(f"o[{i}]", fr.F) for i, fr in enumerate(pp.dyn)]
        outs += [(f"o[{nx + j}]", fr.F) for j, fr in enumerate(pp.path)]
        S.append(f"  static constexpr int G_NOUT = {max(nx + nc, 1)};")
        S.append("  __device__ static __forceinline__ void mid_g(const double* __restrict__ a, double* __restrict__ o) {")
        S.append(_emit_body(outs, base, nm))
        S.append("  }")
        if wide:
            S.append(ctable_("D_c0", [c[0] for c in dch]))
            S.append(ctable_("D_cn", [c[1] for c in dch]))
            for gi, (lo, cnt) in enumerate(dch):
                S.append(f"  __device__ static __forceinline__ void mid_dyn_g(pk::Grp<{gi}>, const double* __restrict__ a, "
                         "double* __restrict__ o) {")
                S.append(_emit_body([(f"o[{e}]", fr.F) for e, fr in enumerate(pp.dyn[lo:lo + cnt])], base, nm))
                S.append("  }")
            # the states / controls a chunk's dynamics read (the mesh error kernel interpolates only those: err_args)
            argsym = {sym: i for i, sym in enumerate(p.x + p.u)}
            used = []
            for lo, cnt in dch:
                fs = set()
                for fr in pp.dyn[lo:lo + cnt]:
                    fs |= sp.sympify(base.get(fr.F, fr.F)).free_symbols      # (fr.F: the placeholder of the function's value)
                used.append(sorted(argsym[sy] for sy in fs if sy in argsym))
            off = [0]
            for lst in used:
                off.append(off[-1] + len(lst))
            S.append(ctable_("D_na", [len(lst) for lst in used]))
            S.append(f"  __host__ __device__ static constexpr int D_ar(int g, int k) {{ constexpr int o[] = "
                     f"{{{', '.join(str(v) for v in off)}}}; constexpr int t[] = {{{', '.join(str(i) for lst in used for i in lst) or '0'}}}; "
                     f"return t[o[g] + k]; }}")
            S.append("  __device__ static __forceinline__ void mid_path(const double* __restrict__ a, double* __restrict__ o) {")
            S.append(_emit_body([(f"o[{j}]", fr.F) for j, fr in enumerate(pp.path)], base, nm))
            S.append("  }")
        loc = self.int_local[k]
        outs = [(f"o[{r}]", pp.integ[plan.I_owner[a][1]].F) for r, a in enumerate(loc)]
        S.append(f"  static constexpr int INT_N = {len(loc)};")
        S.append("  __device__ static __forceinline__ void mid_int(const double* __restrict__ a, double* __restrict__ o) {")
        S.append(_emit_body(outs, base, nm))
        S.append("  }")

        # ---- Jacobian / Hessian segments ----
        ctable = ctable_

        for cbname, tag in (("jac", "J"), ("hess", "H"), ("aux", "A")):
            cb = getattr(plan, cbname)
            segs = cb.segs[k]
            isegs = [s for s in segs if s.kind == "I"]
            nsegs = [s for s in segs if s.kind == "N"]
            S.append(f"  static constexpr int {tag}_NI = {len(isegs)}, {tag}_NN = {len(nsegs)};")
            states = ", ".join(str(s.state) for s in isegs) or "0"
            S.append(f"  __device__ static __forceinline__ int {tag}_state(int e) {{ constexpr int t[] = {{{states}}}; return t[e]; }}")
            groups = split_groups(len(isegs), len(nsegs), self.group_cap) if cbname != "aux" else [(0, len(isegs), 0, len(nsegs))]
            self.groups[(cbname, k)] = groups
            S.append(f"  static constexpr int {tag}_NG = {len(groups)}, {tag}_GMAX = {max(g[1] for g in groups)};")
            for col, nm_ in enumerate(("gi0", "gni", "gn0", "gnn")):
                S.append(ctable(f"{tag}_{nm_}", [g[col] for g in groups]))
            if cbname == "hess":
                # the states whose defect multipliers a pass reads: all of them, or -- a WIDE phase -- the range its segments
                # belong to (the reference's order is state by state, phasebase.py:1234-1285, so a group's range is short)
                rng = []
                for i0, ni, _, _ in groups:
                    st = [s_.state for s_ in isegs[i0:i0 + ni]]
                    rng.append((min(st), max(st) - min(st) + 1) if (wide and st) else ((0, 0) if wide else (0, nx)))
                self.h_rows = getattr(self, "h_rows", {})
                self.h_rows[k] = max([0] + [r[1] for r in rng])
                S.append(ctable("H_gs0", [r[0] for r in rng]))
                S.append(ctable("H_gsn", [r[1] for r in rng]))
            if len(groups) == 1:
                outs = [(f"o[{e}]", s.expr) for e, s in enumerate(isegs)]
                outs += [(f"o[{len(isegs) + e}]", s.expr) for e, s in enumerate(nsegs)]
                S.append(f"  __device__ static __forceinline__ void mid_{cbname}({sig_node}, double* __restrict__ o) {{")
                S.append(_emit_body(outs, base, nm))
                S.append("  }")
                S.append(f"  __device__ static __forceinline__ void mid_{cbname}_g(pk::Grp<0>, {sig_node}, double* __restrict__ o) {{")
                S.append(f"    mid_{cbname}(a, pk_tau, pk_dt, pk_w, sy, lp, o);\n  }}")
            else:
                # one straight-line function PER GROUP, each with its own joint CSE: o = [the group's I segments | its N segments]
                for gi, (i0, ni, n0, nn) in enumerate(groups):
                    outs = [(f"o[{e}]", s.expr) for e, s in enumerate(isegs[i0:i0 + ni])]
                    outs += [(f"o[{ni + e}]", s.expr) for e, s in enumerate(nsegs[n0:n0 + nn])]
                    S.append(f"  __device__ static __forceinline__ void mid_{cbname}_g(pk::Grp<{gi}>, {sig_node}, double* __restrict__ o) {{")
                    S.append(_emit_body(outs, base, nm))
                    S.append("  }")
            for w, wname in (("f", "front"), ("b", "back")):
                exprs = cb.lists.get((w, k), [])
                outs = [(f"E[{e}]", ex) for e, ex in enumerate(exprs)]
                S.append(f"  static constexpr int {tag}_N{wname.upper()} = {len(exprs)};")
                S.append(f"  __device__ static __forceinline__ void {wname}_{cbname}({sig_node}, double* __restrict__ E) {{")
                S.append(_emit_body(outs, base, nm))
                S.append("  }")

        # ---- compact Hessian: one value per distinct (row, col) class of a node ----
        sig_c = sig_node + (", const double* __restrict__ mu, const double* __restrict__ ltf, "
                            "const double* __restrict__ ltb")
        self.hc_rows = getattr(self, "hc_rows", {})
        self.hc_rows[k] = nx
        if self.compact:
            cb = plan.hessc
            if wide:
                passes = self.hc_passes[k]
                chunks = [(lo, cnt) for lo, cnt, _ in passes]
                self.hc_rows[k] = max([1] + [len(lst) for _, _, lst in passes])
                off = [0]
                for _, _, lst in passes:
                    off.append(off[-1] + len(lst))
                flat = [i for _, _, lst in passes for i in lst]
                S.append(ctable("HC_nmu", [len(lst) for _, _, lst in passes]))
                S.append(f"  __host__ __device__ static constexpr int HC_mus(int g, int k) {{ constexpr int o[] = "
                         f"{{{', '.join(str(v) for v in off)}}}; constexpr int t[] = {{{', '.join(str(v) for v in flat) or '0'}}}; "
                         f"return t[o[g] + k]; }}")
            else:
                chunks = split_chunks(len(cb.segs[k]), self.group_cap)
                # (More, smaller chunks -- one per pass of the reference layout's Hessian role, to use all of its workgroups in a
                #  pass-parallel launch -- were measured SLOWER: every chunk repeats the contraction of the multipliers and loses
                #  the subexpressions it shared; drone_stabilization 2000 x 4 12.2 -> 14.9 us, rocket_powered_descent 11.3 -> 12.7,
                #  orbit_transfer 8.6 -> 10.3: profiles/r05_d_large_models_at_8k_nodes.txt)
            self.groups[("hessc", k)] = chunks
            S.append(f"  static constexpr int HC_NN = {len(cb.segs[k])}, HC_NG = {len(chunks)}, HC_LROWS = {self.hc_rows[k]};")
            S.append(ctable("HC_c0", [c[0] for c in chunks]))
            S.append(ctable("HC_cn", [c[1] for c in chunks]))
            for gi, (lo, cnt) in enumerate(chunks):
                nmg = nm
                if wide:        # (row kk of the pass's staged multiplier rows and mu[kk] belong to state lst[kk])
                    nmg = _Names()
                    nmg.map = dict(nm.map)
                    for kk, i in enumerate(self.hc_passes[k][gi][2]):
                        nmg.add(mu_sym(i), f"mu[{kk}]")
                S.append(f"  __device__ static __forceinline__ void mid_hessc_g(pk::Grp<{gi}>, {sig_c}, double* __restrict__ o) {{")
                S.append(_emit_body([(f"o[{e}]", sg.expr) for e, sg in enumerate(cb.segs[k][lo:lo + cnt])], base, nmg))
                S.append("  }")
            for w, wname in (("f", "front"), ("b", "back")):
                exprs = cb.lists.get((w, k), [])
                S.append(f"  __device__ static __forceinline__ void {wname}_hessc({sig_c}, double* __restrict__ E) {{")
                S.append(_emit_body([(f"E[{e}]", ex) for e, ex in enumerate(exprs)], base, nm))
                S.append("  }")
        else:
            S.append(f"  static constexpr int HC_NN = 0, HC_NG = 1, HC_LROWS = {nx};")
            S.append(ctable("HC_nmu", [0]))
            S.append("  __host__ __device__ static constexpr int HC_mus(int, int) { return 0; }")
            S.append(ctable("HC_c0", [0]))
            S.append(ctable("HC_cn", [0]))
            S.append(f"  __device__ static __forceinline__ void mid_hessc_g(pk::Grp<0>, {sig_c}, double* __restrict__ o) {{}}")

        # ---- compact Jacobian: expanded (per-node column), contracted (dense column) and per-node entries ----
        if self.compact_j:
            cb = plan.jacc
            segs = cb.segs[k]
            ci = [sg for sg in segs if sg.kind == "I"]
            cd = [sg for sg in segs if sg.kind == "D"]
            cn = [sg for sg in segs if sg.kind == "N"]
            S.append(f"  static constexpr int JC_NI = {len(ci)}, JC_ND = {len(cd)}, JC_NN = {len(cn)};")
            cap = self.group_cap
            if len(ci) + len(cd) <= cap and len(ci) + len(cd) + len(cn) <= cap + cap // 2:
                jgroups = [(-1, 0, 0)]                      # one pass over everything (the single-pass kernel code)
            else:                                           # runs of one kind each: expanded, dense-column, per-node
                jgroups = [(kind, lo, cnt) for kind, lst in ((0, ci), (1, cd), (2, cn)) for lo, cnt in split_chunks(len(lst), cap) if cnt]
            self.groups[("jacc", k)] = jgroups
            S.append(f"  static constexpr int JC_NG = {len(jgroups)}, "
                     f"JC_GMAX = {len(ci) + len(cd) if jgroups[0][0] < 0 else max([1] + [c for kd, _, c in jgroups if kd < 2])};")
            S.append(ctable("JC_gk", [g[0] for g in jgroups]))
            S.append(ctable("JC_g0", [g[1] for g in jgroups]))
            S.append(ctable("JC_gn", [g[2] for g in jgroups]))
            if jgroups[0][0] < 0:
                S.append(f"  __device__ static __forceinline__ void mid_jacc({sig_node}, double* __restrict__ o) {{")
                S.append(_emit_body([(f"o[{e}]", sg.expr) for e, sg in enumerate(ci + cd + cn)], base, nm))
                S.append("  }")
                for wname, attr in (("front", "front"), ("back", "back")):
                    S.append(f"  __device__ static __forceinline__ void {wname}_jacc_dense({sig_node}, double* __restrict__ o) {{")
                    S.append(_emit_body([(f"o[{e}]", getattr(sg, attr)) for e, sg in enumerate(cd)], base, nm))
                    S.append("  }")
                S.append("  __device__ static __forceinline__ void jacc_tdense(const double* __restrict__ a, "
                         "double* __restrict__ otf, double* __restrict__ otb) {")
                S.append(_emit_body([(f"otf[{e}]", sg.tfront) for e, sg in enumerate(cd)] +
                                    [(f"otb[{e}]", sg.tback) for e, sg in enumerate(cd)], base, nm))
                S.append("  }")
            else:
                for gi, (kind, lo, cnt) in enumerate(jgroups):
                    part = (ci, cd, cn)[kind][lo:lo + cnt]
                    S.append(f"  __device__ static __forceinline__ void mid_jacc_g(pk::Grp<{gi}>, {sig_node}, double* __restrict__ o) {{")
                    S.append(_emit_body([(f"o[{e}]", sg.expr) for e, sg in enumerate(part)], base, nm))
                    S.append("  }")
                    if kind != 1:
                        continue
                    for wname, attr in (("front", "front"), ("back", "back")):
                        S.append(f"  __device__ static __forceinline__ void {wname}_jacc_dense_g(pk::Grp<{gi}>, {sig_node}, double* __restrict__ o) {{")
                        S.append(_emit_body([(f"o[{e}]", getattr(sg, attr)) for e, sg in enumerate(part)], base, nm))
                        S.append("  }")
                    S.append(f"  __device__ static __forceinline__ void jacc_tdense_g(pk::Grp<{gi}>, const double* __restrict__ a, "
                             "double* __restrict__ otf, double* __restrict__ otb) {")
                    S.append(_emit_body([(f"otf[{e}]", sg.tfront) for e, sg in enumerate(part)] +
                                        [(f"otb[{e}]", sg.tback) for e, sg in enumerate(part)], base, nm))
                    S.append("  }")
            for w, wname in (("f", "front"), ("b", "back")):
                exprs = cb.lists.get((w, k), [])
                S.append(f"  __device__ static __forceinline__ void {wname}_jacc({sig_node}, double* __restrict__ E) {{")
                S.append(_emit_body([(f"E[{e}]", ex) for e, ex in enumerate(exprs)], base, nm))
                S.append("  }")
        else:
            S.append("  static constexpr int JC_NI = 0, JC_ND = 0, JC_NN = 0, JC_NG = 1, JC_GMAX = 1;")
            S.append(f"  __device__ static __forceinline__ void mid_jacc({sig_node}, double* __restrict__ o) {{}}")
            S.append(f"  __device__ static __forceinline__ void front_jacc_dense({sig_node}, double* __restrict__ o) {{}}")
            S.append(f"  __device__ static __forceinline__ void back_jacc_dense({sig_node}, double* __restrict__ o) {{}}")
            S.append("  __device__ static __forceinline__ void jacc_tdense(const double*, double*, double*) {}")

        # ---- dense objective gradient ----
        slots = plan.grad_red_slots[k]
        S.append(f"  static constexpr int GR_NR = {len(slots)};")
        for w, wname in (("f", "front"), ("m", "mid"), ("b", "back")):
            S.append(f"  __device__ static __forceinline__ void {wname}_grad({sig_node}, double* __restrict__ ov, double* __restrict__ orr) {{")
            if w in pp.where:
                outs = [(f"ov[{a}]", e) for a, e in enumerate(plan.grad_var[k][w])]
                outs += [(f"orr[{r}]", plan.grad_red[k][w].get(s, sp.Integer(0))) for r, s in enumerate(slots)]
                S.append(_emit_body(outs, base, nm))
            S.append("  }")
        # ---- fused x-callbacks: g values, Jacobian segments, gradient entries, integrands; ONE CSE ----
        jsegs = plan.jac.segs[k]
        ji = [sg for sg in jsegs if sg.kind == "I"]
        jn = [sg for sg in jsegs if sg.kind == "N"]
        outs = [(f"og[{i}]", fr.F) for i, fr in enumerate(pp.dyn)]
        outs += [(f"og[{nx + j}]", fr.F) for j, fr in enumerate(pp.path)]
        outs += [(f"oj[{e}]", sg.expr) for e, sg in enumerate(ji + jn)]
        outs += [(f"ov[{a}]", e) for a, e in enumerate(plan.grad_var[k]["m"])]
        outs += [(f"ot[{r}]", plan.grad_red[k]["m"].get(sl, sp.Integer(0))) for r, sl in enumerate(slots)]
        outs += [(f"op[{r}]", pp.integ[plan.I_owner[a][1]].F) for r, a in enumerate(loc)]
        if len(self.groups[("jac", k)]) == 1 and not wide:
            S.append("  __device__ static __forceinline__ void mid_xall(const double* __restrict__ a, double pk_tau, "
                     "double pk_dt, double pk_w, const PkSys& sy, double* __restrict__ og, double* __restrict__ oj, "
                     "double* __restrict__ ov, double* __restrict__ ot, double* __restrict__ op) {")
            S.append("    const double* lp = nullptr; (void)lp;")
            S.append(_emit_body(outs, base, nm))
            S.append("  }")
        else:
            # grouped Jacobian: the x-kernels evaluate the VALUES part here (g, gradient entries, integrands) and the Jacobian
            # segments group by group with mid_jac_g, as pk_jac does
            # (the dynamics values go straight into the staging rows -- ogl[i * ogs] is row i at this lane / node, ogs the row
            #  length --, so that a wide model does not carry n_x more register pairs to the end of the function)
            # (a WIDE phase: no dynamics values here at all -- mid_dyn_g, chunk of states by chunk of states)
            vouts = [((f"ogl[{int(lv[3:-1])} * ogs]" if (lv.startswith("og[") and int(lv[3:-1]) < nx) else lv), e)
                     for lv, e in outs if not lv.startswith("oj[") and not (wide and lv.startswith("og[") and int(lv[3:-1]) < nx)]
            S.append("  __device__ static __forceinline__ void mid_xval(const double* __restrict__ a, double pk_tau, "
                     "double pk_dt, double pk_w, const PkSys& sy, double* __restrict__ ogl, int ogs, double* __restrict__ og, "
                     "double* __restrict__ ov, double* __restrict__ ot, double* __restrict__ op) {")
            S.append("    const double* lp = nullptr; (void)lp;")
            S.append(_emit_body(vouts, base, nm))
            S.append("  }")
        S.append("};")
        return "\n".join(S)

    # ------------------------------------------------------------------ system-level functions
    def _system_functions(self):
        plan = self.plan
        nm = self._sys_names(_Names(), with_s=True)
        S = []
        S.append("__device__ static __forceinline__ double sys_objective(const PkSys& sy) {")
        S.append(f"  return {ccode(nm.apply(plan.F_o.expr))};\n}}")
        S.append("__device__ static __forceinline__ void sys_constraints(const PkSys& sy, double* __restrict__ g) {")
        S.append(_emit_body([(f"g[{c}]", f.expr) for c, f in enumerate(plan.F_c)], {}, nm, "  "))
        S.append("}")
        S.append("__device__ static __forceinline__ void sys_grad_static(const PkSys& sy, double* __restrict__ gs) {")
        S.append(_emit_body([(f"gs[{i}]", plan.grad_static.get(i, sp.Integer(0))) for i in range(plan.n_s)],
                            {}, nm, "  "))
        S.append("}")
        for cbname in ("jac", "hess", "aux") + (("hessc",) if self.compact else ()) + (("jacc",) if self.compact_j else ()):
            exprs = getattr(plan, cbname).lists.get(("s",), [])
            S.append(f"__device__ static __forceinline__ void sys_{cbname}(const PkSys& sy, double* __restrict__ E) {{")
            S.append(_emit_body([(f"E[{e}]", ex) for e, ex in enumerate(exprs)], {}, nm, "  "))
            S.append("}")
        return "\n".join(S)

    # ------------------------------------------------------------------ whole translation unit
    def _generate(self):
        plan = self.plan
        nP = self.nphase
        S = []
        S.append("// Generated by pockit_amd.codegen -- model code only; kernels are in pk_kernels.hip.h")
        # phase records the kernels take by value in their arguments: 8 unless the model has more (pk_abi.h)
        self.max_phases = 8 if nP <= 8 else nP
        if nP > 128:
            raise ValueError(f"{nP} phases: the MI355X evaluator passes at most 128 phase records in its kernel arguments "
                             "(PK_HOST_MAX_PHASES, pk_abi.h)")
        if self.max_phases != 8:
            S.append(f"#define PK_MAX_PHASES {self.max_phases}")
        S.append(f"#define PK_NRED {self.nred}")
        S.append(f"#define PK_NPHASE {nP}")
        S.append(f"#define PK_NS {max(plan.n_s, 1)}")
        S.append(f"#define PK_NSYS {max(plan.n_sys, 1)}")
        S.append(f"#define PK_NI {max(len(plan.I_syms), 1)}")
        # Capacity of the staged pattern tables: the ONE mesh fact the code object depends on -- 64 entries (every interval
        # has K <= 8, one entry per lane) or 256 (some interval has 9 <= K <= 16).  A refinement that first raises an
        # order beyond 8 costs one more compile; both variants stay cached.  POCKIT_AMD_TAB_CAP=64|256 overrides (A/B).
        need = max([1] + [max(kd.nnzI, kd.R * kd.K) for pp in plan.phase_plans for kd in pp.layout.kinds
                          if max(kd.nnzI, kd.R * kd.K) <= 256])
        self.tab_cap = 256 if need > 64 else 64
        if os.environ.get("POCKIT_AMD_TAB_CAP") in ("64", "256"):
            self.tab_cap = int(os.environ["POCKIT_AMD_TAB_CAP"])
        S.append(f"#define PK_TAB_CAP {self.tab_cap}")
        if os.environ.get("POCKIT_AMD_TRACE", "0") == "1":   # developer tracing of the wave timeline (tools/wave_trace.py)
            S.append("#define PK_TRACE 1")
        if os.environ.get("POCKIT_AMD_BIG_MFMA", "0") == "1":        # option: big intervals' products on the fp64 matrix cores
            S.append("#define PK_BIG_MFMA 1")                        # (measured slower than the VALU form, kept under test)
        if self.big:
            S.append("#define PK_BIG 1")
            S.append("//@PK_BIG_GLOBAL@")      # (decided below, once the groups of every phase are known)
        if self.sharded:
            S.append("#define PK_SHARDED 1")
        # Cache policy of the 16-byte streaming stores, by the size of what one launch writes (profiles/r03_store_policy_*.txt):
        # inside the Infinity Cache agent-scope write-through ("sc1": +20 % over plain stores, nontemporal -15 ... -30 %);
        # beyond it -- quadrotor from 262 MB = 300k nodes on -- the nontemporal form, which does not leave 300+ MB of lines
        # for the MALL to evict one by one: 93 -> 50 us at 262 MB, 120 -> 58 us at 314 MB (0.47 -> 0.97 of the roof); the
        # humanoid crosses over later (315 MB: 69 vs 87 us, 451 MB: 146 vs 117 us).
        self.stream_policy = "nt" if self.output_bytes > self.MALL_BYTES else "sc1"
        if self.stream_policy != "sc1":
            S.append(f'#define PK_STREAM_FLAGS "{self.stream_policy}"')
        S.append('#include "pk_kernels.hip.h"')
        S.append("namespace pkgen {")
        for k in range(nP):
            S.append(self._phase_struct(k))
        S.append(self._system_functions())

        arr = lambda xs: ", ".join(str(v) for v in xs) if xs else "0"  # noqa: E731

        def table_fn(name, values):
            return (f"  __device__ static __forceinline__ int {name}(int i) {{\n"
                    f"    constexpr int t[] = {{{arr(values)}}};\n    return t[i];\n  }}")

        def switch(body, indent="    "):
            out = [f"{indent}switch (phase) {{"]
            for k in range(nP):
                out.append(f"{indent}  case {k}: {body.format(P=f'P{k}', k=k)}; break;")
            out.append(f"{indent}  default: break;\n{indent}}}")
            return "\n".join(out)

        self.grouped = any(len(v) > 1 for v in self.groups.values()) or any(len(c) > 1 for c in self.dyn_chunks)
        S.append("struct Gen {")
        # LDS doubles per wave for the staged per-node values
        # (pk_g: the dynamics values [nx][64]; pk_hessc: the multiplier rows [nx][64] + the base offsets of its output runs)
        n_hc = (lambda k: len(plan.hessc.segs[k])) if self.compact else (lambda k: 0)
        # (a WIDE phase: rows of ONE chunk of states / of the multipliers one pass refers to, not of all states)
        xrows = lambda k: max(c[1] for c in self.dyn_chunks[k]) if self.wide[k] else plan.phase_plans[k].nx  # noqa: E731
        self.lds_g = 64 * max([1] + [max(xrows(k) if self.wide[k] else 0, self.hc_rows[k] + -(-n_hc(k) // 64))
                                     for k, pp in enumerate(plan.phase_plans)])
        # (rows of 64 doubles per wave; a role evaluated in groups stages one group at a time: its largest group counts)
        gmax = lambda cb, k: max(g[1] for g in self.groups[(cb, k)])  # noqa: E731
        self.lds_j = 64 * max([1] + [gmax("jac", k) for k in range(nP)])
        # Hessian: staged segment values + the tile's defect multipliers [state][row]
        self.lds_h = 64 * max([1] + [self.h_rows[k] + gmax("hess", k) for k, pp in enumerate(plan.phase_plans)])
        # (big: + the node values [NX][256] -- a workgroup-wide interval keeps rows of every state.  Where those rows do not
        #  fit a workgroup's LDS -- many states -- they live in the device staging buffer that intervals beyond 256 points use
        #  anyway: PK_BIG_GLOBAL, md.big_global / big_rows)
        self.big_rows = max([1] + [max(pp.nx * 2 + gmax("jac", k), gmax("hess", k)) for k, pp in enumerate(plan.phase_plans)]) \
            if self.big else 0
        tab_bytes = 8 * 4 * (2 * self.tab_cap + 2 * 64 + self.tab_cap // 2)
        self.big_global = bool(self.big and 8 * 256 * self.big_rows + tab_bytes > self.LDS_LIMIT)
        self.lds_x = 64 * max([1] + [(pp.nx * 2 + gmax("jac", k)) if (self.big and not self.big_global) else (xrows(k) + gmax("jac", k))
                                     for k, pp in enumerate(plan.phase_plans)])
        # (mesh error estimation: [x | u | f] rows per wave; a workgroup-wide interval adds the interpolated [x | u] rows)
        # (a WIDE phase: the dynamics values of one chunk of states -- its arguments come straight from x, pass by pass)
        self.lds_e = 64 * max([1] + [xrows(k) if self.wide[k] else 3 * pp.nx + 2 * pp.nu for k, pp in enumerate(plan.phase_plans)])
        self.lds_jc = 64 * max([1] + [(sum(1 for sg in plan.jacc.segs[k] if sg.kind in "ID") if self.groups[("jacc", k)][0][0] < 0 else
                                       max([1] + [c for kd, _, c in self.groups[("jacc", k)] if kd < 2])) for k in range(nP)]) if self.compact_j else 64
        # (a cycle launch that serves the compact Jacobian runs tile_jacc in the x-part's LDS rows)
        self.lds_x = max(self.lds_x, self.lds_jc)
        S.append(f"  static constexpr int LDS_G = {self.lds_g}, LDS_J = {self.lds_j}, LDS_H = {self.lds_h}, "
                 f"LDS_X = {self.lds_x}, LDS_E = {self.lds_e}, LDS_JC = {self.lds_jc};")
        S.append("  __device__ static __forceinline__ void interval_err(int phase, const PkArgs& A, int first, int cnt, "
                 "double* __restrict__ lds, int lane) {")
        S.append(switch("pk::interval_err<{P}>(A, first, cnt, lds, lane)"))
        S.append("  }")
        S.append("  __device__ static __forceinline__ void interval_err_big(int phase, const PkArgs& A, int first, "
                 "double* __restrict__ lds) {")
        S.append(switch("pk::interval_err_big<{P}>(A, first, lds)"))
        S.append("  }")
        targets = [(n, f"pk::tile_{n}<{{P}}>") for n in ("int", "g", "grad", "jac", "hess", "aux", "hessc", "jacc")]
        targets += [("xall", "pk::tile_xall<{P}, 0>"), ("xall1", "pk::tile_xall<{P}, 1>"),
                    ("xall2", "pk::tile_xall<{P}, 2>"), ("xall1c", "pk::tile_xall<{P}, 1, true>"),
                    # (values role of a pass-parallel cycle: a WIDE phase's dynamics passes are waves of their own)
                    ("xall1p", "pk::tile_xall<{P}, 1, false, true>"), ("xall1cp", "pk::tile_xall<{P}, 1, true, true>")]
        for name, target in targets:
            pub = name.startswith("xall")      # the x-kernels take the hand-off block of a pk_cycle launch (-1: none)
            S.append(f"  __device__ static __forceinline__ void tile_{name}(int phase, const PkArgs& A, const PkTile& tl, "
                     f"double* __restrict__ lds, double* __restrict__ wint, double* __restrict__ wgrad, int lane"
                     f"{', int pub_blk' if pub else ''}) {{")
            S.append(switch(f"{target}(A, tl, lds, wint, wgrad, lane{', pub_blk' if pub else ''})"))
            S.append("  }")
        # pk_cycle of a model evaluated in groups gives every pass of the Jacobian / Hessian role a wave of its own
        self.j_ngmax = max([1] + [len(self.groups[("jac", k)]) for k in range(nP)])
        self.h_ngmax = max([1] + [len(self.groups[("hess", k)]) for k in range(nP)])
        self.d_ngmax = max([0] + [len(c) for c in self.dyn_chunks])      # (passes over chunks of states: WIDE phases only)
        self.cycle_subs = (1 + self.j_ngmax + self.d_ngmax + self.h_ngmax) \
            if (self.j_ngmax > 1 or self.h_ngmax > 1 or self.d_ngmax > 1) else 0
        # Pass-parallel roles pay when several workgroups of the launch fit a CU: every workgroup of pk_cycle gets the launch's
        # LDS size (the largest role's rows + the table blocks).  Measured at 2000 x 4 (tools/fat_model_probe.py): orbit_transfer
        # (23 rows = 56 KB, 23 workgroups per tile block) 40 -> 22 us per cycle; drone_stabilization (47 rows = 105 KB: one
        # workgroup per CU, so nine workgroups per tile block queue up behind each other) 15 -> 20 us -- it keeps the passes
        # in one wave.  POCKIT_AMD_PASS_PARALLEL=0|1 overrides (A/B).
        wg_bytes = 8 * 4 * (max(self.lds_x, self.lds_h, self.lds_g) + 2 * self.tab_cap + 2 * 64 + self.tab_cap // 2)
        want = os.environ.get("POCKIT_AMD_PASS_PARALLEL", "auto")
        if want == "0" or (want != "1" and 2 * wg_bytes > 160 * 1024):
            # A model with a WIDE phase keeps its passes as workgroups of their own whatever the LDS says and whatever the
            # switch asks for: the SEQUENTIAL values role of such a phase (the dynamics passes inside the values wave,
            # tile_xall_grouped with PP = false) returned wrong f / grad / g / J for some models and raised GPU memory
            # faults (round 5, DESIGN.md section 11: an open defect on the compiler's SGPR-spill path of that kernel);
            # no wide model has shown it in the pass-parallel form.  Containment, not a fix.
            if any(self.wide) and self.cycle_subs:
                if want == "0":
                    import warnings

                    warnings.warn("pockit_amd: POCKIT_AMD_PASS_PARALLEL=0 is ignored for a model with a wide phase (the sequential "
                                  "form of its values role has an open defect, DESIGN.md section 11)", RuntimeWarning, stacklevel=3)
            else:
                self.cycle_subs = 0
        S.append(f"  static constexpr bool GROUPED = {'true' if self.cycle_subs else 'false'};")
        S.append(f"  static constexpr int J_NGMAX = {self.j_ngmax}, H_NGMAX = {self.h_ngmax}, D_NGMAX = {self.d_ngmax};")
        # (the compact roles of a pass-parallel launch share their passes round robin: workgroup `sub` of `stride`)
        self.hc_ngmax = max([1] + [len(self.groups[("hessc", k)]) for k in range(nP)]) if self.compact else 1
        self.jc_ngmax = max([1] + [len(self.groups[("jacc", k)]) for k in range(nP)]) if self.compact_j else 1
        S.append(f"  static constexpr int HC_NGMAX = {self.hc_ngmax}, JC_NGMAX = {self.jc_ngmax};")
        for name, fn in (("hesscp", "tile_hessc_part"), ("jaccp", "tile_jacc_part")):
            S.append(f"  __device__ static __forceinline__ void tile_{name}(int phase, int sub, int stride, const PkArgs& A, "
                     f"const PkTile& tl, double* __restrict__ lds, int lane) {{")
            S.append(switch(f"pk::{fn}<{{P}}>(sub, stride, A, tl, lds, lane)"))
            S.append("  }")
        for name, fn in (("jacg", "tile_jac_pick"), ("hessg", "tile_hess_pick"), ("dyn", "tile_dyn_pick")):
            S.append(f"  __device__ static __forceinline__ void tile_{name}(int phase, int grp, const PkArgs& A, const PkTile& tl, "
                     f"double* __restrict__ lds, int lane) {{")
            S.append(switch(f"pk::{fn}<{{P}}>(grp, A, tl, lds, lane)"))
            S.append("  }")
        if self.big:
            for name, target in (("bigx0", "pk::big_xall<{P}, 0>"), ("bigx1", "pk::big_xall<{P}, 1>"),
                                 ("bigx2", "pk::big_xall<{P}, 2>")):
                S.append(f"  __device__ static __forceinline__ void {name}(int phase, const PkArgs& A, const PkTile& tl, "
                         f"double* __restrict__ lds, double* __restrict__ wint, double* __restrict__ wgrad, int pub_blk) {{")
                S.append(switch(f"{target}(A, tl, lds, wint, wgrad, pub_blk)"))
                S.append("  }")
            S.append("  __device__ static __forceinline__ void bigh(int phase, const PkArgs& A, const PkTile& tl, "
                     "double* __restrict__ lds) {")
            S.append(switch("pk::big_hess<{P}>(A, tl, lds)"))
            S.append("  }")
            S.append("  __device__ static __forceinline__ void bigi(int phase, const PkArgs& A, const PkTile& tl, "
                     "double* __restrict__ wint) {")
            S.append(switch("pk::big_int<{P}>(A, tl, wint)"))
            S.append("  }")
            S.append("  __device__ static __forceinline__ void biga(int phase, const PkArgs& A, const PkTile& tl) {")
            S.append(switch("pk::big_aux<{P}>(A, tl)"))
            S.append("  }")
            S.append("  __device__ static __forceinline__ void bigjc(int phase, const PkArgs& A, const PkTile& tl, "
                     "double* __restrict__ lds) {")
            S.append(switch("pk::big_jacc<{P}>(A, tl, lds)"))
            S.append("  }")
        ncmax = max([1] + [pp.phase.n_c for pp in plan.phase_plans])
        for cbname, tag in (("jac", "J"), ("hess", "H"), ("aux", "A")):
            off = self.list_off[cbname]
            S.append(f"  static constexpr int {tag}_NE = {max(off['total'], 1)};")
            S.append(f"  __device__ static __forceinline__ void edge_{cbname}(int li, const PkArgs& A, const PkSys& sy, "
                     f"double* __restrict__ E) {{")
            S.append("    switch (li) {")
            for li, key in enumerate(self.list_keys):
                if key[0] == "s":
                    S.append(f"      case {li}: sys_{cbname}(sy, E + {off[key]}); break;")
                    continue
                k = key[1]
                n_here = len(getattr(plan, cbname).lists.get(key, []))
                if n_here == 0:
                    continue
                wname = "front" if key[0] == "f" else "back"
                S.append(f"      case {li}: {{ double s_[PK_NS], a[P{k}::NARG], tau, dt, w, lp[{ncmax}];")
                S.append(f"        pk::load_edge<P{k}>(A, {1 if key[0] == 'b' else 0}, s_, a, tau, dt, w, lp);")
                S.append(f"        const PkSys sy2{{s_, sy.I, sy.sigma, sy.lams}};")
                S.append(f"        P{k}::{wname}_{cbname}(a, tau, dt, w, sy2, lp, E + {off[key]}); }} break;")
            S.append("      default: break;\n    }\n  }")
        if self.compact:
            off = self.list_off["hessc"]
            S.append(f"  static constexpr int HC_NE = {max(off['total'], 1)};")
            S.append("  __device__ static __forceinline__ void edge_hessc(int li, const PkArgs& A, const PkSys& sy, "
                     "double* __restrict__ E) {")
            S.append("    switch (li) {")
            for li, key in enumerate(self.list_keys):
                if key[0] == "s":
                    S.append(f"      case {li}: sys_hessc(sy, E + {off[key]}); break;")
                    continue
                k = key[1]
                if not plan.hessc.lists.get(key):
                    continue
                wname = "front" if key[0] == "f" else "back"
                nxk = plan.phase_plans[k].nx
                S.append(f"      case {li}: {{ double s_[PK_NS], a[P{k}::NARG], tau, dt, w, lp[{ncmax}], mu[{nxk}], "
                         f"ltf[{nxk}], ltb[{nxk}];")
                S.append(f"        pk::load_edge_c<P{k}>(A, {1 if key[0] == 'b' else 0}, s_, a, tau, dt, w, lp, mu, ltf, ltb);")
                S.append(f"        const PkSys sy2{{s_, sy.I, sy.sigma, sy.lams}};")
                S.append(f"        P{k}::{wname}_hessc(a, tau, dt, w, sy2, lp, mu, ltf, ltb, E + {off[key]}); }} break;")
            S.append("      default: break;\n    }\n  }")
        else:
            S.append("  static constexpr int HC_NE = 1;")
            S.append("  __device__ static __forceinline__ void edge_hessc(int, const PkArgs&, const PkSys&, double*) {}")
        if self.compact_j:
            off = self.list_off["jacc"]
            S.append(f"  static constexpr int JC_NE = {max(off['total'], 1)};")
            S.append("  __device__ static __forceinline__ void edge_jacc(int li, const PkArgs& A, const PkSys& sy, "
                     "double* __restrict__ E) {")
            S.append("    switch (li) {")
            for li, key in enumerate(self.list_keys):
                if key[0] == "s":
                    S.append(f"      case {li}: sys_jacc(sy, E + {off[key]}); break;")
                    continue
                k = key[1]
                if not plan.jacc.lists.get(key):
                    continue
                wname = "front" if key[0] == "f" else "back"
                S.append(f"      case {li}: {{ double s_[PK_NS], a[P{k}::NARG], tau, dt, w, lp[{ncmax}];")
                S.append(f"        pk::load_edge<P{k}>(A, {1 if key[0] == 'b' else 0}, s_, a, tau, dt, w, lp);")
                S.append(f"        const PkSys sy2{{s_, sy.I, sy.sigma, sy.lams}};")
                S.append(f"        P{k}::{wname}_jacc(a, tau, dt, w, sy2, lp, E + {off[key]}); }} break;")
            S.append("      default: break;\n    }\n  }")
        else:
            S.append("  static constexpr int JC_NE = 1;")
            S.append("  __device__ static __forceinline__ void edge_jacc(int, const PkArgs&, const PkSys&, double*) {}")
        S.append(f"  static constexpr int NLISTS = {len(self.list_keys)};")
        ints = [(a, plan.I_owner[a][0], self.int_local[plan.I_owner[a][0]].index(a)) for a in self.int_needed]
        S.append(f"  static constexpr int N_INT = {len(ints)};")
        S.append(table_fn("int_global", [a for a, _, _ in ints]))
        S.append(table_fn("int_phase", [k for _, k, _ in ints]))
        S.append(table_fn("int_slot", [r for _, _, r in ints]))
        S.append(table_fn("gr_nr", [len(v) for v in plan.grad_red_slots]))
        # rows of pk_cycle's in-launch finalize: the needed integrands, then the shared gradient slots phase by phase
        rows = [(0, k, r) for _, k, r in ints]
        rows += [(1, k, r) for k, v in enumerate(plan.grad_red_slots) for r in range(len(v))]
        if len(rows) > self.MAX_ROWS:
            raise ValueError(f"{len(rows)} sums over all nodes (integrals any system function refers to + gradient slots shared by "
                             f"the nodes of a phase): the MI355X evaluator's finalize workgroup takes at most {self.MAX_ROWS} "
                             "(one thread per sum)")
        S.append(f"  static constexpr int N_ROWS = {len(rows)};")
        S.append(table_fn("row_arr", [a for a, _, _ in rows]))
        S.append(table_fn("row_phase", [k for _, k, _ in rows]))
        S.append(table_fn("row_slot", [r for _, _, r in rows]))
        S.append("  __device__ static __forceinline__ double phase_dt(int phase, const PkArgs& A) {")
        S.append(switch("return pk::phase_dt<{P}>(A)"))
        S.append("    return 0.0;\n  }")
        S.append("  __device__ static __forceinline__ double sys_objective(const PkSys& sy) { return pkgen::sys_objective(sy); }")
        S.append("  __device__ static __forceinline__ void sys_constraints(const PkSys& sy, double* g) { pkgen::sys_constraints(sy, g); }")
        S.append("  __device__ static __forceinline__ void sys_grad_static(const PkSys& sy, double* g) { pkgen::sys_grad_static(sy, g); }")
        S.append("};")
        S.append("}  // namespace pkgen")
        S.append("PK_DEFINE_KERNELS(pkgen::Gen)")
        return ("\n".join(S) + "\n").replace("//@PK_BIG_GLOBAL@", "#define PK_BIG_GLOBAL 1" if self.big_global else
                                                "// (workgroup-wide intervals stage their rows in LDS)")
