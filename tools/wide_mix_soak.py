#!/usr/bin/env python3
"""Soak of the DEFAULT path over randomly shaped models that are wide in several directions (benchmarks.wide_mix): for each seed
a shape (states, controls, path constraints, integrals in 4 ... 40, statics 2 ... 34, one or two phases, LGR or LGL, fixed or free
final time) -- structures, the five callbacks, the one-launch cycle, every stand-alone kernel and the compact layouts against the
oracle to 1e-11, and the refusal of the two-launch form (tests/test_gpu_wide_models._check_everything: since the containment
of the round-5 defect, DESIGN.md section 11, a model with a wide phase is pass-parallel only and pk_xall is refused for it).
usage: wide_mix_soak.py [--compile-only] seed [seed ...]     (without a GPU, --compile-only fills the code-object cache)"""
import importlib
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402


def shape_of(seed):
    r = np.random.default_rng(1000 + seed)
    n_ph = 1 if r.uniform() < 0.6 else 2
    shapes = tuple((int(r.integers(17, 41)) if k == 0 else int(r.integers(4, 25)), int(r.integers(1, 31)),
                    int(r.integers(0, 31)), int(r.integers(1, 31))) for k in range(n_ph))
    return dict(shapes=shapes, statics=int(r.integers(n_ph + 1, 35)), mesh=int(r.integers(6, 61)), num_point=int(r.integers(3, 8)),
                free_time=bool(r.uniform() < 0.5)), ("radau" if r.uniform() < 0.6 else "lobatto")


def main():
    compile_only = "--compile-only" in sys.argv
    seeds = [int(a) for a in sys.argv[1:] if not a.startswith("--")]
    bad = 0
    for seed in seeds:
        kw, scheme = shape_of(seed)
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            system, _, guess = models.wide_mix(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
            if compile_only:
                from pockit_amd.evaluator import compile_plan

                try:
                    src, _ = compile_plan(system.plan)
                except (RuntimeError, ValueError) as exc:      # (a hipcc failure is a result of the soak, not its end)
                    print(f"seed {seed} {scheme} {kw}: COMPILE FAILED {str(exc)[:300]!r} ({time.time() - t0:.0f} s)", flush=True)
                    continue
                print(f"seed {seed} {scheme} {kw}: cap {src.group_cap} subs {src.cycle_subs} wide {src.wide} spills {src.spilling_kernels} "
                      f"({time.time() - t0:.0f} s)", flush=True)
                continue
            try:
                src = system.evaluator.src
            except (RuntimeError, ValueError) as exc:
                print(f"seed {seed} {scheme} {kw}: COMPILE / LOAD FAILED {str(exc)[:300]!r}", flush=True)
                bad += 1
                continue
        ref, _, _ = models.wide_mix(importlib.import_module(f"oracle.{scheme}"), **kw)
        import test_gpu_wide_models as T

        try:
            T._check_everything(system, ref, guess, f"seed {seed}")
            verdict = "OK"
        except AssertionError as exc:
            verdict, bad = f"FAILED: {str(exc)[:160]}", bad + 1
        ev = system._evaluator
        how = "stand-alone kernels (both builds failed the self-check)" if getattr(ev, "separate_x", False) else \
            ("REBUILT with SGPR spills in scratch (default build failed the self-check)" if getattr(ev, "hipcc_flags", ()) else "default build")
        print(f"seed {seed} {scheme} {kw}: cap {src.group_cap} subs {src.cycle_subs} spills {bool(src.spilling_kernels)} [{how}] -> {verdict}", flush=True)
        system._invalidate()
    log = os.environ.get("POCKIT_AMD_COMPILE_LOG")      # (objects compiled on the GPU box -- rebuilds of the checked evaluator -- go home)
    if log and os.path.exists(log) and os.environ.get("GRAFT_REPO_ROOT"):
        import shutil

        from pockit_amd import hipbuild

        dst = os.path.join(ROOT, "gpurun_out", "cache_new")
        os.makedirs(dst, exist_ok=True)
        for key in {ln.split()[0] for ln in open(log) if ln.strip()}:
            for ext in (".hsacoz", ".gen", ".res.json"):
                if os.path.exists(os.path.join(hipbuild.CACHE_DIR, key + ext)):
                    shutil.copy(os.path.join(hipbuild.CACHE_DIR, key + ext), dst)
    if not compile_only:
        print(f"{len(seeds)} models, {bad} failed")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
