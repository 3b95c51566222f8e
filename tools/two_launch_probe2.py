#!/usr/bin/env python3
"""(Historical: since the containment at the end of round 5 the library refuses pk_xall for a model with a wide phase, error 27 -- this probe reproduces the defect only on a tree before that commit.)
GPU box: the two-launch cycle (pk_set_cycle_mode 0: pk_xall, then pk_hess with the reductions) of benchmarks.wide_mix
(30, 30, 30, 30) + 30 statics against the oracle AT POINTS THE CONTEXT HAS NOT SEEN (x (1 + 1e-3 U_k): a result served from an
earlier iterate would be off by ~1e-3), all five outputs.  Second version of tools/two_launch_probe.py, whose same-x calls
could not tell a fresh evaluation from a stale one."""
import importlib
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402

shape, statics = (30, 30, 30, 30), 30
n_two = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = dict(shapes=(shape,), statics=statics, free_time=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    system, _, guess = models.wide_mix(importlib.import_module("pockit_amd.radau"), **kw)
    ev = system.evaluator
ref, _, _ = models.wide_mix(importlib.import_module("oracle.radau"), **kw)
x0, lam, sigma = models.bench_inputs(system, guess)
print(f"tree {ROOT}  TRACE={os.environ.get('POCKIT_AMD_TRACE', '0')} DEBUG_FLAGS={os.environ.get('POCKIT_AMD_DEBUG_FLAGS', '0')} "
      f"cap {ev.src.group_cap} subs {ev.src.cycle_subs}", flush=True)


def check(tag, k):
    x = x0 * (1.0 + 1.0e-3 * np.random.default_rng(100 + k).uniform(-1.0, 1.0, x0.shape))
    want = (ref.objective(x), ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma))
    got = ev.cycle(x, lam, sigma)
    errs = []
    for a, b in zip(got, want):
        a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
        errs.append(float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))))
    print(f"   {tag:28s} point {k}: rel err f {errs[0]:.2e} grad {errs[1]:.2e} g {errs[2]:.2e} J {errs[3]:.2e} H {errs[4]:.2e}", flush=True)


check("one launch", 0)
ev.set_cycle_mode(False)
for k in range(1, 1 + n_two):
    check("two launches", k)
ev.set_cycle_mode(True)
check("one launch again", 9)
system._invalidate()
