#!/usr/bin/env python3
"""GPU box: what Evaluator.checked does for one wide_mix_soak seed -- the unchecked default build's self-check, which build the
checked evaluator ends up with, and its callbacks / one-launch cycle against the oracle.  usage: self_check_probe.py seed"""
import importlib
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np  # noqa: E402

import models  # noqa: E402
from pockit_amd.evaluator import Evaluator  # noqa: E402
from wide_mix_soak import shape_of  # noqa: E402

seed = int(sys.argv[1])
kw, scheme = shape_of(seed)
system, _, guess = models.wide_mix(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
ref, _, _ = models.wide_mix(importlib.import_module(f"oracle.{scheme}"), **kw)
with warnings.catch_warnings(record=True) as seen:
    warnings.simplefilter("always")
    plain = Evaluator(system.plan)
    print(f"seed {seed} {scheme} {kw}\n   default build: cap {plain.src.group_cap} subs {plain.src.cycle_subs}; self-check -> {plain.self_check()}", flush=True)
    plain.close()
    try:
        ev = system.evaluator
    except RuntimeError as exc:
        print("   checked evaluator: RAISED", str(exc)[:300], flush=True)
        sys.exit(0)
print(f"   checked evaluator: flags {ev.hipcc_flags} cap {ev.src.group_cap} subs {ev.src.cycle_subs}; self-check -> {ev.self_check()}; "
      f"warnings: {[str(w.message)[:90] for w in seen if 'self-check' in str(w.message)]}", flush=True)
x, lam, sigma = models.bench_inputs(system, guess)
want = (ref.objective(x), ref.gradient(x), ref.constraints(x), ref.jacobian(x), ref.hessian(x, lam, sigma))


def err(a, b):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


got = (system.objective(x), system.gradient(x), system.constraints(x), system.jacobian(x), system.hessian(x, lam, sigma))
print("   callbacks vs oracle:      ", " ".join(f"{n} {err(a, b):.1e}" for n, a, b in zip(("f", "grad", "g", "J", "H"), got, want)), flush=True)
print("   one-launch cycle vs oracle:", " ".join(f"{n} {err(a, b):.1e}" for n, a, b in zip(("f", "grad", "g", "J", "H"), ev.cycle(x, lam, sigma), want)), flush=True)
system._invalidate()
