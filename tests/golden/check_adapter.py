#!/usr/bin/env python3
"""Build-container only: import the REFERENCE (behind the shims of make_golden.py), build every small case with it,
convert each configured reference system with pockit_amd.adapter.plan_from_reference_system and compare the resulting
plan with the golden vectors: structures exactly, values through the NumPy plan interpreter.  Prints one JSON object.
Run in a process of its own (the shims must not leak into the test process)."""
import json
import os
import sys
import types
import typing

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
if not hasattr(typing, "Self"):
    typing.Self = typing.TypeVar("Self")
sys.path.insert(0, os.path.join(HERE, "refharness"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))

import numpy as np  # noqa: E402

import pockit.lobatto as ref_lobatto  # noqa: E402
import pockit.radau as ref_radau  # noqa: E402

import models  # noqa: E402
from plan_interp import Interp  # noqa: E402
from pockit_amd.adapter import plan_from_reference_system  # noqa: E402

NS = {"radau": ref_radau, "lobatto": ref_lobatto}
out = {}
for name, (builder, scheme, kw) in sorted(models.SMALL_CASES.items()):
    if name in models.SLOW_ON_CPU:      # (minutes in the NumPy plan interpreter)
        continue
    gold = np.load(os.path.join(HERE, "small", name + ".npz"))
    ref_system, _, _ = builder(NS[scheme], **kw)
    plan = plan_from_reference_system(ref_system)
    ok = (np.array_equal(plan.jac_row, gold["jr"]) and np.array_equal(plan.jac_col, gold["jc"])
          and np.array_equal(plan.hess_row, gold["hr"]) and np.array_equal(plan.hess_col, gold["hc"]))
    it = Interp(plan, gold["x"], gold["lam"], float(gold["sigma"]))
    err = 0.0
    for got, key in ((it.objective(), "f"), (it.gradient(), "grad"), (it.constraints(), "g"), (it.jacobian(), "J"),
                     (it.hessian(), "H")):
        want = np.asarray(gold[key], dtype=np.float64)
        got = np.asarray(got, dtype=np.float64).reshape(want.shape)
        if want.size:
            err = max(err, float(np.max(np.abs(got - want)) / max(1.0, float(np.max(np.abs(want))))))
    out[name] = {"structure": bool(ok), "err": err, "n": int(plan.n), "m": int(plan.m)}
# bang-bang flags: the reference keeps only compiled scaled functions; the adapter must recover which constraints they mark
import pockit_amd.lobatto as my_lobatto  # noqa: E402
import pockit_amd.radau as my_radau  # noqa: E402
from pockit_amd.adapter import system_from_reference  # noqa: E402

MINE = {"radau": my_radau, "lobatto": my_lobatto}
for scheme in ("radau", "lobatto"):
    for second in (False, True):
        ref_system, _, _ = models.bang_bang_model(NS[scheme], second=second)
        mine, _, _ = models.bang_bang_model(MINE[scheme], second=second)
        conv = system_from_reference(ref_system)
        got = [(k, i) for k, i, _, _ in conv._phase[0]._bang_bang]
        want = [(k, i) for k, i, _, _ in mine._phase[0]._bang_bang]
        out[f"bang_bang_{scheme}_{int(second)}"] = {"structure": got == want and len(want) == (2 if second else 1), "err": 0.0,
                                                     "n": int(conv.plan.n), "m": int(conv.plan.m)}
print(json.dumps(out))
