#!/usr/bin/env python3
"""One wide_mix_soak seed, kernel by kernel, with a progress line BEFORE every device call (flushed): if the GPU faults, the last
line names the call.  Default path and stand-alone kernels only (no two-launch form).  usage: wide_mix_steps.py seed"""
import importlib
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np  # noqa: E402

import models  # noqa: E402
from wide_mix_soak import shape_of  # noqa: E402  (argv parsing there is harmless: no seeds -> no work)

seed = int(sys.argv[1])
kw, scheme = shape_of(seed)


def say(msg):
    print(msg, flush=True)
    sys.stderr.flush()


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    system, _, guess = models.wide_mix(importlib.import_module(f"pockit_amd.{scheme}"), **kw)
    say(f"seed {seed} {scheme} {kw} flags {os.environ.get('POCKIT_AMD_HIPCC_FLAGS', '')!r}: creating the evaluator")
    ev = system.evaluator
src = ev.src
say(f"   cap {src.group_cap} subs {src.cycle_subs} wide {src.wide} spills {src.spilling_kernels}")
ref, _, _ = models.wide_mix(importlib.import_module(f"oracle.{scheme}"), **kw)
x, lam, sigma = models.bench_inputs(system, guess)
want = dict(f=ref.objective(x), grad=ref.gradient(x), g=ref.constraints(x), J=ref.jacobian(x), H=ref.hessian(x, lam, sigma))


def err(a, b):
    a, b = np.atleast_1d(np.asarray(a, dtype=np.float64)), np.atleast_1d(np.asarray(b, dtype=np.float64))
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))) if a.size else 0.0


steps = [("objective_direct (pk_int + pk_fin)", lambda: err(ev.objective_direct(x), want["f"])),
         ("gradient_direct (pk_grad)", lambda: err(ev.gradient_direct(x), want["grad"])),
         ("constraints_direct (pk_g)", lambda: err(ev.constraints_direct(x), want["g"])),
         ("jacobian_direct (pk_jac)", lambda: err(ev.jacobian_direct(x), want["J"])),
         ("hessian_direct (pk_hess)", lambda: err(ev.hessian_direct(x, lam, sigma), want["H"])),
         ("callbacks f, grad, g, J (x-part of pk_cycle)", lambda: max(err(system.objective(x), want["f"]), err(system.gradient(x), want["grad"]),
                                                                        err(system.constraints(x), want["g"]), err(system.jacobian(x), want["J"]))),
         ("callback H", lambda: err(system.hessian(x, lam, sigma), want["H"])),
         ("one-launch cycle (pk_cycle)", lambda: max(err(a, want[k]) for a, k in zip(ev.cycle(x, lam, sigma), ("f", "grad", "g", "J", "H")))),
         ("hessian_compact (pk_hessc)", lambda: float(np.isfinite(ev.hessian_compact(x, lam, sigma)).all()) - 1.0),
         ("jacobian_compact (pk_jacc)", lambda: (system.plan.jacc, float(np.isfinite(ev.jacobian_compact(x)).all()) - 1.0)[1])]
for name, fn in steps:
    say(f"   -> {name}")
    say(f"      rel err {fn():.2e}")
say("all steps done")
system._invalidate()
