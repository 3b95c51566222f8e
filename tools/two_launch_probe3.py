#!/usr/bin/env python3
"""(Historical: since the containment at the end of round 5 the library refuses pk_xall for a model with a wide phase, error 27 -- this probe reproduces the defect only on a tree before that commit.)
GPU box: WHICH entries of grad and g the two-launch cycle of wide_mix (30, 30, 30, 30) + 30 statics gets wrong (index sets
mapped to the layout: state / control / static of the variable, defect row of which state / path row / system row)."""
import importlib
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402

kw = dict(shapes=((30, 30, 30, 30),), statics=30, free_time=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    system, _, guess = models.wide_mix(importlib.import_module("pockit_amd.radau"), **kw)
    ev = system.evaluator
ref, _, _ = models.wide_mix(importlib.import_module("oracle.radau"), **kw)
x0, lam, sigma = models.bench_inputs(system, guess)
plan = system.plan
pp = plan.phase_plans[0]
lay = pp.layout
print("n", plan.n, "m", plan.m, "state_len", lay.l_v[1] - lay.l_v[0], "L_m", lay.l_v[31] - lay.l_v[30] if len(lay.l_v) > 31 else None, flush=True)
ev.set_cycle_mode(False)
x = x0 * (1.0 + 1.0e-3 * np.random.default_rng(101).uniform(-1.0, 1.0, x0.shape))
f, grad, g, J, H = ev.cycle(x, lam, sigma)
wf, wgrad, wg = ref.objective(x), ref.gradient(x), ref.constraints(x)
bad = np.flatnonzero(~(np.abs(grad - wgrad) <= 1e-9 * (1 + np.abs(wgrad))))
sl = lay.l_v[1] - lay.l_v[0]
print("f", float(f), "want", float(wf))
print("grad wrong:", len(bad), "variable rows (index // state_len):", sorted(set((bad // sl).tolist()))[:20], "node range", (bad % sl).min() if len(bad) else None, (bad % sl).max() if len(bad) else None)
badg = np.flatnonzero(~(np.abs(g - wg) <= 1e-9 * (1 + np.abs(wg))))
print("g wrong:", len(badg), "of", len(g), "first", badg[:8], "last", badg[-8:])
L = len(g)
# g layout: defect rows (30 states x rows), path rows (30 x L_m), system rows at the end
nd = 30 * 160
print("   in defect rows:", int(np.sum(badg < nd)), "states", sorted(set((badg[badg < nd] // 160).tolist())),
      "| in path rows:", int(np.sum((badg >= nd) & (badg < nd + 30 * 160))), "constraints", sorted(set(((badg[(badg >= nd) & (badg < nd + 4800)] - nd) // 160).tolist())),
      "| beyond:", int(np.sum(badg >= nd + 4800)))
good = np.setdiff1d(np.arange(len(g)), badg)
print("   sample wrong g values:", g[badg[:4]], "want", wg[badg[:4]])
print("   sample wrong grad values:", grad[bad[:4]], "want", wgrad[bad[:4]])
system._invalidate()
