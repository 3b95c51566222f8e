#!/usr/bin/env python3
"""Developer helper: timeline of pk_xall's waves from s_memtime marks (POCKIT_AMD_TRACE=1 must be set so that the
model is generated with PK_TRACE).  Prints, per checkpoint, the median / p10 / p90 offset from the earliest mark
of the launch, in shader-clock ticks and microseconds (tick rate estimated from the kernel's event duration)."""
import ctypes as C
import os
import sys

os.environ["POCKIT_AMD_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

NAMES = ["entry", "x+tables loaded", "eval start", "eval end", "before barrier", "after barrier", "defects issued",
         "translation issued", "stream issued", "stores acked"]
system, _, guess = models.planar_quadrotor(radau, mesh=2000, num_point=6)
x, lam, sigma = models.bench_inputs(system, guess)
ev = system.evaluator
lib, h = ev.ctx.lib, ev.ctx.handle
ev.ctx.check(lib.pk_trace_read(h, None, 0))                      # arm
n = len(ev.tables.tiles)
buf = np.zeros(n * 16, dtype=np.uint64)
for rep in range(4):
    ev.cycle(x, lam, sigma)
    ev.ctx.check(lib.pk_trace_read(h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), len(buf)))
    m = buf.reshape(n, 16).astype(np.int64)
    live = m[:, 0] > 0
    m = m[live]
    t0 = m[:, 0].min()
    print(f"rep {rep}: {live.sum()} waves; span of the launch {m[:, :10].max() - t0} ticks")
    # the counters of different XCDs are not aligned: only differences inside one wave are meaningful
    order = [10, 11] + list(range(10)) + [12]
    names = {10: "kernel entry", 11: "tile record loaded", 12: "partials published"}
    names.update({k: nm for k, nm in enumerate(NAMES)})
    prev = order[0]
    for k in order:
        d = m[:, k] - m[:, prev]
        tot = m[:, k] - m[:, 10]
        print(f"  {names[k]:20s} since previous mark: median {np.median(d):7.0f} p10 {np.percentile(d, 10):7.0f} "
              f"p90 {np.percentile(d, 90):7.0f}   since kernel entry: median {np.median(tot):7.0f}")
        prev = k
