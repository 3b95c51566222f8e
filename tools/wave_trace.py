#!/usr/bin/env python3
"""Developer helper: timeline of one pk_cycle launch from s_memrealtime marks (the model is generated with PK_TRACE
because this script sets POCKIT_AMD_TRACE=1).  The clock is the constant-rate device clock (100 MHz, common to all
XCDs), so marks of different waves are comparable: everything is printed in microseconds since the earliest mark of
the launch.  Usage: wave_trace.py [workload] [intervals]   (default planar_quadrotor 2000)"""
import ctypes as C
import os
import sys

os.environ["POCKIT_AMD_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pockit_amd import benchmarks as models  # noqa: E402
import pockit_amd.radau as radau  # noqa: E402

TICK_US = 0.01
MARKS = {0: "wave entry (tile record in SGPRs)", 1: "x + tables loaded", 2: "evaluation starts", 3: "evaluation done",
         10: "wave sums done (DPP trees)", 11: "barrier + partial sums handed off", 12: "gradient stores issued",
         13: "LDS staging stores issued", 4: "path / per-node stores issued",
         5: "phase B starts", 6: "defects issued", 7: "translation issued", 8: "streaming issued",
         9: "stores acknowledged"}
ROLES = ["values wave", "Jacobian wave", "Hessian wave"]
name = sys.argv[1] if len(sys.argv) > 1 else "planar_quadrotor"
intervals = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if name.startswith("example:"):          # an example program's model (tests/golden/examples/<name>.model.json) re-meshed to
    import json                          # `intervals` x argv[3] points: example:drone_stabilization 2000 4

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_io

    with open(os.path.join(ROOT, "tests", "golden", "examples", name.split(":", 1)[1] + ".model.json")) as fh:
        desc = json.load(fh)
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    for pd in desc["phases"]:
        pd["mesh"] = [float(v) for v in np.linspace(0.0, 1.0, intervals + 1)]
        pd["num_point"] = [K] * intervals
    system = model_io.load_system(desc)
    rng = np.random.default_rng(7)
    x, lam, sigma = rng.uniform(0.6, 1.4, size=system.plan.n), rng.standard_normal(system.plan.m), 0.7
else:
    system, _, guess = bench.build_workload(name, intervals, radau)
    x, lam, sigma = models.bench_inputs(system, guess)
ev = system.evaluator
lib, h = ev.ctx.lib, ev.ctx.handle
ev.ctx.check(lib.pk_trace_read(h, None, 0))                      # arm
n = len(ev.tables.tiles)
buf = np.zeros((3 * n + 3) * 16, dtype=np.uint64)
b2b = os.environ.get("POCKIT_AMD_TRACE_B2B") == "1"
if b2b:
    import torch

    dev = torch.device("cuda", 0)
    dx, dlam = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
    plan = system.plan
    outs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (1, plan.n, plan.m, plan.nnz_J, plan.nnz_H)]
    torch.cuda.synchronize()
for rep in range(4):
    if b2b:
        for _ in range(6):
            ev.cycle_dev(dx.data_ptr(), dlam.data_ptr(), sigma, *[o.data_ptr() for o in outs])
        ev.sync()
    for _ in range(0 if b2b else 3):                             # the last of three cycles is read
        ev.cycle(x, lam, sigma)                                  # (ev.cycle synchronizes: every launch starts on an idle GPU;
                                                                 #  POCKIT_AMD_TRACE_B2B=1 traces the last of 3 queued launches)
    ev.ctx.check(lib.pk_trace_read(h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), len(buf)))
    m = buf.reshape(3 * n + 3, 16).astype(np.int64)
    clk = m[:, 14:].copy()                                        # (marks 14 / 15 of the three extra records hold s_memtime,
    entry = m[:3 * n, 14].copy()                                  #  mark 14 of a tile wave its device clock at kernel entry)
    m[:, 14:] = 0
    t0 = m[m > 0].min()
    us = np.where(m > 0, (m - t0) * TICK_US, np.nan)
    print(f"rep {rep}: launch spans {np.nanmax(us):.2f} us from its first to its last mark")
    if rep < 3:
        continue
    for role in range(3):
        r = us[role:3 * n:3]
        ent = entry[role:3 * n:3]
        ok = ~np.isnan(r[:, 0])
        r = r[ok]
        if not len(r):
            continue
        print(f"  {ROLES[role]} ({len(r)} waves)")
        ent = ent[ok]
        if (ent > 0).any():                                      # kernel entry -> tile record and kernel arguments in SGPRs
            d = (m[role:3 * n:3, 0][ok][ent > 0] - ent[ent > 0]) * TICK_US
            e0 = (ent[ent > 0] - t0) * TICK_US
            print(f"    {'kernel entry':36s} median {np.median(e0):6.2f}  p10 {np.percentile(e0, 10):6.2f}  p90 {np.percentile(e0, 90):6.2f}  "
                  f"max {e0.max():6.2f}   (entry -> tile record in SGPRs: median {np.median(d):.2f} us, p10 {np.percentile(d, 10):.2f}, "
                  f"p90 {np.percentile(d, 90):.2f})")
        for k, label in MARKS.items():
            col = r[:, k][~np.isnan(r[:, k])]
            if len(col):
                print(f"    {label:36s} median {np.median(col):6.2f}  p10 {np.percentile(col, 10):6.2f}  "
                      f"p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f}")
    # tiles are dealt to the XCDs in contiguous ranges (xcd_tile_block): entry / end per eighth of the tile list
    jw = us[1:3 * n:3] if not np.all(np.isnan(us[1:3 * n:3, 0])) else us[0:3 * n:3]
    parts = np.array_split(np.arange(len(jw)), 8)
    print("  Jacobian waves by eighth of the tile list (~XCD): first entry / median entry / last store acknowledged")
    print("   " + "  ".join(f"{np.nanmin(jw[ix, 0]):.2f}/{np.nanmedian(jw[ix, 0]):.2f}/{np.nanmax(jw[ix, 9]):.2f}" for ix in parts))
    for i, label in enumerate(("boundary workgroup (g, J)", "boundary workgroup (H)", "finalize workgroup")):
        row = us[3 * n + i]
        print(f"  {label}: " + "  ".join(f"[{k}] {row[k]:.2f}" for k in range(14) if not np.isnan(row[k])))
    raw = m[3 * n + 2]                                            # finalize workgroup: s_memtime at marks 0 and 9
    ck = clk[3 * n + 2]
    if ck[1] > ck[0] and raw[9] > raw[0]:
        print(f"  shader clock over the finalize workgroup's life: {(ck[1] - ck[0]) / ((raw[9] - raw[0]) * TICK_US):.0f} "
              f"s_memtime ticks per us")
