#!/usr/bin/env python3
"""Developer helper (GPU box): is the back-to-back cycle loop of bench.py paced by the GPU or by the host?  Enqueues
N launches of pk_cycle through ctypes in chunks of 200, records a HIP event and the host clock after every chunk and
prints both paces (us per launch) chunk by chunk: where the host column is the larger one the GPU waited for the host.
Usage: launch_rate_probe.py [workload] [intervals] [chunks] [many|graph]   (many: every chunk is one pk_eval_cycle_dev_repeat call; graph: replayed from a hipGraph)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "planar_quadrotor"
intervals = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 100
torch.cuda.set_device(0)
w = bench.GpuWorkload(name, intervals, 0, 1, None)
for _ in range(2000):
    w.step()
w.sync()
use_many = len(sys.argv) > 4 and sys.argv[4] in ("many", "graph")   # chunks enqueued by ONE library call (pk_eval_cycle_dev_repeat)
if len(sys.argv) > 4 and sys.argv[4] == "graph":                       # ... replayed from one hipGraph of 200 kernel nodes
    w.ev.set_cycle_graph(True)
    w.step.many(200)
    w.sync()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
host = [0.0] * (chunks + 1)
ev[0].record(w.stream)
host[0] = time.perf_counter()
for c in range(chunks):
    if use_many:
        w.step.many(200)
    else:
        for _ in range(200):
            w.step()
    ev[c + 1].record(w.stream)
    host[c + 1] = time.perf_counter()
w.sync()
gpu = [ev[c].elapsed_time(ev[c + 1]) * 1e3 / 200 for c in range(chunks)]
hst = [(host[c + 1] - host[c]) * 1e6 / 200 for c in range(chunks)]
lag = [(sum(gpu[: c + 1]) - sum(hst[: c + 1])) * 200 for c in range(chunks)]     # us the GPU is behind the host's enqueueing
print("chunk  gpu us/launch  host us/launch  GPU behind host (us)")
for c in range(chunks):
    print(f"{c:4d}   {gpu[c]:8.3f}      {hst[c]:8.3f}      {lag[c]:10.0f}")
print(f"median gpu {sorted(gpu)[chunks // 2]:.3f}  median host {sorted(hst)[chunks // 2]:.3f}  total gpu {sum(gpu) / chunks:.3f}  total host {sum(hst) / chunks:.3f}")
