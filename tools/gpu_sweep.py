#!/usr/bin/env python3
"""Developer helper: run bench.py under several environment settings and print one line each.
usage: python tools/gpu_sweep.py VAR v1 v2 ... [-- extra bench args]"""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
var, values = args[0], args[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in values:
    env = dict(os.environ)
    env[var] = v
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "200"]
                         + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        d = {**json.load(open(os.path.join(root, "bench_detail.json"))), **d}      # (the long tables live in the detail file)
        ku = {k: round(x, 2) for k, x in d["kernel_us"].items() if x}
        print(f"{var}={v} tiles={d['config']['tiles']} cycles/s={d['value']:.0f} ms={d['ms_per_step']:.4f} "
              f"roof={d['roofline']['kernel']}:{d['roofline']['achieved']:.0f}GB/s kernels_us={ku}", flush=True)
    except Exception as exc:
        print(f"{var}={v} FAILED {exc!r}\n{out.stderr[-2000:]}", flush=True)
