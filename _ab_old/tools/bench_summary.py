#!/usr/bin/env python3
"""Developer helper: one line per bench JSON given on the command line (value, kernel times, side workloads)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    side = " ".join(f"{k}:{round(v['cycles_per_s'])}" for k, v in d.get("other_workloads", {}).items() if "cycles_per_s" in v)
    print(path.split("/")[-1], round(d["value"]), f"{d['ms_per_step'] * 1e3:.2f}us",
          {k: round(v, 2) for k, v in d["kernel_us"].items() if v}, d["outputs_finite"], "|", side)
