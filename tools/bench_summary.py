#!/usr/bin/env python3
"""Developer helper: one line per bench DETAIL file given on the command line (bench_detail.json: value, kernel times, side
workloads)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    side = " ".join(f"{k}:{round(v['cycles_per_s'])}" for k, v in d.get("other_workloads", {}).items() if "cycles_per_s" in v)
    dr = d["device_resident"]
    print(path.split("/")[-1], round(dr["value"]), f"{dr['ms_per_step'] * 1e3:.2f}us",
          {k: round(v, 2) for k, v in d["kernel_us"].items() if v}, d["outputs_finite"], "|", side)
