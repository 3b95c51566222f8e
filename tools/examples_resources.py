#!/usr/bin/env python3
"""Generate and compile (hipcc, gfx950; no GPU needed) the code object of every example model of tests/golden/examples and
print what the compiler reports per kernel: registers, spills, scratch, occupancy, and the group size the evaluator settled
on (evaluator.compile_plan).  Fills pockit_amd/_cache on the way.

Usage: examples_resources.py [name-substring ...]  ->  one JSON object on stdout, a table on stderr"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import model_io  # noqa: E402
from pockit_amd import hipbuild  # noqa: E402
from pockit_amd.evaluator import compile_plan  # noqa: E402

EX = os.path.join(ROOT, "tests", "golden", "examples")


def one(name):
    with open(os.path.join(EX, name + ".model.json")) as fh:
        system = model_io.load_system(json.load(fh))
    t0 = time.time()
    plan = system.plan
    t1 = time.time()
    c0 = hipbuild.COMPILE_SECONDS["total"]
    src, _ = compile_plan(plan)
    usage = hipbuild.resource_usage(src.source, fastmath=system._fastmath) or {}
    worst = lambda key: max([0] + [v.get(key, 0) for v in usage.values()])  # noqa: E731
    ng = {f"{cb}{k}": len(g) for (cb, k), g in src.groups.items() if len(g) > 1}
    return {"group_cap": src.group_cap, "groups": ng, "lds_rows": [src.lds_g // 64, src.lds_j // 64, src.lds_h // 64, src.lds_x // 64, src.lds_jc // 64],
            "max_vgpr": worst("vgpr"), "max_agpr": worst("agpr"), "max_sgpr_spill": worst("sgpr_spill"), "max_vgpr_spill": worst("vgpr_spill"),
            "max_scratch": worst("scratch"), "min_occupancy": min([99] + [v.get("occupancy", 99) for v in usage.values()]),
            "spilling": src.spilling_kernels, "plan_s": round(t1 - t0, 1), "codegen_s": round(time.time() - t1 - (hipbuild.COMPILE_SECONDS["total"] - c0), 1),
            "hipcc_s": round(hipbuild.COMPILE_SECONDS["total"] - c0, 1),
            "kernels": {k: [v.get("vgpr"), v.get("agpr"), v.get("sgpr_spill"), v.get("vgpr_spill"), v.get("scratch"), v.get("occupancy")]
                        for k, v in usage.items()}}


def main():
    only = sys.argv[1:]
    out = {}
    for name in sorted(f[:-11] for f in os.listdir(EX) if f.endswith(".model.json")):
        if only and not any(o in name for o in only):
            continue
        try:
            out[name] = one(name)
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": repr(exc)[:500]}
        print(name, json.dumps({k: v for k, v in out[name].items() if k != "kernels"}), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
