#!/usr/bin/env python3
"""Build-container only (developer tool): run every example program of the reference with ``pockit`` resolving to
``pockit_amd`` up to its ``ipopt.solve`` call and print, per model, what sizes the code generator and the kernels see:
states / controls, segment counts per role, boundary-list lengths, LDS rows per wave, and how long the plan took.

Usage: examples_survey.py [name-substring ...]"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.dont_write_bytecode = True

import check_examples as ce  # noqa: E402


def survey(name):
    ce.use_product()
    t0 = time.time()
    cap = ce.run_program(os.path.join(ce.EXAMPLES, name))
    plan = cap.system.plan
    plan.jac, plan.hess  # noqa: B018
    t_plan = time.time() - t0
    rows = []
    for k, pp in enumerate(plan.phase_plans):
        def cnt(cb, kind):
            return sum(1 for s in cb.segs[k] if s.kind == kind)

        rows.append({"nx": pp.nx, "nu": pp.nu, "nc": pp.phase.n_c, "K": sorted(set(int(v) for v in pp.layout.K)),
                     "N": int(len(pp.layout.K)), "scheme": pp.layout.scheme,
                     "J_NI": cnt(plan.jac, "I"), "J_NN": cnt(plan.jac, "N"), "H_NI": cnt(plan.hess, "I"), "H_NN": cnt(plan.hess, "N"),
                     "lists": {w: [len(plan.jac.lists.get((w, k), [])), len(plan.hess.lists.get((w, k), []))] for w in "fb"}})
    return {"n": int(plan.n), "m": int(plan.m), "nnz_J": int(plan.nnz_J), "nnz_H": int(plan.nnz_H), "n_s": int(plan.n_s),
            "outer": bool(plan.outer), "phases": rows, "sys_lists": [len(plan.jac.lists.get(("s",), [])), len(plan.hess.lists.get(("s",), []))],
            "plan_s": round(t_plan, 1)}


def main():
    only = sys.argv[1:]
    out = {}
    for name in sorted(p for p in os.listdir(ce.EXAMPLES) if p.endswith(".py") and not p.startswith("_")):
        if only and not any(o in name for o in only):
            continue
        try:
            out[name[:-3]] = survey(name)
        except Exception as exc:  # noqa: BLE001
            out[name[:-3]] = {"error": repr(exc)[:300]}
        print(name, json.dumps(out[name[:-3]]), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
