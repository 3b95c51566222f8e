#!/usr/bin/env python3
"""Build-container only: golden vectors of EVERY example program of the reference (/root/reference/examples/*.py, run
where they lie, never copied), at the program's own mesh, for the GPU parity test tests/test_gpu_examples.py.

Per program two files under tests/golden/examples/:
* ``<name>.npz``  -- inputs and the REFERENCE's outputs: x (the program's initial guess, perturbed and seeded as
  pockit_amd.benchmarks.bench_inputs does), lambda, sigma -> f, grad f, g, J, H of the reference's own callbacks
  (systembase.py:602-835, NumPy execution of its generated functions behind refharness/numba), its triplet structures and
  bounds;
* ``<name>.model.json`` -- what the GPU box needs to rebuild the evaluator WITHOUT the program text: the model the program
  configured on this package's modeling API (``pockit`` resolving to ``pockit_amd``), as written by tests/model_io.py
  (SymPy expressions as srepr text + settings: data, not source).

Usage: make_examples.py [name-substring ...]      (runs in a process of its own: the shims must not leak)"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import check_examples as ce  # noqa: E402  (the harness that runs a program up to its ipopt.solve call)
import numpy as np  # noqa: E402

OUT = os.path.join(HERE, "examples")


def make(name):
    import model_io
    from pockit_amd import benchmarks as models

    path = os.path.join(ce.EXAMPLES, name)
    ce.use_product()
    mine = ce.run_program(path)
    ce.use_reference()
    ref = ce.run_program(path)
    system, rsys = mine.system, ref.system
    x, lam, sigma = models.bench_inputs(system, ce.as_list(mine.guess))
    rsys.update()
    assert int(rsys.L) == x.size and len(rsys.c_lb) == lam.size
    jr, jc = rsys.jacobianstructure()
    hr, hc = rsys.hessianstructure()
    out = dict(x=x, lam=lam, sigma=np.float64(sigma),
               f=np.float64(rsys.objective(x.copy())), grad=rsys.gradient(x.copy()), g=rsys.constraints(x.copy()),
               J=rsys.jacobian(x.copy()), H=rsys.hessian(x.copy(), lam, sigma),
               jr=np.asarray(jr, dtype=np.int32), jc=np.asarray(jc, dtype=np.int32),
               hr=np.asarray(hr, dtype=np.int32), hc=np.asarray(hc, dtype=np.int32),
               v_lb=rsys.v_lb, v_ub=rsys.v_ub, c_lb=rsys.c_lb, c_ub=rsys.c_ub)
    for k in ("grad", "g", "J", "H"):
        if not np.all(np.isfinite(out[k])):
            raise RuntimeError(f"{name}: the reference's {k} is not finite at the perturbed guess")
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name[:-3] + ".npz"), **out)
    desc = model_io.dump_system(system)
    # the description must rebuild the same NLP on this package (checked here, where the program-built system exists)
    again = model_io.load_system(desc)
    assert again.plan.n == system.plan.n and again.plan.m == system.plan.m
    for a, b in zip(again.jacobianstructure() + again.hessianstructure(), (jr, jc, hr, hc)):
        assert np.array_equal(np.asarray(a), np.asarray(b)), name + ": rebuilt model has another triplet structure"
    with open(os.path.join(OUT, name[:-3] + ".model.json"), "w") as fh:
        json.dump(desc, fh, indent=0, sort_keys=True)
    return {"n": int(x.size), "m": int(lam.size), "nnz_J": int(len(jr)), "nnz_H": int(len(hr)),
            "sha": hashlib.sha256(out["J"].tobytes() + out["H"].tobytes()).hexdigest()[:16]}


def main():
    only = sys.argv[1:]
    index = {}
    for name in sorted(p for p in os.listdir(ce.EXAMPLES) if p.endswith(".py") and not p.startswith("_")):
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        index[name[:-3]] = make(name)
        print(name, index[name[:-3]], f"{time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    print(json.dumps(index))


if __name__ == "__main__":
    main()
