#!/usr/bin/env python3
"""Build-container only: run every example PROGRAM of the reference (/root/reference/examples/*.py, read where it lies,
never copied) twice -- once against the reference itself (behind the shims of refharness/), once with ``pockit`` resolving
to ``pockit_amd`` -- up to its call of ``ipopt.solve``, which is intercepted, and compare what the two hand to the solver:

* the initial guess (the programs build it with linear_guess / constant_guess / Variable.V_x ...),
* the solver options,
* variable and constraint bounds, problem sizes,
* the triplet structures of J and H (exactly),
* the plan the adapter (pockit_amd/adapter.py) derives from the configured reference system, bang-bang flags included,
* f, grad f, g, J, H at a perturbed guess -- the reference's own callbacks (NumPy execution of its generated functions)
  against the NumPy execution of the product's plan (tests/plan_interp.py), to 1e-11.

That is the drop-in claim on the programs users actually write: same modeling API, same NLP.  Prints one JSON object.
Run in a process of its own (the shims and the module aliases must not leak into the test process).
Usage: check_examples.py [name-substring ...]     (all 33 programs by default; POCKIT_AMD_EXAMPLES_COMPILE=1: also generate
and compile every model's gfx950 code object with hipcc, no GPU needed)"""
import importlib
import json
import os
import runpy
import sys
import time
import types
import typing
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXAMPLES = "/root/reference/examples"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
warnings.simplefilter("ignore")
if not hasattr(typing, "Self"):
    typing.Self = typing.TypeVar("Self")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


class Captured(Exception):
    def __init__(self, system, guess, options):
        self.system, self.guess, self.options = system, guess, options


def _solve(system, guess, optimizer_options=None, **kw):
    raise Captured(system, guess, dict(optimizer_options=optimizer_options, **kw))


def _forget_pockit():
    for k in [k for k in sys.modules if k == "pockit" or k.startswith("pockit.") or k == "_plotting"]:
        del sys.modules[k]


def use_product():
    """``pockit`` -> pockit_amd (radau, lobatto, optimizer.ipopt with the intercepting solve)"""
    _forget_pockit()
    for p in ("/root/reference", os.path.join(HERE, "refharness")):
        while p in sys.path:
            sys.path.remove(p)
    root = types.ModuleType("pockit")
    root.__path__ = []
    sys.modules["pockit"] = root
    for sub in ("radau", "lobatto"):
        m = importlib.import_module(f"pockit_amd.{sub}")
        sys.modules[f"pockit.{sub}"] = m
        setattr(root, sub, m)
    opt = types.ModuleType("pockit.optimizer")
    opt.__path__ = []
    ip = types.ModuleType("pockit.optimizer.ipopt")
    ip.solve = _solve
    opt.ipopt = ip
    root.optimizer = opt
    sys.modules["pockit.optimizer"], sys.modules["pockit.optimizer.ipopt"] = opt, ip


def use_reference():
    """``pockit`` -> the reference (its numba / cyipopt imports satisfied by the harness stubs)"""
    _forget_pockit()
    sys.path.insert(0, os.path.join(HERE, "refharness"))
    sys.path.insert(0, "/root/reference")
    sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))
    import pockit.optimizer.ipopt as ip

    ip.solve = _solve


def run_program(path):
    argv, sys.argv = sys.argv, [path]
    sys.path.insert(0, os.path.dirname(path))
    try:
        runpy.run_path(path, run_name="__main__")
    except Captured as c:
        return c
    finally:
        sys.argv = argv
        sys.path.remove(os.path.dirname(path))
    raise RuntimeError("the program did not call ipopt.solve")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def as_list(g):
    return list(g) if isinstance(g, (list, tuple)) else [g]


def check(name):
    from plan_interp import Interp
    from pockit_amd import benchmarks as models

    path = os.path.join(EXAMPLES, name)
    use_product()
    mine = run_program(path)
    use_reference()
    ref = run_program(path)
    out = {}
    gm, gr = as_list(mine.guess), as_list(ref.guess)
    out["guess"] = max([rel(getattr(a, "data", a), getattr(b, "data", b)) for a, b in zip(gm, gr)] + [0.0 if len(gm) == len(gr) else float("inf")])
    out["options"] = mine.options == ref.options
    system, rsys = mine.system, ref.system
    x, lam, sigma = models.bench_inputs(system, gm)
    rsys.update()
    out["sizes"] = [int(system.plan.n), int(system.plan.m), int(system.plan.nnz_J), int(system.plan.nnz_H)]
    out["bounds"] = bool(np.array_equal(system.v_lb, rsys.v_lb) and np.array_equal(system.v_ub, rsys.v_ub)
                         and np.array_equal(system.c_lb, rsys.c_lb) and np.array_equal(system.c_ub, rsys.c_ub)
                         and int(rsys.L) == system.plan.n)
    jr, jc = rsys.jacobianstructure()
    hr, hc = rsys.hessianstructure()
    pjr, pjc = system.jacobianstructure()
    phr, phc = system.hessianstructure()
    out["structure"] = bool(np.array_equal(jr, pjr) and np.array_equal(jc, pjc) and np.array_equal(hr, phr) and np.array_equal(hc, phc))
    # the maintainer-side route: the configured REFERENCE system converted by the adapter must give
    # the same plan as the program run against this package, bang-bang flags included
    from pockit_amd.adapter import system_from_reference

    conv = system_from_reference(rsys)
    cjr, cjc = conv.jacobianstructure()
    chr_, chc = conv.hessianstructure()
    out["adapter"] = bool(np.array_equal(cjr, jr) and np.array_equal(cjc, jc) and np.array_equal(chr_, hr) and np.array_equal(chc, hc)
                          and np.array_equal(conv.v_lb, rsys.v_lb) and np.array_equal(conv.c_ub, rsys.c_ub)
                          and [[(k, i) for k, i, _, _ in p._bang_bang] for p in conv._phase]
                          == [[(k, i) for k, i, _, _ in p._bang_bang] for p in system._phase])
    if os.environ.get("POCKIT_AMD_EXAMPLES_COMPILE") == "1":      # also: generate the device code and compile it for gfx950
        from pockit_amd import hipbuild
        from pockit_amd.codegen import ModelSource

        t0 = time.time()
        hipbuild.compile_model(ModelSource(system.plan).source, fastmath=system._fastmath, keep_source=False)
        out["hipcc_seconds"] = round(time.time() - t0, 1)
    it = Interp(system.plan, x, lam, sigma)
    out["err"] = max(rel(it.objective(), rsys.objective(x.copy())), rel(it.gradient(), rsys.gradient(x.copy())),
                     rel(it.constraints(), rsys.constraints(x.copy())), rel(it.jacobian(), rsys.jacobian(x.copy())),
                     rel(it.hessian(), rsys.hessian(x.copy(), lam, sigma)))
    return out


def main():
    only = sys.argv[1:]
    out = {}
    for name in sorted(p for p in os.listdir(EXAMPLES) if p.endswith(".py") and not p.startswith("_")):
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        try:
            out[name] = check(name)
        except Exception as exc:  # noqa: BLE001 -- reported per program
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        out[name]["seconds"] = round(time.time() - t0, 1)
        print(name, out[name], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
