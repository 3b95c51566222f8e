#!/usr/bin/env python3
"""Build-container only: golden vectors of seeded RANDOM models (tests/random_models.py) from the imported reference.

For every seed the model is built through the REFERENCE's modeling API (``pockit.radau`` / ``pockit.lobatto`` behind the
identity-njit stub of refharness/, as in make_golden.py) and its seven callbacks are evaluated at a seeded point:
tests/golden/random/<scheme>_<seed>.npz = x, lambda, sigma -> f, grad f, g, J, H, the triplet structures and the bounds.
Nothing of the reference is stored but inputs and outputs.  Usage: python tests/golden/make_random.py"""
import os
import sys
import types
import typing

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
if not hasattr(typing, "Self"):
    typing.Self = typing.TypeVar("Self")
sys.path.insert(0, os.path.join(HERE, "refharness"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.modules.setdefault("cyipopt", types.ModuleType("cyipopt"))

import numpy as np  # noqa: E402

import pockit.lobatto as ref_lobatto  # noqa: E402
import pockit.radau as ref_radau  # noqa: E402

import random_models as rm  # noqa: E402

NS = {"radau": ref_radau, "lobatto": ref_lobatto}
OUT = os.path.join(HERE, "random")


def main():
    os.makedirs(OUT, exist_ok=True)
    for scheme, seeds in rm.SEEDS.items():
        for seed in seeds:
            system, _ = rm.random_model(NS[scheme], seed, scheme)
            x, lam, sigma = rm.random_inputs(system, seed)
            jr, jc = system.jacobianstructure()
            hr, hc = system.hessianstructure()
            out = dict(x=x, lam=lam, sigma=np.float64(sigma), f=np.float64(system.objective(x.copy())), grad=system.gradient(x.copy()),
                       g=system.constraints(x.copy()), J=system.jacobian(x.copy()), H=system.hessian(x.copy(), lam, sigma),
                       jr=np.asarray(jr, np.int32), jc=np.asarray(jc, np.int32), hr=np.asarray(hr, np.int32), hc=np.asarray(hc, np.int32),
                       v_lb=system.v_lb, v_ub=system.v_ub, c_lb=system.c_lb, c_ub=system.c_ub)
            for k in ("grad", "g", "J", "H"):
                assert np.all(np.isfinite(out[k])), (scheme, seed, k)
            np.savez_compressed(os.path.join(OUT, f"{scheme}_{seed}.npz"), **out)
            print(scheme, seed, "n", x.size, "m", lam.size, "nnz_J", len(jr), "nnz_H", len(hr), "phases", system.n_p,
                  "n_s", system.n_s, flush=True)


if __name__ == "__main__":
    main()
