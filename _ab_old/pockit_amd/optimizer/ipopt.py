"""IPOPT adapter: ``solve(system, guess, optimizer_options)`` as ``pockit.optimizer.ipopt.solve``
(/root/reference/pockit/optimizer/ipopt.py:11-61).  The ``problem_obj`` handed to cyipopt is the
``System`` itself, whose callbacks run on the GPU."""
from __future__ import annotations

from ._common import postprocess, preprocess


def solve(system, guess, optimizer_options=None):
    try:
        import cyipopt
    except ImportError as exc:  # cyipopt / Ipopt are third-party and not part of this package
        raise ImportError("pockit_amd.optimizer.ipopt needs cyipopt (pip install cyipopt) and Ipopt") from exc
    x_0, guess_is_variable, optimizer_options = preprocess(system, guess, optimizer_options)
    solver = cyipopt.Problem(n=int(system.L), m=len(system.c_lb), problem_obj=system, lb=system.v_lb,
                             ub=system.v_ub, cl=system.c_lb, cu=system.c_ub)
    for k, v in optimizer_options.items():
        solver.add_option(k, v)
    # cyipopt copies every callback result into Ipopt's own arrays immediately, so the evaluator may hand out
    # its pinned DMA buffers instead of fresh copies while the solver runs
    system.evaluator.zero_copy = True
    try:
        x, info = solver.solve(x_0)
    finally:
        system.evaluator.zero_copy = False
    return postprocess(system, x, guess_is_variable), info
