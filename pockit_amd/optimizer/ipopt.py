"""IPOPT adapter: ``solve(system, guess, optimizer_options)`` as ``pockit.optimizer.ipopt.solve``
(/root/reference/pockit/optimizer/ipopt.py:11-61).  The ``problem_obj`` handed to cyipopt is the
``System`` itself, whose callbacks run on the GPU.

Layouts.  IPOPT needs no particular triplet list, only that ``jacobianstructure()`` / ``hessianstructure()`` and the values
agree (it sums repeated positions itself).  The reference hands it one Hessian triplet per nonzero of the integration matrix
and derivative entry (phasebase.py:1280-1285: 4.3x more values than distinct positions at 12k nodes, 6.6x at 40k), and those
values cross PCIe every iteration.  By default (``layout="auto"``) the solve therefore runs on the COMPACT layouts (one value
per distinct position of a node) whenever the model has them and the reference's Jacobian +
Hessian values are at least ``AUTO_COMPACT_BYTES`` per iterate: the same matrices, fewer bytes (+16 % cycles/s at 12k nodes,
+70 % at 40k, x2 for a brachistochrone at 10k).  Below that size -- most of the reference's example programs -- an iterate
costs what its launches cost, and the compact kernels' chain is 3-8 % longer than the reference layout's
(tools/layout_crossover_probe.py, profiles/r04_zn_layout_crossover.txt): the reference layouts stay.  ``layout="compact"`` /
``layout="reference"`` force one or the other; the system's own layout settings are restored when the solve returns."""
from __future__ import annotations

from ._common import postprocess, preprocess


AUTO_COMPACT_BYTES = 1 << 20      # measured crossover: 0.67 MB 1.03, 1.3 MB 0.98, 2.1 MB 0.93 (compact / reference iterate time)


def solve(system, guess, optimizer_options=None, *, layout="auto"):
    if layout not in ("auto", "compact", "reference"):
        raise ValueError('layout must be "auto", "compact" or "reference"')
    try:
        import cyipopt
    except ImportError as exc:  # cyipopt / Ipopt are third-party and not part of this package
        raise ImportError("pockit_amd.optimizer.ipopt needs cyipopt (pip install cyipopt) and Ipopt") from exc
    x_0, guess_is_variable, optimizer_options = preprocess(system, guess, optimizer_options)
    keep = (system._hessian_layout, system._jacobian_layout)
    if layout == "auto":
        layout = "compact" if 8 * (system.plan.nnz_J + system.plan.nnz_H) >= AUTO_COMPACT_BYTES else "reference"
    if layout == "compact" and not system.plan.outer:      # (models nonlinear in the integrals keep the reference layout)
        if system.evaluator.src.compact:                   # (so does a model one of whose entries couples too many states)
            system.set_hessian_layout("compact")
        system.set_jacobian_layout("compact")
    else:                                                  # "reference" FORCES the reference's lists, whatever the system was set to
        system.set_hessian_layout("reference")
        system.set_jacobian_layout("reference")
    try:
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
    finally:
        system.set_hessian_layout(keep[0])
        system.set_jacobian_layout(keep[1])
    return postprocess(system, x, guess_is_variable), info
