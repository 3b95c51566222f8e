"""CPU restatement (NumPy) of pockit's NLP-callback hot path -- TEST INFRASTRUCTURE ONLY.

This package is the *oracle* for the MI355X evaluator in ``pockit_amd``: a from-scratch NumPy
restatement of the reference algorithm (pockit v0.1.1 at /root/reference), each function citing
the reference file:line it follows.  It is pinned against golden vectors produced by importing
the unmodified reference in the build container (tests/golden/make_golden.py ->
tests/golden/small/*.npz, tests/golden/tables.npz, tests/golden/full.json), see
tests/test_oracle_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it -- and there only as the checker / the reported CPU baseline.  The product package
``pockit_amd`` never imports, links or executes anything under ``oracle/``; its evaluator fails
loudly when the HIP library is missing.

Layout
  tables.py      LGR/LGL nodes, weights, integration matrices        (reference: radau/lobatto discretization.py, discretizationbase.py)
  symfunc.py     SymPy expr -> sparse G/H with the reference ordering (reference: base/fastfunc.py)
  chain.py       sparse forward chain rule on a node DAG             (reference: base/easyderiv.py)
  mesh.py        per-phase index partitions, T/I COO splits           (reference: */discretization.py Discretization)
  phase.py       per-phase values/gradients/Hessians + layout         (reference: base/phasebase.py:41-1337)
  system.py      NLP assembly + the 7 cyipopt callbacks               (reference: base/systembase.py:16-835)
  refine.py      mesh error estimation + continuous hp-refinement    (reference: base/phasebase.py:1339-1437,1522-1617)
  variable.py    minimal Variable + constant/linear guesses           (reference: base/variablebase.py:92-470)
  radau.py / lobatto.py   the two user-facing namespaces (System, Phase, Variable, guesses)
"""
