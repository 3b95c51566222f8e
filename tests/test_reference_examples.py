"""The reference's example programs as a drop-in test: every program of /root/reference/examples runs unchanged with
``pockit`` resolving to ``pockit_amd`` and hands the solver the same guess, options, bounds, structures and callback values
as it does with the reference itself (tests/golden/check_examples.py).  Needs the reference, so it runs in the build
container only (skipped where /root/reference is absent, e.g. on the GPU box); the reference is imported in a child
process, never into the test process, and nothing of it is copied."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


SLOW = ("drone_stabilization", "humanoid_whole_body_control", "orbit_transfer", "rocket_powered_descent")


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="the reference is only present in the build container")
def test_reference_example_programs_build_the_same_nlp_with_this_package():
    """All 33 programs.  Four of them take a minute each in the NumPy execution of their plan: the programs are dealt to four
    child processes, one of the four in each."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    names = sorted(p[:-3] for p in os.listdir("/root/reference/examples") if p.endswith(".py") and not p.startswith("_"))
    assert len(names) == 33
    rest = [n for n in names if n not in SLOW]
    shares = [[SLOW[k] + ".py"] + [n + ".py" for n in rest[k::4]] for k in range(4)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "golden", "check_examples.py")] + share, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for share in shares]
    out = {}
    for proc in procs:
        stdout, stderr = proc.communicate(timeout=1500)
        assert proc.returncode == 0, stderr[-3000:]
        out.update(json.loads(stdout.strip().splitlines()[-1]))
    assert len(out) == 33
    for name, r in out.items():
        assert "error" not in r, (name, r.get("error"))
        assert r["guess"] <= 1e-10, (name, "initial guess", r["guess"])
        assert r["options"], name + ": solver options differ"
        assert r["bounds"], name + ": bounds or sizes differ"
        assert r["structure"], name + ": triplet structure differs from the reference's"
        assert r["adapter"], name + ": the adapter's plan of the configured reference system differs"
        assert r["err"] <= 1e-11, (name, r["err"])
