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


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="the reference is only present in the build container")
def test_reference_example_programs_build_the_same_nlp_with_this_package():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "check_examples.py")], capture_output=True, text=True,
                         env=env, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(out) >= 25
    for name, r in out.items():
        assert "error" not in r, (name, r.get("error"))
        assert r["guess"] <= 1e-10, (name, "initial guess", r["guess"])
        assert r["options"], name + ": solver options differ"
        assert r["bounds"], name + ": bounds or sizes differ"
        assert r["structure"], name + ": triplet structure differs from the reference's"
        assert r["adapter"], name + ": the adapter's plan of the configured reference system differs"
        assert r["err"] <= 1e-11, (name, r["err"])
