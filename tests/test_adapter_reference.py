"""``pockit_amd.adapter.plan_from_reference_system`` (the reference-side binding of INTEGRATION.md section 2) against the
golden vectors.  Needs the reference itself, so it runs in the build container only (skipped where /root/reference is
absent, e.g. on the GPU box); the reference is imported in a child process, never into the test process."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/pockit"), reason="the reference is only present in the build container")
def test_converted_reference_systems_reproduce_the_golden_vectors():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "check_adapter.py")], capture_output=True, text=True,
                         env=env, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(out) >= 20
    for name, r in out.items():
        assert r["structure"], name + ": triplet structure differs from the reference's"
        assert r["err"] <= 1e-11, (name, r["err"])


def test_adapter_module_does_not_import_the_reference():
    import pockit_amd.adapter as adapter

    src = open(adapter.__file__).read()
    assert "import pockit\n" not in src and "from pockit " not in src and "from pockit." not in src
