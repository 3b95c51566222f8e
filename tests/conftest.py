import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """How much of the session went into hipcc: the GPU suite is sized for a warm code-object cache (pockit_amd/_cache, filled
    without a GPU by tools/warm_cache.sh and shipped with the tree); objects compiled here mean the cache was cold for them --
    e.g. after an edit of the kernel header, which is part of every object's key."""
    try:
        from pockit_amd import hipbuild
    except Exception:  # noqa: BLE001
        return
    c = hipbuild.COMPILE_SECONDS
    if c["count"]:
        terminalreporter.write_line(f"[pockit_amd] {c['count']} code object(s) were compiled in this session ({c['total']:.0f} s of hipcc): "
                                    f"cold cache entries -- run tools/warm_cache.sh before a GPU lease")
