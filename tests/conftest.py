import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # a session on a GPU box (gpurun_out/ exists): log every code object that had to be compiled, in this process or a child
    out = os.path.join(ROOT, "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT") and os.path.isdir(out) and "POCKIT_AMD_COMPILE_LOG" not in os.environ:
        os.environ["POCKIT_AMD_COMPILE_LOG"] = os.path.join(out, "compiled_keys.txt")
        os.environ.setdefault("POCKIT_AMD_USED_LOG", os.path.join(out, "used_keys.txt"))      # every object the suite was served
        for name in ("POCKIT_AMD_COMPILE_LOG", "POCKIT_AMD_USED_LOG"):
            try:
                os.remove(os.environ[name])
            except OSError:
                pass


def pytest_terminal_summary(terminalreporter):
    """How much of the session went into hipcc: the GPU suite is sized for a warm code-object cache (pockit_amd/_cache, filled
    without a GPU by tools/warm_cache.sh and shipped with the tree).  Objects compiled here mean the cache was cold for them
    (tests that build several models stop at the first one when warmed without a GPU).  On a GPU box they are copied to
    gpurun_out/cache_new/ (<= 48 MB), from where ``cp gpurun_out/cache_new/* pockit_amd/_cache/`` completes the local cache."""
    log = os.environ.get("POCKIT_AMD_COMPILE_LOG")
    if not log or not os.path.exists(log):
        return
    rows = [ln.split() for ln in open(log) if ln.strip()]
    if not rows:
        return
    total = sum(float(r[1]) for r in rows if len(r) > 1)
    terminalreporter.write_line(f"[pockit_amd] {len(rows)} code object(s) were compiled in this session ({total:.0f} s of hipcc): "
                                f"cold cache entries")
    try:
        import shutil

        from pockit_amd import hipbuild

        dst = os.path.join(os.path.dirname(log), "cache_new")
        os.makedirs(dst, exist_ok=True)
        used = 0
        for key in dict.fromkeys(r[0] for r in rows):
            for ext in (".hsacoz", ".gen", ".res.json"):
                src = os.path.join(hipbuild.CACHE_DIR, key + ext)
                if os.path.exists(src) and used + os.path.getsize(src) < 48 * 2**20:
                    shutil.copy(src, dst)
                    used += os.path.getsize(src)
        terminalreporter.write_line(f"[pockit_amd] copied {used / 2**20:.1f} MB of new objects to {dst}")
    except Exception as exc:  # noqa: BLE001
        terminalreporter.write_line(f"[pockit_amd] could not export the new objects: {exc!r}")
