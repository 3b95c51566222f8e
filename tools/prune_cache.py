#!/usr/bin/env python3
"""Drop code objects / sources from pockit_amd/_cache that the current kernel header can no longer serve.

The cache key holds the hash of csrc/pk_kernels.hip.h + pk_abi.h, so every header edit orphans all earlier
entries; they would otherwise travel to every GPU lease (46 MB at the end of round 1).  An index file
(_cache/index.json: key -> header hash) written by hipbuild.compile_model tells the generations apart;
entries without an index record are stale by definition."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pockit_amd import hipbuild  # noqa: E402


def main(emit_older_than=None):
    """``emit_older_than`` (a time stamp): also drop the memoised function bodies under _cache/emit that nothing has read or
    written since then (codegen touches a body whenever it serves one; tools/warm_cache.sh passes the time it started, after
    having generated every model of the GPU suite)."""
    d = hipbuild.CACHE_DIR
    if not os.path.isdir(d):
        return
    keep = hipbuild.live_keys()
    removed = 0
    for name in os.listdir(d):
        stem, ext = os.path.splitext(name)
        if name.endswith(".res.json"):
            stem, ext = name[: -len(".res.json")], ".res.json"
        if ext in (".hsacoz", ".hsaco", ".hip", ".res.json") and stem not in keep:
            os.remove(os.path.join(d, name))
            removed += 1
        elif ext == ".part":
            os.remove(os.path.join(d, name))
    hipbuild.write_index(keep)
    emit, dropped = os.path.join(d, "emit"), 0
    if emit_older_than is not None and os.path.isdir(emit):
        for name in os.listdir(emit):
            path = os.path.join(emit, name)
            if os.path.getmtime(path) < emit_older_than:
                os.remove(path)
                dropped += 1
    print(f"pruned {removed} stale files, {len(keep)} live code objects" + (f", {dropped} unused function bodies" if dropped else ""))


def keep_only(list_files):
    """``--keep-list a.txt [b.txt ...]``: drop every code object whose key is in none of the lists (one key per line, written
    by hipbuild.compile_model under POCKIT_AMD_USED_LOG: gpurun_out/used_keys.txt of a whole GPU suite run, plus the list of a
    local ``__graft_entry__.build()``).  A set, not a time stamp: objects that only the GPU box compiles (later models of
    tests that build several) and that were imported from gpurun_out/cache_new are in the suite's list like all others."""
    d = hipbuild.CACHE_DIR
    want = set()
    for f in list_files:
        with open(f) as fh:
            want |= {ln.split()[0] for ln in fh if ln.strip()}
    live = hipbuild.live_keys()
    if len(want & live) < 50:
        raise SystemExit(f"prune_cache: the lists name only {len(want & live)} live objects -- not a whole suite run; nothing removed")
    removed = 0
    for name in os.listdir(d):
        for ext in (".hsacoz", ".res.json", ".gen", ".hip"):
            if name.endswith(ext) and name[: -len(ext)] in live and name[: -len(ext)] not in want:
                os.remove(os.path.join(d, name))
                removed += 1
    print(f"kept {len(want & live)} of {len(live)} live code objects, removed {removed} files")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--keep-list":
        keep_only(sys.argv[2:])
    else:
        main(float(sys.argv[1]) if len(sys.argv) > 1 else None)
