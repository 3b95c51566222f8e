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


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else None)
