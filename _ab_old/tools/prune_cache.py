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


def main():
    d = hipbuild.CACHE_DIR
    if not os.path.isdir(d):
        return
    keep = hipbuild.live_keys()
    removed = 0
    for name in os.listdir(d):
        stem, ext = os.path.splitext(name)
        if ext in (".hsaco", ".hip") and stem not in keep:
            os.remove(os.path.join(d, name))
            removed += 1
        elif ext == ".part":
            os.remove(os.path.join(d, name))
    hipbuild.write_index(keep)
    print(f"pruned {removed} stale files, {len(keep)} live code objects")


if __name__ == "__main__":
    main()
