#!/bin/bash
# Fill pockit_amd/_cache with the gfx950 code objects of the GPU tests' models WITHOUT a GPU (build container):
# every GPU test is started on the CPU; its evaluator compiles (or finds) the model's code object and then fails
# loudly for lack of a device -- the failures are expected, the cache is what is kept.  Run before a gpurun call
# after a kernel-header change, so that the GPU box does not spend its lease on hipcc.  Afterwards the cache is pruned to
# what the current header and the current tests use (stale generations, function bodies nothing asked for): what ships.
cd "$(dirname "$0")/.." || exit 1
stamp=$(python3 -c "import time; print(time.time() - 5)")
python3 tools/prune_cache.py
python3 -m pytest tests -m gpu -q -n "${JOBS:-6}" -p no:cacheprovider > /dev/null 2>&1
python3 -c "import __graft_entry__ as g; g.build()"
python3 tools/prune_cache.py "$stamp"
ls pockit_amd/_cache/*.hsacoz | wc -l
du -sh pockit_amd/_cache
