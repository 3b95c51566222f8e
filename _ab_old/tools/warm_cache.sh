#!/bin/bash
# Fill pockit_amd/_cache with the gfx950 code objects of the GPU tests' models WITHOUT a GPU (build container):
# every GPU test is started on the CPU; its evaluator compiles (or finds) the model's code object and then fails
# loudly for lack of a device -- the failures are expected, the cache is what is kept.  Run before a gpurun call
# after a kernel-header change, so that the GPU box does not spend its lease on hipcc.
cd "$(dirname "$0")/.." || exit 1
python3 tools/prune_cache.py
python3 -m pytest tests -m gpu -q -n "${JOBS:-6}" -p no:cacheprovider > /dev/null 2>&1
python3 -c "import __graft_entry__ as g; g.build()"
ls pockit_amd/_cache/*.hsaco | wc -l
