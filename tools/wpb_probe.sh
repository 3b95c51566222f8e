#!/bin/bash
# GPU box (scratch copy of the tree): waves per workgroup of the tile kernels -- library and code objects rebuilt per value.
set -e
for wpb in ${WPBS:-4 2 1}; do
  echo "== POCKIT_AMD_WPB=$wpb"
  POCKIT_AMD_WPB=$wpb python -c "from pockit_amd import hipbuild; hipbuild.build_runtime(force=True)"
  POCKIT_AMD_WPB=$wpb python tools/cycle_probe.py C3 C5 C4 2>&1 | grep -v amdgpu.ids
done
python -c "from pockit_amd import hipbuild; hipbuild.build_runtime(force=True)"
