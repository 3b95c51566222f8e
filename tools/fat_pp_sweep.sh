#!/bin/bash
# GPU box: the large example models with a group size of 16 and every pass as a workgroup of its own, over intervals per wave.
set -e
for ipw in 6 10 16; do
  echo "== POCKIT_AMD_GROUP_CAP=16 POCKIT_AMD_PASS_PARALLEL=1 POCKIT_AMD_IPW=$ipw"
  POCKIT_AMD_GROUP_CAP=16 POCKIT_AMD_PASS_PARALLEL=1 POCKIT_AMD_IPW=$ipw python tools/fat_model_probe.py ${1:-2000} ${2:-4}
done
