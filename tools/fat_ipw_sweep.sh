#!/bin/bash
# GPU box: intervals per wave of the large example models at benchmark size (pass-parallel cycle: fuller waves?).
set -e
for ipw in "" 3 4 8 12 16; do
  echo "== POCKIT_AMD_IPW=${ipw:-default}"
  POCKIT_AMD_IPW=$ipw python tools/fat_model_probe.py ${1:-2000} ${2:-4}
done
