#!/bin/bash
# GPU box: where does the single pass overtake the pass-parallel Hessian role of the humanoid as the mesh grows?
M="humanoid_wbc:1000:8 humanoid_wbc:1500:8 humanoid_wbc:2000:8 humanoid_wbc:2500:8 humanoid_wbc:3000:8 humanoid_wbc:4000:8 humanoid_wbc:5000:8"
for cap in 32 16; do
  echo "== POCKIT_AMD_GROUP_CAP=$cap"
  POCKIT_AMD_GROUP_CAP=$cap python tools/cycle_probe.py $M 2>&1 | grep -v amdgpu.ids | cut -c1-130
done
