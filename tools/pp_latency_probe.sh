#!/bin/bash
# GPU box: does splitting the roles of a moderately fat single-pass model (humanoid) into pass-parallel workgroups shorten the
# cycle on small and medium meshes?  (latency of one wave's chain against the number of workgroups)
M="humanoid_wbc:25:8 humanoid_wbc:100:8 humanoid_wbc:500:8 humanoid_wbc:2000:4 C5 planar_quadrotor:100:6 planar_quadrotor:500:6"
echo "== default"; python tools/cycle_probe.py $M 2>&1 | grep -v amdgpu.ids
for cap in 16 8; do
  echo "== POCKIT_AMD_GROUP_CAP=$cap POCKIT_AMD_PASS_PARALLEL=1"
  POCKIT_AMD_GROUP_CAP=$cap POCKIT_AMD_PASS_PARALLEL=1 python tools/cycle_probe.py $M 2>&1 | grep -v amdgpu.ids
done
