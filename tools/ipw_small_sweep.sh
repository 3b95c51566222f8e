#!/bin/bash
# GPU box: intervals per wave 1 ... 8 on small meshes (50 ... 400 intervals) of the benchmark models: us per cycle.
for wl in "brachistochrone:200" "planar_quadrotor:100" "planar_quadrotor:400" "humanoid_wbc:100" "humanoid_wbc:200" "two_stage_rocket:100" "brachistochrone:50"; do
  IFS=: read name iv <<< "$wl"
  for ipw in 1 2 3 4 6 8; do
    POCKIT_AMD_IPW=$ipw python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name $iv ipw $ipw tiles', d['config']['tiles'], 'us', round(d['ms_per_step']*1e3,2))"
  done
done
