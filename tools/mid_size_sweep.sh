#!/bin/bash
# GPU box: intervals per wave and split / unsplit x-part between 12k and 60k nodes (quadrotor, 6 points per interval)
for iv in 3000 4000 6000 8000 10000; do
  for split in 1 0; do
    for ipw in 6 8 10; do
      POCKIT_AMD_IPW=$ipw POCKIT_AMD_SPLIT=$split python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload planar_quadrotor --intervals $iv 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('quadrotor $iv split $split ipw $ipw tiles', d['config']['tiles'], 'us', round(d['ms_per_step']*1e3,2), 'frac', round(d['roofline']['frac'],3))"
    done
  done
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload planar_quadrotor --intervals $iv 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('quadrotor $iv DEFAULT ipw', d['config']['intervals_per_wave'], 'tiles', d['config']['tiles'], 'us', round(d['ms_per_step']*1e3,2), 'frac', round(d['roofline']['frac'],3))"
done
