#!/bin/bash
# GPU box: where the nontemporal form of the 16-byte streaming stores starts to win over the write-through form (the default
# inside the Infinity Cache): quadrotor and humanoid over output sizes from 100 to 560 MB, alternating runs on one box.
mkdir -p gpurun_out
for wl in "planar_quadrotor:20000" "planar_quadrotor:30000" "planar_quadrotor:40000" "planar_quadrotor:50000" "planar_quadrotor:60000" "humanoid_wbc:5000" "humanoid_wbc:8000" "humanoid_wbc:10000" "humanoid_wbc:14000" "humanoid_wbc:20000"; do
  IFS=: read name iv <<< "$wl"
  for stream in sc1 nt; do
    POCKIT_AMD_STREAM=$stream python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > gpurun_out/policy.json 2>gpurun_out/policy.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/policy.json').read().strip().splitlines()[-1]);r=d['roofline'];ob=8*(1+sum(int(t.split('=')[1].strip(' ),')) for t in d['config']['workload'].split(';')[1].split(',') if t.split('=')[0].strip() in ('n','m','nnz_J','nnz_H')))
print('$name $iv'.ljust(24), 'outputs MB', str(round(ob/1e6)).rjust(4), 'streaming stores ${stream}'.ljust(24), 'us/cycle', str(round(d['ms_per_step']*1e3,2)).rjust(8), 'frac', round(r['frac'],3))"
  done
done
