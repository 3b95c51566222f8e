#!/bin/bash
# GPU box: A/B of how pk_cycle reads its PkArgs -- lazily from the kernarg segment (default) or as a by-value kernel
# argument loaded en bloc on entry (POCKIT_AMD_KA_LAZY=0) -- on the benchmark workloads, alternating runs on one box.
for rep in 1 2; do
for v in lazy byvalue; do
  [ $v = lazy ] && envs="POCKIT_AMD_KA_LAZY=1" || envs="POCKIT_AMD_KA_LAZY=0"
  for wl in ${WORKLOADS:-"planar_quadrotor:2000" "brachistochrone:1250" "brachistochrone:200" "humanoid_wbc:5000" "two_stage_rocket:1000"}; do
    IFS=: read name iv <<< "$wl"
    env $envs python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > gpurun_out/ka_$v.json 2>gpurun_out/ka_$v.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/ka_$v.json').read().strip().splitlines()[-1]);print('rep $rep', '$v'.ljust(8), '$name $iv'.ljust(24), 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3), 'isolated us', round(d['roofline']['dispatch_isolated_us'],3))"
  done
done
done
