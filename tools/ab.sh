#!/bin/bash
# GPU box: A/B of build / run switches on the benchmark workloads, alternating runs on one box.
#   VARIANTS="name[:ENV=val[,ENV=val...]] ..."   (a name alone = the defaults)
#   WORKLOADS="workload:intervals ..."  REPS=n
# e.g. VARIANTS="auto seq:POCKIT_AMD_PASS_PARALLEL=0" tools/ab.sh      (the surviving switches: INTEGRATION.md section 7)
for rep in $(seq 1 ${REPS:-2}); do
for spec in ${VARIANTS:-default}; do
  v=${spec%%:*}
  envs=""
  [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  for wl in ${WORKLOADS:-"planar_quadrotor:2000" "brachistochrone:1250" "brachistochrone:200" "humanoid_wbc:5000" "two_stage_rocket:1000"}; do
    IFS=: read name iv <<< "$wl"
    env $envs python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]);print('rep $rep', '$v'.ljust(10), '$name $iv'.ljust(24), 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3), 'isolated us', round(d['roofline']['dispatch_isolated_us'],3))"
  done
done
done
