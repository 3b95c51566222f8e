#!/bin/bash
for wl in "two_stage_rocket 1000" "planar_quadrotor 2000"; do
  set -- $wl
  for v in 0 1 0 1; do
    POCKIT_AMD_SHARDED=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > /tmp/ab_out.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);print('sharded_code=$v', '$1', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3))"
  done
  (cd _ab_old && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > /tmp/ab_out.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);print('old tree', '$1', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3))")
done
