#!/bin/bash
for wl in "planar_quadrotor 2000" "two_stage_rocket 1000" "brachistochrone 1250"; do
  set -- $wl
  for ipw in 3 4 5 6 7 8 10; do
    POCKIT_AMD_IPW=$ipw python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > /tmp/o.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('$1 ipw=$ipw', 'tiles', d['config']['tiles'], 'cycles/s', round(d['value']), 'us', round(d['ms_per_step']*1e3,3))"
  done
  POCKIT_AMD_SPLIT=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > /tmp/o.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('$1 split=0 default ipw', 'cycles/s', round(d['value']), 'us', round(d['ms_per_step']*1e3,3))"
done
