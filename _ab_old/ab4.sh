#!/bin/bash
for rep in 1 2; do
for v in 0 1; do
  POCKIT_AMD_STATIC_TABS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload two_stage_rocket --intervals 1000 > /tmp/ab_out.json 2>/tmp/ab_err.txt; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);r=d['roofline'];print('static_tabs=$v', 'rocket cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3), 'isolated', round(r.get('dispatch_isolated_us', r.get('avg_launch_us')),3))" || tail -3 /tmp/ab_err.txt
done; done
(cd _ab_x2 && python3 -c "
from pockit_amd import hipbuild
hipbuild.build_runtime(force=True)" && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload two_stage_rocket --intervals 1000 > /tmp/ab_out.json 2>/tmp/ab_err.txt; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);print('tree 29ea19e rocket cycles/s', round(d['value']))")
