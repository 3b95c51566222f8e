#!/bin/bash
# A/B on ONE box: the tree of commit 92fac92 (start of round 2: 211k in gpurun_out/r02_a) against the working tree
cd _ab_old && python3 -c "
from pockit_amd import hipbuild
hipbuild.build_runtime(force=True)" && cd ..
for i in 1 2 3; do
  for d in _ab_old .; do
    (cd $d && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end > /tmp/ab_out.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);r=d['roofline'];print('tree=$d', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3), 'isolated', round(r.get('dispatch_isolated_us', r.get('avg_launch_us')),3))")
  done
done
for wl in "two_stage_rocket 1000" "humanoid_wbc 5000"; do
  set -- $wl
  for d in _ab_old . _ab_old .; do
    (cd $d && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > /tmp/ab_out.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]);print('tree=$d', '$1', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3))")
  done
done
