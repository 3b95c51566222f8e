#!/bin/bash
# GPU box: cycles/s of the headline workload (and the humanoid) for the cache-policy variants of the 16-byte streaming
# stores (POCKIT_AMD_STREAM at code generation: default sc1 = agent-scope write-through).
for v in ${VARIANTS:-sc1 sc1nt nt plain sc0sc1}; do
  for wl in "planar_quadrotor 2000" "humanoid_wbc 5000"; do
    set -- $wl
    POCKIT_AMD_STREAM=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $1 --intervals $2 > gpurun_out/stream_$v.json 2>/dev/null
    python3 -c "
import json;d=json.loads(open('gpurun_out/stream_$v.json').read().strip().splitlines()[-1]);print('stream=$v', '$1', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,3), 'isolated us', round(d['roofline']['dispatch_isolated_us'],3))"
  done
done
