#!/bin/bash
# GPU box: store policy of the cycle BEYOND the Infinity Cache.  At 360 000 quadrotor nodes one launch
# writes 312 MB -- more than the 256 MiB MALL --, the only point of the size sweep that is bound by HBM itself (0.53 of the
# roof with the default write-through stores, which were chosen inside the MALL regime).  Variants: the 16-byte streaming
# stores' cache policy (POCKIT_AMD_STREAM) x the 8-byte stores' flavour (POCKIT_AMD_NT: 0 plain, 1 nontemporal, 2 agent scope =
# default, 3 system scope), alternating runs on one box.  Also the humanoid at 160 000 nodes (556 MB of outputs).
mkdir -p gpurun_out
for rep in 1 2; do
for wl in "planar_quadrotor:60000" "humanoid_wbc:20000"; do
  IFS=: read name iv <<< "$wl"
  for v in "sc1:2" "sc1nt:2" "nt:1" "nt:2" "plain:0" "plain:2" "sc0sc1:3" "sc1nt:1"; do
    IFS=: read stream nt <<< "$v"
    POCKIT_AMD_STREAM=$stream POCKIT_AMD_NT=$nt python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > gpurun_out/policy.json 2>gpurun_out/policy.err
    python3 -c "
import json;d=json.loads(open('gpurun_out/policy.json').read().strip().splitlines()[-1]);r=d['roofline'];print('rep $rep', '$name $iv'.ljust(24), 'streaming stores ${stream}'.ljust(26), 'other stores NT=${nt}', 'us/cycle', str(round(d['ms_per_step']*1e3,2)).rjust(8), 'GB/s of B', str(round(r['achieved'])).rjust(5), 'frac', round(r['frac'],3), 'regime', r['regime'])"
  done
done
done
