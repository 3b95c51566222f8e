#!/bin/bash
# GPU box: the cycle over problem sizes -- cycles/s, us per cycle, algorithmic GB/s and fraction of the 8 TB/s HBM peak for the
# quadrotor (6 points per interval) from 600 to 360 000 nodes and the humanoid (8 points) from 800 to 160 000 nodes.
for wl in "planar_quadrotor:100" "planar_quadrotor:400" "planar_quadrotor:1000" "planar_quadrotor:2000" "planar_quadrotor:4000" "planar_quadrotor:8000" "planar_quadrotor:20000" "planar_quadrotor:60000" \
          "humanoid_wbc:100" "humanoid_wbc:500" "humanoid_wbc:1500" "humanoid_wbc:5000" "humanoid_wbc:20000"; do
  IFS=: read name iv <<< "$wl"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > gpurun_out/size.json 2>gpurun_out/size.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/size.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$name'.ljust(18), 'intervals', str($iv).rjust(6), 'cycles/s', str(round(d['value'])).rjust(7), 'us/cycle', str(round(d['ms_per_step']*1e3,2)).rjust(7), 'MB/cycle', str(round(r['algorithmic_bytes_per_launch']/1e6,2)).rjust(8), 'GB/s', str(round(r['achieved'])).rjust(5), 'frac', round(r['frac'],3), d['device_resident']['timing']['batch_launch']['form'])"
done
