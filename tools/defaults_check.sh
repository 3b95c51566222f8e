#!/bin/bash
# GPU box: what the DEFAULT tiling (evaluator._intervals_per_wave) gives over problem sizes and models -- intervals per wave, tiles,
# us per cycle, fraction of the HBM peak (to be read next to tools/ipw_small_sweep.sh and tools/mid_size_sweep.sh).
for wl in "planar_quadrotor:100" "planar_quadrotor:1000" "planar_quadrotor:2000" "planar_quadrotor:3000" "planar_quadrotor:4000" "planar_quadrotor:6000" "planar_quadrotor:8000" "planar_quadrotor:10000" "planar_quadrotor:20000" "brachistochrone:200" "brachistochrone:1250" "two_stage_rocket:1000" "humanoid_wbc:100" "humanoid_wbc:1500" "humanoid_wbc:5000" "planar_quadrotor_lgl:2000"; do
  IFS=: read name iv <<< "$wl"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-end-to-end --workload $name --intervals $iv > /dev/null 2>&1
  python3 -c "
import json;d=json.load(open('bench_detail.json'));print('$name $iv default ipw', d['config']['intervals_per_wave'], 'tiles', d['config']['tiles'], 'us', round(d['device_resident']['ms_per_step']*1e3,2), 'frac', round(d['roofline']['frac'],3))"
done
