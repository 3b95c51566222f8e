# Developer helper (GPU box): pk_cycle's duration with parts of the launch switched off (POCKIT_AMD_DEBUG_FLAGS):
# 0 full, 65536 no finalize workgroup, 2048 no boundary workgroups, 67584 neither, 256 no phase after the barrier
for f in ${FLAGS:-0 65536 2048 67584 256}; do
  POCKIT_AMD_DEBUG_FLAGS=$f python3 bench.py --no-cpu-baseline --no-extra --workload ${WORKLOAD:-planar_quadrotor} --intervals ${INTERVALS:-2000} > gpurun_out/flags_$f.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('gpurun_out/flags_$f.json').read().strip().splitlines()[-1]);print('flags=$f', 'cycles/s', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,2), 'pk_cycle us', round(d['roofline']['avg_launch_us'],2))"
done
