# Developer helper (GPU box): cycles/s of the headline and side workloads for several intervals-per-wave tilings,
# with the x-part split (POCKIT_AMD_SPLIT=1: values / Jacobian waves) and unsplit
for split in ${SPLITS:-1 0}; do
for ipw in ${IPWS:-1 2 3 4 5 7 10}; do
  POCKIT_AMD_SPLIT=$split POCKIT_AMD_IPW=$ipw python3 bench.py --no-cpu-baseline > gpurun_out/ipw_${split}_$ipw.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/ipw_${split}_$ipw.json").read().strip().splitlines()[-1])
d={**json.load(open("bench_detail.json")), **d}      # (the long tables live in the detail file)
print("split",$split,"ipw",$ipw, round(d["value"]), {k:round(v,2) for k,v in d["kernel_us"].items() if v}, "|", " ".join(f"{k.split('_')[0][:5]}{k.split('_')[-1]}:{round(v['cycles_per_s'])}" for k,v in d["other_workloads"].items()))
PY
done
done
