# Developer helper (GPU box): output store flavour A/B (plain / nontemporal / agent-scope write-through / system scope)
for nt in ${NTS:-0 1 2 3}; do
  POCKIT_AMD_NT=$nt python3 bench.py --no-cpu-baseline > gpurun_out/nt_$nt.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/nt_$nt.json").read().strip().splitlines()[-1])
d={**json.load(open("bench_detail.json")), **d}      # (the long tables live in the detail file)
print("nt",$nt, round(d["value"]), {k:round(v,2) for k,v in d["kernel_us"].items() if v}, "|", " ".join(f"{k.split('_')[0][:5]}{k.split('_')[-1]}:{round(v['cycles_per_s'])}" for k,v in d["other_workloads"].items()))
PY
done
