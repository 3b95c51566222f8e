for ipw in 1 2 3 4 5 8; do
  POCKIT_AMD_IPW=$ipw python bench.py --no-cpu-baseline > gpurun_out/ipw_$ipw.json 2>/dev/null
done
python - <<PY
import json
for ipw in (1,2,3,4,5,8):
    d=json.loads(open("gpurun_out/ipw_%d.json"%ipw).read().strip().splitlines()[-1])
    print("ipw",ipw, round(d["value"]), {k:round(v,2) for k,v in d["kernel_us"].items() if v}, "|", " ".join(f"{k.split('_')[0][:5]}{k.split('_')[-1]}:{round(v['cycles_per_s'])}" for k,v in d["other_workloads"].items()))
PY
