#!/bin/bash
# GPU box: the headline workload with its derivative set forced into groups and every pass as a workgroup of its own.
for cap in 3 4 6 8; do
  echo "== POCKIT_AMD_GROUP_CAP=$cap POCKIT_AMD_PASS_PARALLEL=1"
  POCKIT_AMD_GROUP_CAP=$cap POCKIT_AMD_PASS_PARALLEL=1 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'])"
done
