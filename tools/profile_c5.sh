#!/bin/bash
# GPU box: rocprofv3 kernel stats and the two HBM-traffic PMC passes of pk_cycle at C5 (humanoid LGR 5000 x 8), same recipe as
# tools/profile_round.sh.   usage: tools/profile_c5.sh <tag>
tag=${1:-r04_c5}
repo=${GRAFT_REPO_ROOT:-$PWD}
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-end-to-end --workload humanoid_wbc --intervals 5000"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o p -- python3 "$repo/bench.py" $ARGS > "$out/${tag}_bench_under_rocprofv3.json" 2> "$out/${tag}_rocprof.err"
f=$(find "$out/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -3 "$f" | cut -c1-160
find "$out/prof_$tag" -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/prof_${tag}_pmc_$c" -o p -- python3 "$repo/bench.py" $ARGS > /dev/null 2>&1
  g=$(find "$out/prof_${tag}_pmc_$c" -name '*counter_collection.csv' | head -1)
  [ -n "$g" ] && python3 - "$g" $c <<'PY'
import csv, sys
src, name = sys.argv[1:3]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(src)) if r["Counter_Name"] == name and r["Kernel_Name"].startswith("pk_cycle")]
v = v[len(v) // 4:]
print(name, "KB per pk_cycle launch:", sum(v) / max(len(v), 1), "launches", len(v))
PY
  rm -rf "$out/prof_${tag}_pmc_$c"
done
