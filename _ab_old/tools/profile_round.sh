# GPU box: the round's evidence for profiles/ -- rocprofv3 kernel stats of the default bench command and the two
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; each in its own run, no trace domains with --pmc).
# usage: bash tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o q -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.json 2>/dev/null
cp $OUT/stats/q_kernel_stats.csv $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > /dev/null 2>&1
  cp $OUT/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
done
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/pmc_%s.csv" % c)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 4:]                      # drop the warm-up launches
        out.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
