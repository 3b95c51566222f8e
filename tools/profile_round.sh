#!/bin/bash
# GPU box: the round's evidence for profiles/ -- rocprofv3 kernel stats of the driver's bench command (headline workload)
# and the two HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; each in its own run, no trace domains with --pmc).
# usage: tools/profile_round.sh <tag> [pmc]     -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprofv3.json,
#                                                  with "pmc": <tag>_pmc_FETCH_SIZE.csv, <tag>_pmc_WRITE_SIZE.csv, <tag>_pmc_summary.json
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-$PWD}
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-end-to-end"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o p -- \
  python3 "$repo/bench.py" $BENCH_ARGS > "$out/${tag}_bench_under_rocprofv3.json" 2> "$out/${tag}_rocprof.err"
rc=$?
f=$(find "$out/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -4 "$f" | cut -c1-160
find "$out/prof_$tag" -name '*kernel_trace.csv' -delete      # the trace itself is large: keep only the stats
# the headline's own command (the host-landed cycle: five callbacks, copies up and down): which kernels it runs and how long
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_e2e" -o p -- \
  python3 "$repo/bench.py" --steps 20 --warmup 5 --no-extra --no-cpu-baseline > "$out/${tag}_bench_e2e_under_rocprofv3.json" 2> "$out/${tag}_rocprof_e2e.err"
f=$(find "$out/prof_${tag}_e2e" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats_e2e.csv"
find "$out/prof_${tag}_e2e" -name '*kernel_trace.csv' -delete
[ "$2" = "pmc" ] || exit $rc
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/prof_${tag}_pmc_$c" -o p -- python3 "$repo/bench.py" $BENCH_ARGS > /dev/null 2>&1
  g=$(find "$out/prof_${tag}_pmc_$c" -name '*counter_collection.csv' | head -1)
  [ -n "$g" ] && python3 - "$g" "$out/${tag}_pmc_$c.csv" $c <<'PY'
import csv, sys
src, dst, name = sys.argv[1:4]
rows = [r for r in csv.DictReader(open(src)) if r["Counter_Name"] == name and r["Kernel_Name"].startswith("pk_")]
with open(dst, "w", newline="") as fh:          # one line per dispatch of our kernels: kernel, dispatch id, counter value
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Dispatch_Id", "Counter_Name", "Counter_Value"])
    for r in rows:
        w.writerow([r["Kernel_Name"], r.get("Dispatch_Id", ""), name, r["Counter_Value"]])
PY
  rm -rf "$out/prof_${tag}_pmc_$c"
done
python3 - "$out" "$tag" <<'PY'
import collections, csv, json, sys
out, tag = sys.argv[1:3]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(f"{out}/{tag}_pmc_{c}.csv")):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    except OSError:
        continue
    for k, v in acc.items():
        v = v[len(v) // 4:]                      # drop the warm-up launches
        res.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
        res[k][c + "_launches"] = len(v)
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
PY
exit $rc
