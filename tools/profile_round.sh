#!/bin/bash
# GPU box: the round's evidence for profiles/, ALL FROM ONE LEASE (the line's roofline and the committed
# rocprof summary must be a pair from the same box and minute):
#   1. the plain bench line of the driver's command                          -> gpurun_out/<tag>_bench_n1.json (+ _bench_detail.json)
#   2. rocprofv3 --kernel-trace --stats of the same device-resident loop     -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprofv3.json
#   3. (with "pmc") the two HBM-traffic PMC passes, each in its own run, no trace domains with --pmc
#                                                                           -> <tag>_pmc_FETCH_SIZE.csv, _WRITE_SIZE.csv, _pmc_summary.json
#   4. tools/update_pmc_traffic.py <tag>: profiles/pmc_traffic.json's pk_cycle / pk_cycle_profiled from THAT triple, and the
#      line of step 1 assembled again with them                              -> <tag>_bench_n1_with_profile.json
# usage: tools/profile_round.sh <tag> [pmc]
tag=${1:-r05}
repo=${GRAFT_REPO_ROOT:-$PWD}
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$repo/bench.py" --steps 20 --warmup 5 > "$out/${tag}_bench_n1.json" 2> "$out/${tag}_bench_n1.err" || exit 1
cp "$repo/bench_detail.json" "$out/${tag}_bench_detail.json" 2>/dev/null
tail -c 600 "$out/${tag}_bench_n1.json"
BENCH_ARGS="--steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-end-to-end"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o p -- \
  python3 "$repo/bench.py" $BENCH_ARGS > "$out/${tag}_bench_under_rocprofv3.json" 2> "$out/${tag}_rocprof.err"
rc=$?
f=$(find "$out/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv" && head -4 "$f" | cut -c1-160
find "$out/prof_$tag" -name '*kernel_trace.csv' -delete      # the trace itself is large: keep only the stats
if [ "$2" = "pmc" ]; then
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
print(json.dumps(res, indent=1)[:400])
json.dump(res, open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
PY
fi
python3 "$repo/tools/update_pmc_traffic.py" "$tag" "$out" "$out/pmc_traffic.json"
exit $rc
