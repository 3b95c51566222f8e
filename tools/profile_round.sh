#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the driver's bench command (headline workload only), summary copied to
# gpurun_out/<tag>_kernel_stats.csv; the bench line of the profiled run to gpurun_out/<tag>_bench_under_rocprofv3.json.
# usage: tools/profile_round.sh <tag> [bench args...]
tag=${1:-r02}; shift
repo=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_$tag" -o p -- \
  python3 "$repo/bench.py" --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-end-to-end "$@" \
  > "$repo/gpurun_out/${tag}_bench_under_rocprofv3.json" 2> "$repo/gpurun_out/${tag}_rocprof.err"
rc=$?
f=$(find "$repo/gpurun_out/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$repo/gpurun_out/${tag}_kernel_stats.csv" && head -5 "$f" | cut -c1-200
# the trace itself is large: keep only the stats
find "$repo/gpurun_out/prof_$tag" -name '*kernel_trace.csv' -delete
exit $rc
