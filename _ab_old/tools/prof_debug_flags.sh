cd /tmp && export TMPDIR=/tmp
for f in ${FLAGS:-0 1024 256 512 768}; do
  export POCKIT_AMD_DEBUG_FLAGS=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg_$f -o d -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/dbg_$f.json 2>/dev/null
  echo "flags=$f"; head -3 $GRAFT_REPO_ROOT/gpurun_out/dbg_$f/d_kernel_stats.csv | cut -c1-120
  python3 -c "
import json;d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/dbg_$f.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step'],d['roofline']['avg_launch_us'])"
done
