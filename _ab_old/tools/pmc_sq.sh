# Developer helper: SQ counters of the bench kernels (one rocprofv3 --pmc pass; no trace domains with it)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT -o sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 > /dev/null 2>&1
ls $OUT
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("pk_cycle", "pk_xall", "pk_hess"):
        if k not in acc:
            continue
        print(f.split("/")[-1], k, {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
PY
