#!/bin/bash
# HBM traffic of the hot-path kernels (run on the GPU box; separate --pmc passes, no tracing domains).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $GRAFT_REPO_ROOT/tools/ubench_fetch > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- $GRAFT_REPO_ROOT/tools/ubench_fetch > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections
def load(d):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
for d in ("cal_fetch", "cal_write", "bench_fetch", "bench_write"):
    print("==", d)
    for (k, c), v in sorted(load(d).items()):
        if "rtdm" in k or "k_" in k:
            print("%-50s %-11s n=%3d mean=%14.1f" % (k, c, len(v), sum(v) / len(v)))
PY
