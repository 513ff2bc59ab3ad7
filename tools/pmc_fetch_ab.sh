#!/bin/bash
# FETCH_SIZE of the search kernel under an environment switch (run on the GPU box): tools/pmc_fetch_ab.sh VAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VAR=$1; shift
for V in "$@"; do
    export $VAR=$V
    OUT=$R/gpurun_out/pmc_fetch_ab/$VAR-$V
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python $R/bench.py --steps 2 --warmup 2 --batch 256 --no-cpu-baseline --no-single-frame > /dev/null 2>&1
    python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "k_search" in k: print("$VAR=$V", k, "FETCH_SIZE x2 = %.2f MB per pair" % (2 * sum(v) / len(v) * 1024 / 256 / 1e6))
PY
done
