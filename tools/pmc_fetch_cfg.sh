#!/bin/bash
# FETCH_SIZE of the search kernels of one configuration under an environment switch (run on the GPU box):
#   tools/pmc_fetch_cfg.sh "W H D w batch" VAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; VAR=$2; shift; shift
B=$(echo $CFG | awk '{print $5}')
for V in "$@"; do
    export $VAR=$V
    OUT=$R/gpurun_out/pmc_fetch_cfg/$VAR-$V
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python $R/tools/run_config.py $CFG > $OUT/run.txt 2>/dev/null
    grep fast $OUT/run.txt
    python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "k_search" in k: print("$VAR=$V", k, "FETCH_SIZE x2 = %.2f MB per pair" % (2 * sum(v) / len(v) * 1024 / $B / 1e6))
PY
done
