#!/bin/bash
# Per-kernel times of the headline workload under rocprofv3 (run on the GPU box): tools/kstats.sh NAME [env assignments...]
# -> gpurun_out/kstats/NAME.csv (kernel stats) and a short table on stdout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=${1:-default}; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/kstats
mkdir -p $OUT; rm -rf $OUT/$NAME.d
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME.d -- python3 $R/bench.py --headline-only --steps ${KSTEPS:-10} --no-cpu-baseline > $OUT/$NAME.json 2> $OUT/$NAME.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/$NAME.d/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
open("$OUT/$NAME.csv", "w").write(open(f[0]).read() if f else "")
for r in rows[:12]:
    print("%-12s %-62s calls %4s avg %10.1f us  %5s %%" % ("$NAME", r["Name"].replace("rtdm::", "").replace("void ", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
