#!/bin/bash
# One 720p frame per call under rocprofv3 --kernel-trace: per-kernel device time and the gaps between the kernels of a call
# (run on the GPU box) -> gpurun_out/single_trace/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/single_trace
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/d -- python3 $R/tools/prof_single.py > $OUT/run.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
f = glob.glob("$OUT/d/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rtdm" in r["Kernel_Name"]]
# a call starts with k_fill_frame, or -- since the fill rides in the prefilter's grid -- with the prefilter
calls, cur = [], []
for r in rows:
    first = "k_fill_frame" in r["Kernel_Name"] or ("k_prefilter" in r["Kernel_Name"] and not (cur and "k_fill_frame" in cur[-1]["Kernel_Name"]))
    if first and cur: calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
calls = calls[5:]                     # warm
agg = collections.OrderedDict(); gaps = []; spans = []
for c in calls:
    spans.append((int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / 1e3)
    for i, r in enumerate(c):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rtdm::", "")
        agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if i: gaps.append((int(r["Start_Timestamp"]) - int(c[i - 1]["End_Timestamp"])) / 1e3)
print("calls %d; first kernel start -> last kernel end: %.1f us avg; sum of kernel times %.1f us; sum of gaps %.1f us (%d gaps/call)" % (
    len(calls), sum(spans) / len(spans), sum(sum(v) for v in agg.values()) / len(calls), sum(gaps) / len(calls), len(gaps) // len(calls)))
for n, v in agg.items(): print("  %-60s %7.1f us x %.1f per call" % (n[:60], sum(v) / len(v), len(v) / len(calls)))
PY
