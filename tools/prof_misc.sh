#!/bin/bash
# rocprofv3 evidence for the kernels either side of the matcher (run on the GPU box): per-kernel time and, in passes of their
# own, FETCH_SIZE / WRITE_SIZE.  -> gpurun_out/misc_prof/{stats.csv,traffic.json,summary.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CALLS=3
OUT=$R/gpurun_out/misc_prof
rm -rf $OUT; mkdir -p $OUT
for PART in filters objects; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$PART/stats -- python3 $R/tools/prof_misc.py $PART $CALLS > /dev/null 2> $OUT/$PART.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$PART/fetch -- python3 $R/tools/prof_misc.py $PART $CALLS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$PART/write -- python3 $R/tools/prof_misc.py $PART $CALLS > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re
OUT = "$OUT"
def short(n): return re.sub(r"^void ", "", n.split("(")[0]).replace("rtdm::", "")
rows = []
for part in ("filters", "objects"):
    stats = glob.glob(OUT + "/" + part + "/stats/*/*kernel_stats.csv")
    for r in (csv.DictReader(open(stats[0])) if stats else []):
        if "rtdm" in r["Name"]:
            r = dict(r); r["Name"] = part + ": " + r["Name"]; rows.append(r)
if rows:
    with open(OUT + "/stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
def load(d):
    agg = collections.defaultdict(list)
    for part in ("filters", "objects"):
        for f in glob.glob(OUT + "/" + part + "/" + d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)): agg[part + ": " + short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg
short0 = short
short = lambda n: n.split(": ")[0] + ": " + short0(n.split(": ", 1)[1]) if ": " in n else short0(n)
fe, wr = load("fetch"), load("write")
# algorithmic bytes per LAUNCH: morph 2 B/px x 64 frames of 1280x720; rectify 2 cameras x (3 + 6 + 1) B per crop pixel x 64 pairs
algo = {"filters: k_morph_open_close": 2 * 1280 * 720 * 64, "filters: k_rectify_gray": 2 * 934 * 404 * 10 * 64,
        "objects: k_hsv_inrange": 4 * 934 * 404, "objects: k_morph_open_close": 2 * 934 * 404}
out = {"unit": "bytes per LAUNCH (FETCH_SIZE x2: gfx950 correction, KB -> bytes; WRITE_SIZE exact); avg_us from --kernel-trace --stats", "kernels": {}}
lines = []
for k in sorted(set(fe) | set(wr)):
    if not re.search(r": k_(morph|rectify|hsv|cc)", k): continue
    f = sum(fe.get(k, [0])) / max(1, len(fe.get(k, [1]))) * 2.0 * 1024; w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [1]))) * 1024
    avg = [float(r["AverageNs"]) / 1e3 for r in rows if short(r["Name"]) == k]
    a = next((v for kk, v in algo.items() if k.startswith(kk)), None)
    e = {"launches": len(fe.get(k, [])), "fetch_bytes": int(f), "write_bytes": int(w), "avg_us": round(avg[0], 2) if avg else None,
         "hbm_GBps": round((f + w) / 1e9 / (avg[0] * 1e-6), 1) if avg and avg[0] > 0 else None, "algorithmic_bytes": a,
         "x_algorithmic": round((f + w) / a, 2) if a else None}
    out["kernels"][k] = e
    lines.append("%-36s avg %9.2f us  fetch %9.2f MB  write %9.2f MB  %s GB/s  x algorithmic %s" % (k[:36], e["avg_us"] or 0, f / 1e6, w / 1e6, e["hbm_GBps"], e["x_algorithmic"]))
json.dump(out, open(OUT + "/traffic.json", "w"), indent=1)
open(OUT + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
