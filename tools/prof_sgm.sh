#!/bin/bash
# rocprofv3 evidence for the StereoSGBM kernels (run on the GPU box): per-kernel time (--kernel-trace --stats) and, in passes
# of their own, FETCH_SIZE / WRITE_SIZE.  usage: tools/prof_sgm.sh TAG [paths=8] [pairs per call=4]  -> gpurun_out/sgm_prof_TAG/{stats.csv,traffic.json,summary.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; PATHS=${2:-8}; N=${3:-4}; CALLS=3
OUT=$R/gpurun_out/sgm_prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/prof_sgm.py $PATHS $N $CALLS > /dev/null 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/prof_sgm.py $PATHS $N $CALLS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/prof_sgm.py $PATHS $N $CALLS > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, json, re
OUT, N, CALLS, PATHS = "$OUT", $N, $CALLS, $PATHS
W1, H, D = 1280 - 128, 720, 128
stats = glob.glob(OUT + "/stats/*/*kernel_stats.csv")
rows = []
if stats:
    rows = [r for r in csv.DictReader(open(stats[0])) if "rtdm" in r["Name"]]
    with open(OUT + "/stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
def load(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(OUT + "/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("rtdm::", "")].append(float(r["Counter_Value"]))
    return agg
fe, wr = load("fetch"), load("write")
out = {"workload": "1280x720 D=128 blockSize 5, %d paths, %d pairs per call" % (PATHS, N), "unit": "bytes per PAIR (FETCH_SIZE x2: gfx950 correction, KB -> bytes; WRITE_SIZE exact)",
       "algorithmic_bytes_per_pair": 4 * 1280 * 720, "volume_elements_per_pair": W1 * H * D, "kernels": {}}
tot = 0
lines = []
for k in sorted(set(fe) | set(wr)):
    if not k.startswith("k_sgm") and not k.startswith("k_spk"): continue
    f = sum(fe.get(k, [0])) * 2.0 * 1024 / (N * CALLS); w = sum(wr.get(k, [0])) * 1024 / (N * CALLS)
    ms = [float(r["TotalDurationNs"]) / 1e6 / (N * CALLS) for r in rows if re.sub(r"^void ", "", r["Name"].split("(")[0]).replace("rtdm::", "") == k]
    out["kernels"][k] = {"launches_per_call": len(fe.get(k, [])) // CALLS, "fetch_bytes_per_pair": int(f), "write_bytes_per_pair": int(w),
                         "ms_per_pair": round(sum(ms), 4) if ms else None,
                         "hbm_GBps": round((f + w) / 1e9 / (sum(ms) * 1e-3), 1) if ms and sum(ms) > 0 else None}
    tot += f + w
    lines.append("%-40s fetch %8.1f MB  write %8.1f MB  %s ms/pair" % (k[:40], f / 1e6, w / 1e6, out["kernels"][k]["ms_per_pair"]))
out["total_hbm_bytes_per_pair"] = int(tot)
out["total_ms_per_pair"] = round(sum(v["ms_per_pair"] or 0 for v in out["kernels"].values()), 4)
out["x_algorithmic"] = round(tot / (4 * 1280 * 720), 1)
json.dump(out, open(OUT + "/traffic.json", "w"), indent=1)
open(OUT + "/summary.txt", "w").write("\n".join(lines) + "\ntotal %.1f MB per pair = %.0f x algorithmic, %.4f ms per pair\n" % (tot / 1e6, out["x_algorithmic"], out["total_ms_per_pair"]))
print(open(OUT + "/summary.txt").read())
PY
