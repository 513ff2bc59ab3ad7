#!/bin/bash
# HBM traffic and SQ activity of the hot-path kernels (run on the GPU box).  Counters are collected in passes of
# their own (no tracing domains): FETCH_SIZE and WRITE_SIZE cannot share a pass; the SQ set fills its 8 slots.
# Writes gpurun_out/pmc_traffic/{summary.txt,traffic.json}; tools/pmc_finalize.py copies them to profiles/rNN_pmc_traffic*
# and adds the git commit.  traffic.json records the hash of the device sources it was measured on (bench.py compares).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
B="--steps 2 --warmup 2 --batch 1024 --headline-only"    # the bench's own batch; warm-up 2: the second call measures the strip count
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $R/tools/ubench_fetch > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- $R/tools/ubench_fetch > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python $R/bench.py $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python $R/bench.py $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/bench_sq -- python $R/bench.py $B > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json, re, sys
sys.path.insert(0, "$R")
import bench as _bench
def load(d):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
lines = []
data = {}
for d in ("cal_fetch", "cal_write", "bench_fetch", "bench_write", "bench_sq"):
    lines.append("== " + d)
    for (k, c), v in sorted(load(d).items()):
        if "rtdm" in k or "k_" in k:
            lines.append("%-60s %-20s n=%3d mean=%16.1f" % (k[-60:], c, len(v), sum(v) / len(v)))
            if d.startswith("bench"):
                key = re.sub(r"^void ", "", k).replace("rtdm::", "").replace(", ", ",")
                data.setdefault(key, {})[c] = sum(v) / len(v)
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
PAIRS = 1024
out = {"kernel_source_sha": _bench.kernel_source_sha(), "source": "tools/pmc_traffic.sh (rocprofv3 --pmc, separate passes)", "pairs_per_launch": PAIRS, "workload": "1280x720 d=64 9x9",
       "fetch_correction": 2.0, "unit": "KB (FETCH_SIZE reads 1/2 on gfx950, calibrated with tools/ubench_fetch; WRITE_SIZE exact)",
       "sq_unit": "quad-cycles summed over all waves / SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs", "kernels": {}}
for k, c in sorted(data.items()):
    e = dict(c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_pair"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / PAIRS)
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        # SIMD-cycles available = clocks x 256 CUs x 4 SIMDs; a VALU instruction occupies its SIMD for the counted quad-cycles x 4
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4
        # SQ_ACTIVE_INST_VALU counts quad-cycles.  The RAW ratio is reported, not clamped: it is an upper bound -- the counter books 4
        # cycles for every VALU instruction, while the cheapest ones (v_add/sub/and/xor/mov) were timed at 2.6 SIMD cycles
        # (profiles/r02_ubench_valu_instruction_costs.txt), so a kernel full of those can read above 1 (k_lrcheck_vec: 1.13);
        # short dispatches also read GRBM_GUI_ACTIVE high (MI355X_MICROARCH.md, DVFS give-back)
        e["valu_busy_raw"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 4)
        e["valu_busy_note"] = "raw SQ_ACTIVE_INST_VALU x 4 / SIMD cycles: an upper bound (4 cycles are booked for instructions that issue in 2.6); not clamped"
        e["valu_insts_per_pixel"] = round(c["SQ_INSTS_VALU"] * 64 / (PAIRS * 1280 * 720), 1) if "SQ_INSTS_VALU" in c else None
        if c.get("SQ_WAVE_CYCLES"):
            e["wave_cycle_split"] = {n: round(c.get(n, 0) / c["SQ_WAVE_CYCLES"], 4) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")}
    out["kernels"][k] = e
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print("\n".join(lines[-40:]))
PY
