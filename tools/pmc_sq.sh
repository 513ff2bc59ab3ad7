#!/bin/bash
# SQ activity of the hot-path kernels (run on the GPU box): one rocprofv3 --pmc pass, no tracing domains.
# usage: tools/pmc_sq.sh [tag] [bench args...]   -> gpurun_out/pmc_sq_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
OUT=$R/gpurun_out/pmc_sq_$TAG
rm -rf $OUT; mkdir -p $OUT
B="--steps 2 --warmup 2 --batch ${PMC_BATCH:-256} --headline-only $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -- python $R/bench.py $B > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
    --output-format csv -d $OUT/sq2 -- python $R/bench.py $B > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void rtdm::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    ks = sorted({k for k, _ in agg})
    for k in ks:
        if "k_search" not in k and "k_lrcheck" not in k and "k_spk" not in k: continue
        c = {cn: sum(v) / len(v) for (kk, cn), v in agg.items() if kk == k}
        print("==", k[:70])
        print("  ", "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            print("   wave-cycle split: issuing %.3f  issue-stalled %.3f  parked %.3f" % (c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc))
            simd = c["GRBM_GUI_ACTIVE"] / 8 * 1024
            print("   VALU busy frac of SIMD cycles %.3f ; VALU insts/pixel (720p) %.1f ; clk-cycles %.4g" % (c["SQ_ACTIVE_INST_VALU"] * 4 / simd, c["SQ_INSTS_VALU"] * 64 / ('${PMC_BATCH:-256}' and int("${PMC_BATCH:-256}") * 1280 * 720), c["GRBM_GUI_ACTIVE"] / 8))
PY
