#!/bin/bash
# Static instruction mix of one k_search_ring instantiation (no GPU needed): compiles csrc/k_search_ring.hip for gfx950 with
# -DRTDM_RING_DEV="X(D, w, LPP)" (RING_CFG, default the headline form "X(64, 9, 4)") and prints the mnemonic histogram of the non-fused kernel, per row
# group (TRIP / LPP groups are unrolled in the loop body).  Usage: tools/ring_isa.sh [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${RING_ISA_OUT:-/tmp/isa}
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S "-DRTDM_RING_DEV=${RING_CFG:-X(64, 9, 4)}" "$@" \
    $R/rt-depth-map_amd/csrc/k_search_ring.hip -o $OUT/ring_dev.s 2>/dev/null || exit 1
python3 - $OUT/ring_dev.s <<'PY'
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN4rtdm13k_search_ringILi\d+ELi\d+ELi\d+ELb0EEE", l)][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
h = collections.Counter(body)
valu = sum(c for k, c in h.items() if k.startswith("v_"))
groups = h["v_qsad_pk_u16_u8"] + h["v_mqsad_pk_u16_u8"]
import os
per = int(os.environ.get("RING_QSAD_PER_GROUP", "48"))     # quad-SADs per 64 pixel-rows: D/4 * ceil(w/4)
print("instructions %d, VALU %d, quad-SADs %d (=> %.1f x 64 pixel-rows), VALU per 64 pixel-rows %.1f" % (len(body), valu, groups, groups / per, valu / max(1, groups / per)))
for k, c in h.most_common(60):
    print("  %-28s %5d  %6.1f / 64 pixel-rows" % (k, c, c / max(1, groups / per)))
for l in lines:
    if "vgpr_count" in l or "vgpr_spill" in l or "sgpr_count" in l: print(l.strip())
PY
