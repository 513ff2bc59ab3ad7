#!/bin/bash
# search-kernel time against the number of row strips (RTDM_FAST_WGS = strips * tiles * batch), batch 128
for s in 6 7 8 9 10 11 12 13 14 15 16 18 20 24; do
  RTDM_FAST_WGS=$((s*640)) timeout -k 10 200 python bench.py --no-cpu-baseline --batch 128 --steps 8 > gpurun_out/bb.log 2>&1
  python -c "
import json;d=json.loads(open('gpurun_out/bb.log').read().strip().splitlines()[-1]);rs=-(-711//$s);print('strips', $s, 'rs', rs, 'stride_B', rs*1280, 'mod64K', (rs*1280)%65536, 'search_ms', d['stage_ms_per_launch']['search'], 'pairs/s', d['value'])"
done
