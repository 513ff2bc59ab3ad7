#!/bin/bash
# sweep the workgroup-count target of k_search_fast (run on the GPU box)
for wgs in 2048 4096 8192 16384 32768; do
  echo "RTDM_FAST_WGS=$wgs"
  RTDM_FAST_WGS=$wgs python bench.py --steps 8 --warmup 2 --batch 64 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_launch'])"
done
