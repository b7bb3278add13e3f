#!/bin/bash
# sweep of the workgroup target of the matrix-core top-2 kernel (tuning aid)
for b in 512 768 1024 2048; do
  ORBX_TOP2_BLOCKS=$b timeout -k 10 100 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print($b, d['hamming']['ms_per_call'], d['hamming']['self_check'])" || exit 1
done
