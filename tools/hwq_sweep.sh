#!/bin/bash
# bench value with 4 / 8 / 16 hardware queues per process (HIP's default is 4) and 4 / 6 / 8 extractor instances
for q in 4 8 16; do
  for cfg in "4 1024" "8 1024"; do
    set -- $cfg
    v=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --no-pcie --no-config5 --steps 20 --instances $1 --batch $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])") || exit 1
    echo "queues $q instances $1 batch $2: $v"
  done
done
