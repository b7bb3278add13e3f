#!/bin/bash
# frames/s of the default workload by number of extractor instances and by the number of hardware queues
# the HIP runtime may use (tuning aid)
for q in 4 8; do
  for inst in 2 4 8; do
    v=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --instances $inst --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])") || exit 1
    echo "queues $q instances $inst: $v frames/s"
  done
done
