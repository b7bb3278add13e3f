#!/bin/bash
# bench value for several numbers of extractor instances and frames per step (run on the GPU box from the repository root)
for cfg in "4 512" "6 768" "8 1024" "4 1024" "2 512" "3 384" "6 384" "8 512"; do
  set -- $cfg
  v=$(timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --no-pcie --no-config5 --steps 20 --instances $1 --batch $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])") || exit 1
  echo "instances $1 batch $2: $v"
done
