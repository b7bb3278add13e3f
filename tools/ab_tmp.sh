#!/bin/bash
timeout -k 10 300 python -u -m pytest tests/test_gpu_extractor.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 2
for rep in 1 2; do
for cfg in "--no-pcie" "" "--no-pipeline --no-pcie" "--no-pipeline" "--hw-queues 16 --no-pcie" "--hw-queues 6 --no-pcie" "--hw-queues 5 --no-pcie" "--instances 6 --batch 1536 --no-pcie" "--instances 3 --batch 768 --no-pcie" "--instances 5 --batch 1280 --no-pcie" "--instances 8 --batch 2048 --hw-queues 16 --no-pcie"; do
  v=$(timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --no-config5 --steps 20 --frames-resident 4096 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])") || { echo "failed $cfg"; exit 1; }
  echo "'$cfg': $v"
done
done
