#!/bin/bash
# Quick iteration on the GPU box: extractor parity tests, then the exclusive and the default bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_extractor.py -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
for mode in "--exclusive" ""; do
  timeout -k 10 200 python3 bench.py $mode --no-cpu-baseline --no-hamming 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('$mode', d['value'], 'fps', d['ms_per_step'], 'ms/step', d['stages_ms_per_launch_exclusive'], d['frames_per_launch'])" || exit 1
done
