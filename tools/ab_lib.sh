#!/bin/bash
# A/B of several builds of the library on one box: bench value and exclusive stage times, alternating
# usage: tools/ab_lib.sh dir...   (each holds a liborbx.so; the in-tree one is restored afterwards)
cp orbslam2_amd/liborbx.so /tmp/liborbx_keep.so
for rep in 1 2 3; do
  for d in "$@"; do
    cp "$d/liborbx.so" orbslam2_amd/liborbx.so || exit 1
    v=$(timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --no-pcie --no-config5 --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['stages_ms_per_launch_exclusive'])") || exit 1
    echo "$d: $v"
  done
done
cp /tmp/liborbx_keep.so orbslam2_amd/liborbx.so
