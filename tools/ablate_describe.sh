for v in 1 2 3 0; do
  ORBX_DEBUG_DESCRIBE_STOP=$v timeout -k 10 100 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hamming 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stop=$v', d['stages_ms_per_batch'])" || exit 1
done
