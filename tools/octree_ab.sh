# batch throughput and exclusive distribution-stage time with the 256- and the 512-thread form of k_octree (ORBX_OCTREE_THREADS)
for rep in 1 2; do
for nt in 256 512; do
  v=$(ORBX_OCTREE_THREADS=$nt timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['stages_ms_per_launch_exclusive']['octree'])") || exit 1
  echo "octree threads $nt: $v"
done
done
