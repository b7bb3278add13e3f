#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/<name> into profiles/<name> (the summaries
# that are kept in the repository; the raw counter and trace directories stay behind) and rebuilds
# profiles/traffic.json and <name>/pmc_summary.txt.  Usage: bash tools/copy_profiles.sh r02
set -e
N=${1:-r02}
SRC=gpurun_out/$N
DST=profiles/$N
mkdir -p "$DST"
for f in bench_default.json bench_exclusive.json exp_fast.txt exp_octree.txt latency_breakdown.txt pcie_rate.txt \
         pmc_top2.txt probe_mfma_f4.txt top2_harness.txt ubench_mfma_fill.txt ubench_mfma_i8.txt ubench_valu.txt \
         ubench_ta.txt sfi_time.txt bow_rate.json config5_pairs.json distinctive_rate.json projection_rate.json \
         stereo_rate.json voc_rate.json blur_mfma_ab.txt bench_gloo_2ranks_no_launcher.json \
         bench_nccl_1rank_forced_dist.json test_sharded_rccl.txt probe_mfma_i8_16.txt; do
  [ -f "$SRC/$f" ] && cp "$SRC/$f" "$DST/$f"
done
# the newest kernel-trace summary of each command
for w in default exclusive; do
  f=$(ls -t "$SRC"/trace_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$DST/kernel_stats_$w.csv"
done
python3 tools/summarize_pmc.py "$SRC" "$DST" > /dev/null
echo "copied to $DST"
