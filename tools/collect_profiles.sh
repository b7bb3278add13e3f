#!/bin/bash
# Collects the evidence kept under profiles/: the default bench line, kernel-trace statistics of the
# default and of the exclusive (serialised, one instance) bench command, and separate counter passes
# (FETCH_SIZE, WRITE_SIZE, instruction counts) of the exclusive command plus the counter calibration.
# Usage (on the GPU box, from the repository root): bash tools/collect_profiles.sh <out-dir-under-gpurun_out>
set -o pipefail
OUT=$PWD/gpurun_out/${1:-r01}
rm -rf "$OUT"
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp
# the serialised runs keep 256 frames per launch, the unit of the tables under profiles/
EXCL="$REPO/bench.py --exclusive --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --no-hamming --no-pcie --no-config5 --frames-resident 256"
timeout -k 10 300 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
timeout -k 10 200 python3 bench.py --exclusive --batch 256 --no-cpu-baseline --no-pcie --no-config5 > "$OUT/bench_exclusive.json" 2>> "$OUT/bench_default.err" || exit 1
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 $REPO/bench.py --no-cpu-baseline --no-pcie --no-config5 > "$OUT/trace_default.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_exclusive" -- python3 $REPO/bench.py --exclusive --batch 256 --no-cpu-baseline --no-pcie --no-config5 > "$OUT/trace_exclusive.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $EXCL > "$OUT/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $EXCL > "$OUT/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python3 $EXCL > "$OUT/pmc_sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_l1" -- python3 $EXCL > "$OUT/pmc_l1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_lds" -- python3 $EXCL > "$OUT/pmc_lds.log" 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_fetch" -- $REPO/tools/ubench_fetch > "$OUT/cal_fetch.log" 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_write" -- $REPO/tools/ubench_fetch > "$OUT/cal_write.log" 2>&1 || exit 1
cd $REPO
timeout -k 10 60 tools/ubench_valu > "$OUT/ubench_valu.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/ubench_ta > "$OUT/ubench_ta.txt" 2>/dev/null || exit 1
timeout -k 10 100 tools/bin/exp_fast 256 20 > "$OUT/exp_fast.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/ubench_mfma_i8 > "$OUT/ubench_mfma_i8.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/probe_mfma_i8_16 > "$OUT/probe_mfma_i8_16.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/probe_mfma_f4 > "$OUT/probe_mfma_f4.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/ubench_mfma_fill > "$OUT/ubench_mfma_fill.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/bin/exp_top2 > "$OUT/top2_harness.txt" 2>/dev/null || exit 1
bash tools/pmc_top2.sh > "$OUT/pmc_top2.txt" 2>&1 || echo "pmc_top2 failed" >> "$OUT/pmc_top2.txt"
timeout -k 10 200 python3 tools/pcie_rate.py > "$OUT/pcie_rate.txt" 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/latency_breakdown.py > "$OUT/latency_breakdown.txt" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/config5_pairs.py > "$OUT/config5_pairs.json" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/stereo_rate.py > "$OUT/stereo_rate.json" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/distinctive_rate.py > "$OUT/distinctive_rate.json" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/bow_rate.py > "$OUT/bow_rate.json" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/voc_rate.py > "$OUT/voc_rate.json" 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/projection_rate.py > "$OUT/projection_rate.json" 2>/dev/null || exit 1
timeout -k 10 100 python3 tools/sfi_time.py > "$OUT/sfi_time.txt" 2>/dev/null || exit 1
timeout -k 10 60 tools/bin/exp_octree > "$OUT/exp_octree.txt" 2>/dev/null || exit 1
# the matrix-core form of the blur beside the default (vector-ALU) form, alternating
for rep in 1 2; do for mode in valu mfma; do
  ORBX_BLUR=$mode timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-hamming --no-pcie --no-config5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$mode', d['value'], d['stages_ms_per_launch_exclusive']['blur'])" >> "$OUT/blur_mfma_ab.txt" || exit 1
done; done
# the launch of N ranks by bench.py itself (gloo, both ranks on this GPU) and the C sharded matcher with one RCCL rank
ORBX_BENCH_BACKEND=gloo ORBX_BENCH_SHARE_GPU=1 timeout -k 10 280 python3 bench.py --gpus 2 --no-pcie --cpu-seconds 3 > "$OUT/bench_gloo_2ranks_no_launcher.json" 2>/dev/null || exit 1
ORBX_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-pcie --no-cpu-baseline --no-config5 > "$OUT/bench_nccl_1rank_forced_dist.json" 2>/dev/null || exit 1
timeout -k 10 100 tests/cpp/test_sharded_rccl 1 777 50000 > "$OUT/test_sharded_rccl.txt" 2>&1 || exit 1
echo collected
