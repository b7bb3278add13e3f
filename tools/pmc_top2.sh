#!/bin/bash
# one counter pass over the Hamming bench (issue and wait breakdown of the matrix-core top-2 kernel)
set -o pipefail
OUT=$PWD/gpurun_out/pmc_top2
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT" -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'top2' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
