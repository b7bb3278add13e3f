#!/bin/bash
# counter passes over the timing harness of the matrix-core top-2 kernel (issue and wait breakdown)
set -o pipefail
OUT=$PWD/gpurun_out/pmc_top2
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $REPO/tools/bin/exp_top2 > "$OUT/log$i.txt" 2>&1 || { tail -5 "$OUT/log$i.txt"; exit 1; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'top2' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
