#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/pmc_bm
rm -rf $OUT; mkdir -p $OUT
REPO=$PWD
export TMPDIR=/tmp
EXCL="$REPO/bench.py --exclusive --batch 256 --steps 10 --warmup 2 --no-cpu-baseline --no-hamming --no-pcie --no-config5 --frames-resident 256"
cd /tmp
ORBX_BLUR=mfma timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 $EXCL > "$OUT/sq.log" 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/sq/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in acc:
    if 'blur' in k or 'fast' in k:
        print(k, n[k], {c: round(v/n[k]) for c,v in acc[k].items()})
PY
