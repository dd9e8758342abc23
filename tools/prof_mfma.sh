#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mfmaprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --list-avail 2>/dev/null | grep -i "Counter_Name" | grep -i "MFMA" | head -20
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT -o p -- python3 $R/tools/run_svd_dense.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        if 'mfma' in k or 'gram' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    n=len(cnt[k]); print(k, "dispatches", n, {c: round(x/n) for c,x in v.items()})
PY
tail -2 $OUT/run.log
find $OUT -name "*.csv" -size +2M -delete
