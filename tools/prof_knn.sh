#!/bin/bash
# MFMA-pipe occupancy and LDS stalls of the kNN candidate kernel (GPU box, via gpurun)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/knnpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $OUT/a -o p -- python3 $R/tools/run_knn.py > $OUT/run.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT --kernel-trace -d $OUT/b -o p -- python3 $R/tools/run_knn.py > $OUT/run2.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'knn_candidates' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k,v in agg.items():
    print(k, {c: round(x/len(cnt[(k,c)])) for c,x in v.items()})
PY
tail -2 $OUT/run.log; tail -2 $OUT/run2.log
find $OUT -name "*.csv" -size +2M -delete
