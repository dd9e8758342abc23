#!/bin/bash
# usage: tools/prof_svd_dense.sh <tag>   (GPU box): kernel stats + MFMA-pipe counters of the dense direct_svd (60000 x 784, rank 20)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/svdprof_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_svd_dense.py > $OUT/run.log 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc -o p -- python3 $R/tools/run_svd_dense.py > $OUT/pmc.log 2>&1
cd $OUT
python3 - <<'PY' > summary.txt
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:90], r['Calls'], 'total_ms', round(float(r['TotalDurationNs']) / 1e6, 3), 'avg_us', round(float(r['AverageNs']) / 1e3, 2))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob('pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if 'mfma' in k or 'gram' in k:
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k, v in agg.items():
    n = len(cnt[k]); print('pmc', k, 'dispatches', n, {c: round(x / n) for c, x in v.items()})
PY
cat summary.txt; tail -1 run.log
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
