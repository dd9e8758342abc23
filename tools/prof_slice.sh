#!/bin/bash
# usage: tools/prof_slice.sh <tag> [n k d steps]   (GPU box): kernel stats + PMC passes of the time-sliced mode at a scale shape
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WITH_SEQ=0
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_slice_scale.py "$@" > $OUT/trace.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/pmc_sq -o p -- python3 $R/tools/run_slice_scale.py "$@" > $OUT/pmc_sq.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- python3 $R/tools/run_slice_scale.py "$@" > $OUT/pmc_fetch.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o p -- python3 $R/tools/run_slice_scale.py "$@" > $OUT/pmc_write.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for d in ['pmc_sq', 'pmc_fetch', 'pmc_write']:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:50]
            if 'sl_' not in k and 'ce_round' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            print(d, k, 'dispatches', len(seen[k]), {c: round(x / len(seen[k])) for c, x in v.items()})
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
