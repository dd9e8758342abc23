#!/bin/bash
# usage: tools/prof_match.sh <tag> <variant> [n k d steps]   (GPU box): kernel stats + FETCH/WRITE PMC passes of AE_CE_SLICED (matchings) at a scale shape
set -u
TAG=$1; shift
export VARIANTS="$1"; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_match_check.py scale "$@" > $OUT/trace.log 2>&1
if [ -z "${NO_PMC:-}" ]; then
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- python3 $R/tools/run_match_check.py scale "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o p -- python3 $R/tools/run_match_check.py scale "$@" > $OUT/pmc_write.log 2>&1
fi
cd $OUT
python3 - <<'PY' > summary.txt
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:80], r['Calls'], 'total_ms', round(float(r['TotalDurationNs']) / 1e6, 2), 'avg_us', round(float(r['AverageNs']) / 1e3, 2))
for d in ['pmc_fetch', 'pmc_write']:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if 'sl_' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            print(d, k, 'dispatches', len(seen[k]), {c: round(x / len(seen[k])) for c, x in v.items()})
PY
cat summary.txt
grep "ms/step" trace.log
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
