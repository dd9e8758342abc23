#!/bin/bash
# usage: tools/prof_seq.sh <tag>   (GPU box): kernel stats of the sequential (dataflow) mode, C2 workload
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_event_check.py c2time > $OUT/trace.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:30]:
        print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete
