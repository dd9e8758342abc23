#!/bin/bash
# usage (on the GPU box via gpurun): tools/prof_shapes.sh <tag> <shapes> [steps]   -- rocprofv3 kernel stats of tools/run_scale_shapes.py
set -u
TAG=$1; SHAPES=$2; STEPS=${3:-3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/tools/run_scale_shapes.py $SHAPES $STEPS > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    print("==", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-110s calls %8s total %12s ns avg %10s ns  %5s %%" % (r['Name'][:110], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']))
PY
python3 $R/tools/brief_shapes.py $OUT/run.log
# keep only the summaries (the raw traces are hundreds of MB)
find trace -name "*kernel_trace.csv" -delete
