#!/bin/bash
# usage (on the GPU box via gpurun): tools/prof_evgen.sh <tag>   -- kernel stats of a configs[3] batch's event generation (bucketed at generation / sorted)
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/tools/run_step_shape.py c4_knn 3 > $OUT/run_trace.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('kernel_stats_top.txt', 'w') as w:
        for r in rows[:24]:
            line = "%-110s calls %8s total %12s ns avg %12s ns  %6s %%" % (r['Name'][:110], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
            print(line); w.write(line + "\n")
PY
