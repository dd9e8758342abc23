#!/bin/bash
# usage: tools/trace_bench.sh <tag> [bench args]  -- kernel trace + stats only
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:22]:
    print(r['Name'][:75].ljust(75), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9),"us", ("%.2f"%(float(r['TotalDurationNs'])/1e6)).rjust(8),"ms")
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
grep -o '"svd_init[^}]*}' $OUT/bench.log
