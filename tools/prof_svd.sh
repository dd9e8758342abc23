#!/bin/bash
# usage: tools/prof_svd.sh <tag> -- kernel trace of the SVD init (bench without CE steps)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/svdprof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python3 $R/${SVDSCRIPT:-tools/run_svd_init.py} > $OUT/run.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:25]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9),"us", ("%.2f"%(float(r['TotalDurationNs'])/1e6)).rjust(8),"ms")
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
tail -3 $OUT/run.log
