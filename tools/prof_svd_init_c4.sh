#!/bin/bash
# usage (GPU box): tools/prof_svd_init_c4.sh <tag> [lattice|knn]  -- kernel stats of the SVD initialisation at the configs[3] size
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/svdinit_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_svd_init_c4.py ${2:-lattice} > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY' > kernel_stats_top.txt
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:18]:
        print("%-110s calls %6s total_ms %10.3f avg_us %10.2f  %5s %%" % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
PY
cat kernel_stats_top.txt; tail -2 run.log
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete
