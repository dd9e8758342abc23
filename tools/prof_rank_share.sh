#!/bin/bash
# usage (GPU box): tools/prof_rank_share.sh <tag> <n> <world>   -- kernel stats of a rank's share (tools/run_rank_share.py); env knobs pass through
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rank_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/run_rank_share.py $2 $3 3 > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY' > kernel_stats_top.txt
import csv, glob
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-100s calls %7s total_ms %10.3f avg_us %10.2f  %5s %%" % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
PY
cat kernel_stats_top.txt; grep RESULT run.log
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete
