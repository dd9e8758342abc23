#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/e2eprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python3 $R/tools/run_e2e.py > $OUT/run.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
rows=[r for r in rows if "at::native" not in r['Name'] and "Cijk" not in r['Name']]
for r in rows[:30]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9),"us", ("%.2f"%(float(r['TotalDurationNs'])/1e6)).rjust(8),"ms")
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
grep "embed()" $OUT/run.log
