#!/bin/bash
# usage: tools/prof_slice_passes.sh <tag> [n k d steps]   (GPU box): per-dispatch durations of the time-sliced mode's kernels, in launch order
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WITH_SEQ=0 ONLY_SLICED=1
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o t -- python3 $R/tools/run_slice_scale.py "$@" > $OUT/trace.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sl_' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)))
rows.sort()
# last 60 dispatches of the run: one line each
t_prev = None
for s, e, name, gx in rows[-48:]:
    gap = (s - t_prev) / 1e3 if t_prev else 0.0
    print("%-40s dur %8.1f us  gap before %6.1f us  grid_x %d" % (name, (e - s) / 1e3, gap, gx))
    t_prev = e
# totals
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for s, e, name, gx in rows:
    agg[name][0] += 1; agg[name][1] += (e - s) / 1e3; busy += (e - s) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
print("span %.1f ms, kernels busy %.1f ms" % (span / 1e3, busy / 1e3))
for k, (c, t) in agg.items():
    print(k, c, "%.1f ms" % (t / 1e3))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete
