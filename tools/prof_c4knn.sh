#!/bin/bash
# usage (on the GPU box via gpurun): tools/prof_c4knn.sh <tag> [shape]   -- kernel stats + HBM counters of the time-sliced mode's step kernel on
# configs[3]'s own graph (tools/run_scale_shapes.py c4_knn); counters in their own passes (FETCH_SIZE, WRITE_SIZE), as the guide prescribes
set -u
TAG=$1; SHAPE=${2:-c4_knn}
export NEEDLE="sl_direct_kernel<8, 16, true, true, 16>"   # (round 6: the node-line instantiation; <8, 16, true, true, 0> under AE_SL_NO_LINES)
case $SHAPE in c5*) export NEEDLE="sl_direct_kernel<16, 32, true, true, 32>";; c3*) export NEEDLE="sl_slice_kernel<2, 16, true, true>";; esac   # (c3_knn: merged slices)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/tools/run_step_shape.py $SHAPE 3 > $OUT/run_trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python3 $R/tools/run_step_shape.py $SHAPE 1 > $OUT/run_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python3 $R/tools/run_step_shape.py $SHAPE 1 > $OUT/run_write.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, json, os
out = {}
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('kernel_stats_top.txt', 'w') as w:
        for r in rows[:16]:
            line = "%-120s calls %8s total %12s ns avg %12s ns  %6s %%" % (r['Name'][:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
            print(line); w.write(line + "\n")
    for r in rows:
        if os.environ['NEEDLE'] in r['Name']:
            out['step_kernel'] = r['Name']; out['calls'] = int(r['Calls']); out['avg_ns'] = float(r['AverageNs'])
def avg(d, counter, needle):
    tot = 0.0; seen = set()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r['Kernel_Name'] and r['Counter_Name'] == counter:
                tot += float(r['Counter_Value']); seen.add(r['Dispatch_Id'])
    return (tot / len(seen), len(seen)) if seen else (None, 0)
fetch, nf = avg('pmc_fetch', 'FETCH_SIZE', os.environ['NEEDLE'])
write, nw = avg('pmc_write', 'WRITE_SIZE', os.environ['NEEDLE'])
out.update({"FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write, "dispatches_counted": [nf, nw], "fetch_correction": 2.0,
            "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024 if fetch and write else None,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/run_scale_shapes.py; gfx950 FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"})
json.dump(out, open('pmc_step_kernel.json', 'w'), indent=1)
print(out)
PY
python3 $R/tools/brief_shapes.py $OUT/run_trace.log | grep -v FULL
find . -name "*.db" -delete; find . -name "*kernel_trace.csv" -delete; find . -name "*counter_collection.csv" -size +1M -delete
du -sh .
