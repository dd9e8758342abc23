#!/bin/bash
# usage (GPU box): tools/prof_rank_share_pmc.sh <tag> <world>  -- HBM counters of the merged slice kernel on a rank's share of configs[3]'s graph (tools/run_shard_time.py);
#                  tools/prof_rank_share_pmc.sh <tag> c3         -- ... on configs[2]'s large graph, one device (tools/run_c3knn_sliced.py; 2 columns)
# FETCH_SIZE and WRITE_SIZE in their own passes, as the guide prescribes (gfx950: FETCH_SIZE doubled)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rankpmc_$1; mkdir -p $OUT
export NEEDLE="sl_slice_kernel<8, 16, true, true>"
PROG="$R/tools/run_shard_time.py $2"
if [ "$2" = "c3" ]; then export NEEDLE="sl_slice_kernel<2, 16, true"; PROG="$R/tools/run_c3knn_sliced.py 1650000 2"; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python3 $PROG > $OUT/run.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python3 $PROG > $OUT/run_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python3 $PROG > $OUT/run_write.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, json, os
out = {}
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ['NEEDLE'] in r['Name']:
            out['kernel'] = r['Name']; out['calls'] = int(r['Calls']); out['avg_ns'] = float(r['AverageNs'])
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
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/run_shard_time.py; gfx950 FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"})
json.dump(out, open('pmc_slice_kernel.json', 'w'), indent=1)
print(out)
PY
grep 'world\|sliced ms' run.log
find . -name "*.db" -delete; find . -name "*kernel_trace.csv" -delete; find . -name "*counter_collection.csv" -size +1M -delete
