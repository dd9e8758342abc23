#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   (runs on the GPU box via gpurun)
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py --no-cpu-baseline --no-full-size "$@" > $OUT/bench_trace.log 2>&1
# counters: their own passes, the C2 workload only (the scale shapes instantiate the same kernel templates)
PMCARGS="--no-cpu-baseline --no-scale-shapes --no-fidelity --no-dense-svd --no-full-size"
# kernel stats of the C2 workload alone: every dispatch of the headline kernel is a warm-up or a timed batch of the bench line (in
# the full run above its average also holds the fidelity schedule's 25 batches from the dmap start, which are deeper chains)
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_c2 -o trace -- python3 $R/bench.py $PMCARGS "$@" > $OUT/bench_trace_c2.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/pmc_sq -o pmc -- python3 $R/bench.py $PMCARGS "$@" > $OUT/bench_pmc_sq.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python3 $R/bench.py $PMCARGS "$@" > $OUT/bench_pmc_fetch.log 2>&1
AE_DEBUG_KNOBS=1 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o pmc -- python3 $R/bench.py $PMCARGS "$@" > $OUT/bench_pmc_write.log 2>&1
cd $OUT
ls -R . | head -50
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    print("==", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r)
for d in ['pmc_sq','pmc_fetch','pmc_write']:
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen=set()
        for r in csv.DictReader(open(f)):
            key=(r['Kernel_Name'][:60], r['Dispatch_Id'])
            if key not in seen: seen.add(key); cnt[r['Kernel_Name'][:60]]+=1
        print("==", f)
        for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            if "anonymous namespace" in k or "ae::" in k:
                print(k, "dispatches", cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
# per-launch HBM traffic of the dominant kernel (guide: FETCH_SIZE / WRITE_SIZE in separate passes, KiB; gfx950 tallies
# 128-B read requests at 64 B -> FETCH_SIZE doubled)
import json
def avg(d, counter, needle):
    tot = 0.0; seen = set()
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r['Kernel_Name'] and r['Counter_Name'] == counter:
                tot += float(r['Counter_Value']); seen.add(r['Dispatch_Id'])
    return (tot / len(seen), len(seen)) if seen else (None, 0)
for needle, fname in (('ce_dataflow_kernel<2, false>', 'pmc_ce_dataflow.json'), ('ce_dataflow_kernel<2, true>', 'pmc_ce_ordered.json'),
                      ('ce_event_window_kernel', 'pmc_ce_event.json'), ('ce_round_node_kernel', 'pmc_ce_round.json')):
    fetch, nf = avg('pmc_fetch', 'FETCH_SIZE', needle)
    write, nw = avg('pmc_write', 'WRITE_SIZE', needle)
    if fetch is not None and write is not None:
        j = {"kernel": needle, "dispatches": nf, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
             "fetch_correction": 2.0, "hbm_bytes_per_launch": 2.0 * fetch * 1024 + write * 1024,
             "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --no-cpu-baseline --no-scale-shapes --no-fidelity --no-dense-svd` (C2 workload); gfx950 FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"}
        json.dump(j, open(fname, 'w'), indent=1)
        print(j)
PY
# keep only small summaries
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT
