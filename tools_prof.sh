#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   (runs on the GPU box via gpurun)
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $OUT/pmc_sq -o pmc -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o pmc -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench_pmc_write.log 2>&1
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
PY
# keep only small summaries
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT
