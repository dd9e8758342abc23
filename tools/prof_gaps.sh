#!/bin/bash
# usage: tools/prof_gaps.sh <tag> <n> <k> <d>   (GPU box): per-dispatch timeline of AE_CE_SLICED (durations and gaps between consecutive launches) + SQ counters
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VARIANTS=match
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o t -- python3 $R/tools/run_match_check.py scale "$@" 2 > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/pmc_sq -o p -- python3 $R/tools/run_match_check.py scale "$@" 2 > $OUT/pmc_sq.log 2>&1
cd $OUT
python3 - <<'PY' > summary.txt
import csv, glob, collections
for f in glob.glob('trace/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the last batch: from the last sl_count_kernel on
    idx = [i for i, r in enumerate(rows) if 'sl_count_kernel' in r['Kernel_Name']]
    seg = rows[idx[-1]:]
    span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
    print('last batch: %d dispatches, span %.2f ms, sum of kernel durations %.2f ms' % (len(seg), span, busy))
    d = collections.defaultdict(list); g = collections.defaultdict(list)
    for a, b in zip(seg[:-1], seg[1:]):
        d[a['Kernel_Name'][:60]].append((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3)
        g[b['Kernel_Name'][:60]].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
    for k in d:
        gg = g.get(k, [0])
        print('%-62s n %5d  dur avg %8.2f us  (min %.2f max %.2f)  gap before avg %.2f us' % (k, len(d[k]), sum(d[k]) / len(d[k]), min(d[k]), max(d[k]), sum(gg) / len(gg)))
    # direct kernel: duration vs grid size
    ds = [(int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in seg if 'sl_direct' in r['Kernel_Name']]
    ds.sort()
    for q in (0, len(ds) // 4, len(ds) // 2, 3 * len(ds) // 4, len(ds) - 1):
        print('direct kernel grid threads %d -> %.2f us' % ds[q])
for f in glob.glob('pmc_sq/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:50]
        if 'sl_direct' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    for k, v in agg.items():
        print('pmc', k, 'dispatches', len(seen[k]), {c: round(x / len(seen[k])) for c, x in v.items()})
PY
cat summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
