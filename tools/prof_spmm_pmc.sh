#!/bin/bash
# usage (GPU box): tools/prof_spmm_pmc.sh <tag> [lattice|knn]  -- HBM bytes fetched per launch of the sparse product of the SVD initialisation at the
# configs[3] size (FETCH_SIZE in its own pass, doubled for gfx950 as the guide prescribes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spmm_pmc_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o pmc -- python3 $R/tools/run_svd_init_c4.py ${2:-lattice} > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, json
tot = 0.0; seen = set()
for f in glob.glob('pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_csr_vec4_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            tot += float(r['Counter_Value']); seen.add(r['Dispatch_Id'])
out = {"kernel": "spmm_csr_vec4_kernel", "dispatches": len(seen), "FETCH_SIZE_KiB_per_launch": tot / max(1, len(seen)), "fetch_correction": 2.0,
       "hbm_fetch_bytes_per_launch": 2.0 * 1024 * tot / max(1, len(seen))}
json.dump(out, open('spmm_pmc.json', 'w'), indent=1); print(out)
PY
tail -1 run.log | cut -c1-200
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
