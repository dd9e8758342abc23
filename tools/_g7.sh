cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4 5; do
  AE_DEBUG_KNOBS=1 AE_MFMA_STREAM_VARIANT=$v rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/sv$v -o t -- python3 $GRAFT_REPO_ROOT/tools/run_svd_dense.py > /tmp/sv$v.log 2>&1
  f=$(find /tmp/sv$v -name "t_kernel_stats.csv" | head -1)
  echo "variant $v: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'stream_kernel' in r['Name']: print(r['Calls'], round(float(r['AverageNs'])/1e3,2),'us')
")"
done
