cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "any_dim or hub or k6_blobs or c3_schedule" 2>&1 | tail -6
(timeout 1500 python bench.py --steps 20 --warmup 3 > gpurun_out/r3/bench_r3_v2.json 2> gpurun_out/r3/bench_r3_v2.err; echo rc=$?)
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r3/bench_r3_v2.json').read().strip().splitlines()[-1])
print({k: j[k] for k in ('value', 'ms_per_step', 'scaling')}, 'frac', j['roofline']['frac'], j['config']['ce_mode'])
print('cpu', j['cpu_baseline'])
print('fidelity', {k: v for k, v in j['fidelity'].items() if isinstance(v, dict)})
print('exact', j['exact_mode']['ms_per_step'], 'event', j['event_mode']['ms_per_step'], 'rounds', j['rounds_mode']['ms_per_step'])
print('svd_init', j['svd_init'], 'svd_dense', j['svd_dense'])
for k, v in j['scale_shapes'].items():
    print(k, v.get('graph')[:80], 'default->', v['default_mode_resolves_to'][:30])
    for m in ('rounds_mode', 'sliced_mode', 'exact_mode'):
        if m in v: print('   ', m, round(v[m]['ms_per_step'], 2), 'ms frac', round(v[m]['roofline']['frac_whole_batch'], 4), 'ce', v[m]['ce_after'], v[m]['roofline'].get('sliced'))
PY
