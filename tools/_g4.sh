cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_configs.py -m gpu -q -k hub_stress -s 2>&1 | grep -v amdgpu.ids | tail -15
(timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/r3/bench_r3_v1.json 2> gpurun_out/r3/bench_r3_v1.err; echo rc=$?)
tail -c 3000 gpurun_out/r3/bench_r3_v1.err | tail -15
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r3/bench_r3_v1.json').read().strip().splitlines()[-1])
    print({k: j[k] for k in ('value', 'ms_per_step', 'scaling')}, j['roofline']['frac'], j['cpu_baseline'])
    print('fidelity', j['fidelity'])
    for k, v in j['scale_shapes'].items():
        print(k, v.get('graph'), 'default->', v['default_mode_resolves_to'])
        for m in ('rounds_mode', 'sliced_mode', 'exact_mode'):
            if m in v: print('   ', m, round(v[m]['ms_per_step'], 2), 'ms frac', round(v[m]['roofline']['frac_whole_batch'], 4), 'ce', v[m]['ce_after'], v[m]['roofline'].get('sliced'))
except Exception as e:
    print('parse failed', e)
PY
