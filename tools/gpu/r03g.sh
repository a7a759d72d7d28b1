# round 3, call g: full GPU tier (incl. the 4-rank tests) + default bench line
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest_gpu.log 2>&1
tail -22 $O/pytest_gpu.log | cut -c1-220
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_default.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'roofline',d['roofline']['frac'],d['roofline'].get('frac_of_format_bytes'),'iter frac',d['roofline_iteration']['frac'])
print('solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')}); print('setup',c['setup_seconds']); print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
PY
