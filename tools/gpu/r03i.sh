# round 3, call i: bitmap first-fit colouring + VM-mapped Krylov basis - tests, setup and solve time at 2 M cells
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "coloring or bench_size or adjoint_vector" > $O/pytest_color.log 2>&1; tail -12 $O/pytest_color.log | cut -c1-220
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py --steps 100 --warmup 100 > $O/bench.json 2> $O/bench.err
grep -E "colouring|maps:|runColoring" $O/bench.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i/bench.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'colors',c['colors'],'roofline',d['roofline']['frac'])
print('solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')}); print('setup',c['setup_seconds']); print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
PY
