# round 3, call p: device graph set-up - full GPU tier + default bench
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=6 > $O/pytest_gpu.log 2>&1
tail -14 $O/pytest_gpu.log | cut -c1-220
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
grep -E "runColoring|device graph|colouring:|maps:" $O/bench.err | head -20 | cut -c1-220
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03p/bench.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'colors',c['colors'],'roofline',d['roofline']['frac'])
print('solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')}); print('setup',c['setup_seconds'])
PY
