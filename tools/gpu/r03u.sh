# round 3, call u: the bench line on an extruded NACA0012 O-grid at 2 M cells (800 x 250 x 10)
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py --workload naca --no-cpu > $O/bench_naca.json 2> $O/bench_naca.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03u/bench_naca.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'colors',c['colors'],'nnz',c['dRdWT_nnz']); print(json.dumps(c['solve'])); print(json.dumps(c['setup_seconds'])); print(json.dumps(d['roofline']))
PY
tail -5 $O/bench_naca.err
