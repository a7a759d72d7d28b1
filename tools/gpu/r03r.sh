# round 3, call r: basis chunks mapped by a helper thread ahead of the iteration - tests touching the Krylov path + full bench
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x -k "krylov or basis or gmres or adjoint or edge or stagnation or block" > $O/pytest_krylov.log 2>&1; tail -3 $O/pytest_krylov.log
timeout 900 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03r/bench.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms']); print(json.dumps(d.get('solve_to_tolerance', c.get('solve_to_tolerance')), indent=None)); print(json.dumps(c.get('setup_s')))
PY
