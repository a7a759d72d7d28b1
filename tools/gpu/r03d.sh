# round 3, call d: packed vector-row operator - parity tests + SpMV time at 2 M cells
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "packed or jac_t_vec or adjoint_vector or cyclic or distributed" > $O/pytest_packed.log 2>&1; tail -5 $O/pytest_packed.log | cut -c1-200
timeout 1500 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_packed.json 2> $O/bench_packed.err; tail -c 1500 $O/bench_packed.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d/bench_packed.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'spmv_ms',c['spmv_ms'],'pc_ms',c['pc_apply_ms'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')},'setup',c['setup_seconds'])
PY
