# round 4, call 5i: the new default bench line (NACA wing about the converged primal) + the new GPU tests
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
free -g | head -2 > $O/host.txt; nproc >> $O/host.txt; cat $O/host.txt
DAS_BENCH_VERBOSE=1 timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -5 $O/bench.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05i/bench_line.json').read().strip().splitlines()[-1]); c=d['config']
    print('value',d['value'],'ms/step',d['ms_per_step'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail','rel_residual','mean_basis_depth','iterations_per_sec_whole_solve')})
    print('roofline',d['roofline']['frac'],'spmv ms',c['spmv_ms'],'pc ms',c['pc_apply_ms'],'setup',c['setup_seconds'])
    print('primal',{k:c['primal_newton_krylov'][k] for k in ('seconds','seconds_2d')}, c['primal_residual_norm'])
    print('cpu',{k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
    print('parity',c['psi_parity_200k'])
except Exception as e: print('parse failed',e)
PY
timeout 900 python -m pytest tests/test_gpu_naca.py tests/test_gpu_parity.py -q -x -k "naca or volcoord_dual" -s > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log | cut -c1-250
