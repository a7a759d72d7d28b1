# round 3, call k: batched PC of the block path - tests, block bench at 200 k and 2 M cells; bench with the overlapped setup
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "block_gmres or field_input or krylov_basis" > $O/pytest_block.log 2>&1; tail -6 $O/pytest_block.log | cut -c1-220
timeout 600 python tools/block_bench.py --n 100 50 40 --iters 60 > $O/block_200k.log 2>&1; grep -E "^single|^block" $O/block_200k.log | cut -c1-330
timeout 1200 python tools/block_bench.py --n 250 100 80 --iters 40 --nrhs 4 > $O/block_2M.log 2>&1; grep -E "^single|^block" $O/block_2M.log | cut -c1-330
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
grep -E "runColoring" $O/bench.err | cut -c1-260
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03k/bench.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms']); print('solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')}); print('setup',c['setup_seconds'])
PY
