# round 3, call 4d: final tree - full GPU tier, then the 2-rank bench flow on one GPU (gloo staging)
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
DAS_BENCH_ONE_GPU=1 DAS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
   bench.py --gpus 2 --steps 20 --warmup 10 --nx 40 --ny 40 --nz 32 --krylov-gb 8 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
tail -c 300 $O/bench_2ranks.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d/bench_2ranks_one_gpu.json').read().strip().splitlines()[-1])
c=d['config']; print('N',d['n_gpus'],'value',d['value'],'cells/gpu',c['cells_per_gpu'],'global coarse',c['pc_coarse_aggregates_global'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')} if c['solve'] else None,'halo_ms',c['halo_ms'])
PY
