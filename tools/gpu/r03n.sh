# round 3, call n: the multi-GPU bench flow on ONE GPU (two and four ranks, gloo staging): global coarse space + solve-to-tolerance
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
for N in 2 4; do
DAS_BENCH_ONE_GPU=1 DAS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
   bench.py --gpus $N --steps 20 --warmup 10 --nx 40 --ny 40 --nz 32 --krylov-gb 8 > $O/bench_${N}ranks_one_gpu.json 2> $O/bench_${N}ranks.err
tail -c 400 $O/bench_${N}ranks.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03n/bench_${N}ranks_one_gpu.json').read().strip().splitlines()[-1])
c=d['config']; print('N',d['n_gpus'],'value',d['value'],'cells/gpu',c['cells_per_gpu'],'global coarse',c['pc_coarse_aggregates_global'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')} if c['solve'] else None,'halo_ms',c['halo_ms'])
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 10 --nx 40 --ny 40 --nz 32 --krylov-gb 8 --no-cpu > $O/bench_1rank.json 2> $O/bench_1rank.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03n/bench_1rank.json').read().strip().splitlines()[-1])
c=d['config']; print('N 1 value',d['value'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail')})
PY
