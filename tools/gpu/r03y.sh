# round 3, call y: the driver's sequence on the final tree - full GPU tier, smoke, default bench, rocprofv3 kernel stats
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
DAS_DEBUG_TIMING=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03y/bench.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'colors',c['colors']); print(json.dumps(c['solve'])); print(json.dumps(c['setup_seconds'])); print(json.dumps(d['roofline'])); print(json.dumps(d['cpu_baseline'])[:600])
PY
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu --no-solve > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_2M.csv
rm -rf $O/prof
head -12 $O/bench_kernel_stats_2M.csv | cut -c1-160
