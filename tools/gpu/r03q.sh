# round 3, call q: is the slower window of r03p reproducible?  bench twice (no cpu, no solve) + kernel stats
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
for i in 1 2; do
timeout 900 python bench.py --no-cpu --no-solve > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03q/bench_$i.json').read().strip().splitlines()[-1])
c=d['config']; print('run $i value',d['value'],'ms',d['ms_per_step'],'spmv',c['spmv_ms'],'pc',c['pc_apply_ms'],'coarse',c['coarse_ms'])
PY
done
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu --no-solve > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_2M.csv
rm -rf $O/prof
python - <<'PY'
import csv,json
d=json.loads(open('gpurun_out/r03q/bench_prof.json').read().strip().splitlines()[-1]); print('profiled run ms', d['ms_per_step'])
for row in csv.DictReader(open('gpurun_out/r03q/bench_kernel_stats_2M.csv')):
    n=row['Name']
    if any(k in n for k in ('multidot2','dcgs2','spmv','bilu_sweep','k_reduce','coarse','tr_','sort_rows','net_','color_first')):
        print(n[:44].replace('void das::','').replace('das::',''), row['Calls'], round(float(row['AverageNs'])/1e6,3),'ms')
PY
