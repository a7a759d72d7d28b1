# round 4, call 5j: the default bench line (driver flags) with the stage log, then the same workload under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
DAS_BENCH_VERBOSE=1 DAS_BENCH_PARITY_CPU_SECONDS=200 timeout 700 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
grep "^\[bench\|naca primal" $O/bench.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05j/bench_line.json').read().strip().splitlines()[-1]); c=d['config']
    print('value',d['value'],'ms/step',d['ms_per_step'],'solve',{k:c['solve'][k] for k in ('iterations','time_to_tolerance_s','fail','rel_residual','mean_basis_depth','iterations_per_sec_whole_solve')})
    print('roofline',d['roofline']['frac'],'spmv ms',c['spmv_ms'],'pc ms',c['pc_apply_ms'],'setup',c['setup_seconds'])
    print('cpu',{k:v for k,v in d['cpu_baseline'].items() if k not in ('sample',)})
    print('parity',c['psi_parity_200k'])
except Exception as e: print('parse failed',e)
PY
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu --no-parity --steps 20 --warmup 5 > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_wing2M.csv
head -16 $O/bench_kernel_stats_wing2M.csv | cut -c1-170
rm -rf $O/prof
