# round 4, call 5l: the default bench line with the CPU-leg deadlines (stage marks on stderr) - where does the CPU port spend its time at 2 M cells?
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
DAS_BENCH_VERBOSE=1 DAS_BENCH_CPU_DEADLINE=170 timeout 330 python bench.py --steps 20 --warmup 5 --no-parity > $O/bench_line.json 2> $O/bench.err
grep "^\[bench" $O/bench.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r05l/bench_line.json').read().strip().splitlines()[-1]); c=d['config']
    print('value',d['value'],'ms/step',d['ms_per_step'],'solve its',c['solve']['iterations'],c['solve']['time_to_tolerance_s'])
    print('cpu',{k:v for k,v in d['cpu_baseline'].items() if k not in ('sample',)})
except Exception as e: print('parse failed',e)
PY
