#!/bin/bash
# Multi-rank bench flows on the ONE GPU of the test box (all ranks on device 0, gloo host staging - DAS_BENCH_ONE_GPU / DAS_BENCH_BACKEND):
#   gpurun --timeout T -- 'bash tools/gpu/multirank.sh <tag> "<name>|<nranks>|<bench args>" ...'
# Every run leaves gpurun_out/<tag>/<name>.json (the bench line) and <name>.err (stage log, GMRES trace when DAS_GMRES_TRACE is set).
export TMPDIR=/tmp PYTHONUNBUFFERED=1 DAS_BENCH_ONE_GPU=1 DAS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 DAS_BENCH_VERBOSE=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
port=29511
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; nr=${rest%%|*}; args=${rest#*|}
  t0=$(date +%s)
  if [ "$nr" = "1" ]; then
    timeout ${RUN_TIMEOUT:-600} python bench.py --gpus 1 $args > $O/$name.json 2> $O/$name.err
  else
    port=$((port+1))
    timeout ${RUN_TIMEOUT:-600} python -m torch.distributed.run --nnodes=1 --nproc-per-node=$nr --master-addr 127.0.0.1 --master-port $port bench.py --gpus $nr $args > $O/$name.json 2> $O/$name.err
  fi
  rc=$?
  python - "$O/$name.json" "$name" "$rc" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]; s = c.get("solve") or {}
    print(f"[multirank] {sys.argv[2]}: ranks {d['n_gpus']} cells {c['global_cells']} iters {s.get('iterations')} fail {s.get('fail')} rel {s.get('rel_residual')} t {s.get('time_to_tolerance_s')} ms/it {d['ms_per_step']:.2f} overlap {c.get('asm_overlap')} |R| {c.get('primal_residual_norm'):.2e} part {str(c.get('partition'))[:40]}")
    print("            every100:", " ".join("%.1e" % v for v in (s.get("rel_residual_every_100") or [])))
except Exception as e:
    print(f"[multirank] {sys.argv[2]}: rc {sys.argv[3]} no line ({e})")
PY
  [ $rc -ne 0 ] && tail -5 $O/$name.err | cut -c1-400
  grep -h "GMRES cycle closed\|explicit projection\|sparse A Z\|node-block ILU" $O/$name.err | head -8 | cut -c1-300
  echo "[multirank] $name: $(( $(date +%s) - t0 )) s"
done
