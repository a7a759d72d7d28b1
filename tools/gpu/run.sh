#!/bin/bash
# One parametrised GPU launch script (replaces the per-call transcripts r03a.sh ... r05n.sh of earlier rounds).
#   gpurun --timeout T -- 'bash tools/gpu/run.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/; summaries worth keeping are copied to profiles/ by hand afterwards.
# Stages:
#   tier            the full GPU tier (pytest -m gpu), log kept
#   test:<expr>     pytest -m gpu -k <expr>
#   bench[:args]    the default bench line (args appended, ',' separated -> spaces)
#   stats[:args]    rocprofv3 --kernel-trace --stats of bench.py --no-cpu --no-parity (args appended)
#   pmc[:args]      two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) of the bench window
#   calib           the PMC counter calibration on pure streams (tools/gpu/pmc_calib.hip)
#   sweep:<args>    tools/naca_adjoint_sweep.py with the args (',' separated)
#   py:<script>[:args]   any python tool under tools/
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for st in "$@"; do
  name=${st%%:*}; args=""; [ "$st" != "$name" ] && args=$(echo "${st#*:}" | tr ',' ' ')
  t0=$(date +%s)
  case $name in
    tier)
      timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-220 ;;
    test)
      timeout 600 python -m pytest tests -x -q -s -m gpu -k "$args" > $O/pytest_k.log 2>&1; grep -v "^\[dafoam" $O/pytest_k.log | tail -8 | cut -c1-300 ;;
    bench)
      DAS_BENCH_VERBOSE=1 DAS_GMRES_TRACE=${DAS_GMRES_TRACE:-} timeout 900 python bench.py $args > $O/bench_line.json 2> $O/bench.err; grep "^\[bench" $O/bench.err | tail -40 | cut -c1-250; cut -c1-1500 $O/bench_line.json ;;
    stats)
      cd /tmp; timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-parity $args > $O/stats_bench_line.json 2> $O/stats.err
      cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -25 $O/kernel_stats.csv | cut -c1-200; rm -rf $O/stats ;;
    pmc)
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu --no-parity --no-solve --window-at-warmup $args > /dev/null 2> $O/pmc_$c.err
      done
      cd $R; python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_per_kernel.json 2> $O/pmc_summary.err; head -c 3000 $O/pmc_per_kernel.json; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
    hop)
      hipcc --offload-arch=gfx950 -O3 -o /tmp/hop tools/gpu/hop_latency.hip 2>/dev/null && timeout 60 /tmp/hop | tee $O/hop_latency.log ;;
    calib)
      hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/gpu/pmc_calib.hip && /tmp/pmc_calib 8 > $O/pmc_calib_times.log
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -o c -- /tmp/pmc_calib 8 > /dev/null 2>&1; done
      cd $R; python tools/pmc_summary.py $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE > $O/pmc_calibration.json; cat $O/pmc_calibration.json | head -c 2000; rm -rf $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE ;;
    sweep)
      timeout 900 python tools/naca_adjoint_sweep.py $args > $O/sweep.log 2> $O/sweep.err; grep SWEEP $O/sweep.log | cut -c1-300 ;;
    py)
      s=${args%% *}; a=""; [ "$args" != "$s" ] && a=${args#* }
      timeout 900 python tools/$s $a > $O/${s%.py}.log 2> $O/${s%.py}.err; tail -30 $O/${s%.py}.log | cut -c1-250 ;;
    *) echo "unknown stage $st" ;;
  esac
  echo "[run.sh] stage $st: $(( $(date +%s) - t0 )) s"
done
