# round 3, call h: HBM allocation micro-benchmark; speculative colouring - tests, then setup time at 200 k and 2 M cells
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 300 ./tools/gpu/alloc_bench > $O/alloc_bench.log 2>&1; cat $O/alloc_bench.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "coloring or bench_size" > $O/pytest_color.log 2>&1; tail -12 $O/pytest_color.log | cut -c1-220
DAS_DEBUG_TIMING=1 timeout 1200 python tools/adjoint_study.py --n 250 100 80 --restart 1000 --maxit 1000 --krylov-gb 200 --combos 0:1:-1:additive > $O/study_2M_spec.log 2>&1
grep -E "^pc |coloring|colouring|dRdWT|maps:" $O/study_2M_spec.log | cut -c1-260
