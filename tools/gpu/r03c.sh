# round 3, call c: edge-case tests again + preconditioner option matrix at 200 k and 2 M cells (iterations / time to 1e-6)
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_zzz_edge_cases.py -q -m gpu -x > $O/pytest_edge.log 2>&1; tail -3 $O/pytest_edge.log
timeout 900 python tools/adjoint_study.py --n 100 50 40 --restart 1000 --maxit 1000 --krylov-gb 200 \
   --combos 0:1:-1:additive 1:1:-1:additive 0:2:-1:additive 0:1:512:additive 0:1:2048:additive 0:1:2048:deflated 0:2:2048:additive 1:2:2048:additive > $O/study_200k.log 2>&1
grep -E "^pc |coloring|dRdWT" $O/study_200k.log | cut -c1-260
timeout 1500 python tools/adjoint_study.py --n 250 100 80 --restart 1000 --maxit 1000 --krylov-gb 200 \
   --combos 0:1:-1:additive 1:1:-1:additive 0:2:-1:additive 0:1:2048:additive 0:1:2048:deflated 1:2:2048:additive > $O/study_2M.log 2>&1
grep -E "^pc |coloring|dRdWT" $O/study_2M.log | cut -c1-260
