# round 3, call f: sweep design 2 (ticket per wave, pipelined) - parity tests, then A/B on the channel (200 k, 2 M) and the NACA0012 O-grid
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zzz_edge_cases.py -q -m gpu -x -k "node_block_ilu or degenerate or packed or two_level" > $O/pytest_pc.log 2>&1; tail -4 $O/pytest_pc.log | cut -c1-200
timeout 600 python tools/adjoint_study.py --n 100 50 40 --restart 1000 --maxit 1000 --krylov-gb 100 --combos 0:1:-1:additive:1 0:1:-1:additive:2 1:1:-1:additive:2 > $O/ab_200k.log 2>&1
grep -E "^pc " $O/ab_200k.log | cut -c1-260
timeout 600 python tools/adjoint_study.py --case naca --n 800 250 1 --restart 1000 --maxit 1000 --krylov-gb 100 --combos 0:1:-1:additive:1 0:1:-1:additive:2 > $O/ab_naca200k.log 2>&1
grep -E "^pc " $O/ab_naca200k.log | cut -c1-260
timeout 1200 python tools/adjoint_study.py --n 250 100 80 --restart 1000 --maxit 1000 --krylov-gb 200 --combos 0:1:-1:additive:1 0:1:-1:additive:2 1:1:-1:additive:2 > $O/ab_2M.log 2>&1
grep -E "^pc " $O/ab_2M.log | cut -c1-260
