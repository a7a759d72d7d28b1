# round 4, call 5k: the new / changed GPU tests on the final tree
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_naca.py tests/test_gpu_bench_flow.py tests/test_gpu_parity.py -q -x -s -k "naca_primal or naca_wing or bench_flow or volcoord_dual or newton_krylov_primal" > $O/pytest_new.log 2>&1
grep -v "^\[dafoam" $O/pytest_new.log | tail -25 | cut -c1-300
