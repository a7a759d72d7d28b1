# round 4, call 5n (last GPU seconds): the 198 k-cell wing test - adjoint in budget + psi vs the independent CPU solve (threads = cgroup quota)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05n; mkdir -p $O
timeout 170 python -m pytest tests/test_gpu_naca.py -q -x -s -k "naca_wing" > $O/pytest_wing.log 2>&1
grep -v "^\[dafoam" $O/pytest_wing.log | tail -12 | cut -c1-300
