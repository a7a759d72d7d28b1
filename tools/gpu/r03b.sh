# round 3, call b: full GPU tier after the ticket / breakdown fixes + 200x loop of the Krylov edge cases
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python tools/gpu/degenerate_loop.py 200 $O/degenerate_loop_200x.log > $O/loop.out 2>&1
tail -4 $O/degenerate_loop_200x.log | cut -c1-300; tail -3 $O/loop.out | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log | cut -c1-250
