export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 60 python -m pytest tests -q -m gpu -x -k "dual_and_difference" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
