export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "volcoord" > $O/pytest_sel.log 2>&1; tail -25 $O/pytest_sel.log
