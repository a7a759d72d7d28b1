export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "watchdog or krylov_basis or volcoord or device_geometry" > $O/pytest_sel.log 2>&1; tail -15 $O/pytest_sel.log
