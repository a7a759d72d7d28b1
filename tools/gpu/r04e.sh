export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "primal_bc or residual_parity_simplefoam or shape_total or smoke" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
