# round 3, call z: full GPU tier on the final tree
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
