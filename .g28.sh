set -x
O=gpurun_out/r02z; mkdir -p $O
timeout 330 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
