set -x
O=gpurun_out/r02z; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "degenerate or delayed or gmres_failure or adjoint_vector_parity or scalar_transport" > $O/test_a.log 2>&1; tail -3 $O/test_a.log
