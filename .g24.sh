set -x
mkdir -p gpurun_out/r02w
timeout 300 python tools/orth_bench.py --K 150 > gpurun_out/r02w/orth.log 2>&1; cat gpurun_out/r02w/orth.log | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "residual_parity or golden or naca or force_function or drdwt_dual or delayed" > gpurun_out/r02w/test.log 2>&1; tail -3 gpurun_out/r02w/test.log
