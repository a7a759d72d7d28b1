set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "node_block_ilu" > gpurun_out/r02b/test_bilu.log 2>&1
tail -15 gpurun_out/r02b/test_bilu.log
timeout 900 python tools/adjoint_study.py --n 100 50 40 --pctype bilu --fp32 0 1 --restart 1000 --maxit 1500 > gpurun_out/r02b/study200k.log 2>&1
grep -E "node-block|^pc |hist" gpurun_out/r02b/study200k.log
timeout 1200 python tools/adjoint_study.py --n 250 100 80 --pctype bilu --fp32 0 --restart 700 --maxit 1400 --krylov-gb 95 > gpurun_out/r02b/study2M.log 2>&1
grep -E "node-block|^pc |hist|Error|error" gpurun_out/r02b/study2M.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02b/test_all.log 2>&1
tail -5 gpurun_out/r02b/test_all.log
