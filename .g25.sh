set -x
mkdir -p gpurun_out/r02x
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -x -q -k "delayed or gmres_failure or two_rank_sharded or adjoint_vector_parity" > gpurun_out/r02x/test.log 2>&1; tail -3 gpurun_out/r02x/test.log
timeout 600 python bench.py --no-cpu > gpurun_out/r02x/bench_nocpu.json 2> gpurun_out/r02x/bench.err; tail -c 400 gpurun_out/r02x/bench_nocpu.json
