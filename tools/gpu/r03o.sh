# round 3, call o: the native RCCL transport with a world of one rank (load, CommInitRank, in-stream all-reduce)
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "native_rccl" > $O/pytest_rccl.log 2>&1; tail -15 $O/pytest_rccl.log | cut -c1-250
