set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
python tools/adjoint_study.py --n 100 50 40 --block 1024 --fill 1 --overlap 1 --restart 1000 --maxit 1500 --threads 32 > gpurun_out/r02a/study200k.log 2>&1
tail -5 gpurun_out/r02a/study200k.log
timeout 900 python tools/adjoint_study.py --n 250 100 80 --block 1024 --fill 1 --overlap 1 --restart 400 --maxit 1200 --threads 32 > gpurun_out/r02a/study2M.log 2>&1
tail -5 gpurun_out/r02a/study2M.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r02a/prof -o bench -- python bench.py --steps 40 --warmup 5 --no-cpu > gpurun_out/r02a/bench.log 2>&1
tail -2 gpurun_out/r02a/bench.log
