set -x
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_coloring or naca" > $O/test_new.log 2>&1
tail -8 $O/test_new.log
DAS_DEBUG_TIMING=1 timeout 600 python bench.py --nx 100 --ny 50 --nz 40 --steps 20 --warmup 20 --no-cpu > $O/bench200k.log 2> $O/bench200k.err
tail -1 $O/bench200k.log | cut -c1-200
grep -E "colouring:" $O/bench200k.err | head
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py --no-cpu --no-solve --steps 10 --warmup 10 > $O/bench2M.log 2> $O/bench2M.err
tail -1 $O/bench2M.log | cut -c1-200
grep -E "colouring:" $O/bench2M.err | head
