set -x
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_coloring or two_level or bench_size" > $O/test_new.log 2>&1
tail -8 $O/test_new.log
DAS_DEBUG_TIMING=1 timeout 1500 python bench.py > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log | cut -c1-300
grep -E "colouring:|maps:|runColoring|coarse|node-block" $O/bench.err | head -30
