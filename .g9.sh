set -x
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
DAS_DEBUG_TIMING=1 timeout 900 python tools/coloring_bench.py 100 50 40 64 128 256 512 1024 > $O/col200k.log 2>&1
grep -E "DAS_COLOR_WGS|device colouring: kernel|host prep|upload" $O/col200k.log
DAS_DEBUG_TIMING=1 timeout 1200 python tools/coloring_bench.py 250 100 80 128 256 512 1024 > $O/col2M.log 2>&1
grep -E "DAS_COLOR_WGS|device colouring: kernel|host prep|upload|colouring: prune|colouring: csc" $O/col2M.log
