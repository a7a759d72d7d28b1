# round 3, call a: how the round-2 head fails on the degenerate meshes (history of every bad run)
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python tools/gpu/degenerate_loop.py 150 $O/degenerate_loop_r02head.log > $O/loop.out 2>&1
tail -5 $O/degenerate_loop_r02head.log | cut -c1-400
grep -c BAD $O/degenerate_loop_r02head.log
