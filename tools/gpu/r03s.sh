# round 3, call s: the full volCoord product on the device - tests + cost at 200 k and 2 M cells
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "device_geometry or volcoord" > $O/pytest_volcoord.log 2>&1; tail -15 $O/pytest_volcoord.log
timeout 600 python tools/volcoord_bench.py --n 100 50 40 > $O/volcoord_200k.log 2>&1; tail -6 $O/volcoord_200k.log
timeout 1200 python tools/volcoord_bench.py --n 250 100 80 > $O/volcoord_2M.log 2>&1; tail -6 $O/volcoord_2M.log
