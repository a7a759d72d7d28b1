export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "device_geometry or volcoord" > $O/pytest_volcoord.log 2>&1; tail -15 $O/pytest_volcoord.log
