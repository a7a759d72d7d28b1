# round 3, call 4f: the tree with kernels templated on the metric scalar + the exact (dual-number) volCoord product: full GPU tier
# (no -x: every test reports), then the cost of the exact product at 2 M cells
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 420 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 100 python tools/volcoord_bench.py --n 250 100 80 --check 0 > $O/volcoord_dual_2M.log 2>&1; tail -3 $O/volcoord_dual_2M.log
