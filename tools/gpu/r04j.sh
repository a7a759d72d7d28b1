export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
timeout 38 python tools/adjoint_study.py --case naca --n 400 125 4 --span 0.4 --restart 1000 --maxit 1000 --coarse-agg 512 --coarse-aggregation strength 2>&1 | grep -E "iters" > $O/naca3d_strength.log
cat $O/naca3d_strength.log | cut -c1-260
