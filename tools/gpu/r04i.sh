# round 3, last GPU seconds: strength-based aggregates on the 2-D NACA (400 x 125) against RCB
export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
timeout 45 python tools/adjoint_study.py --case naca --n 400 125 1 --restart 1000 --maxit 1000 --coarse-agg 128 --coarse-aggregation rcb strength 2>&1 | grep -E "iters" > $O/naca_strength.log
cat $O/naca_strength.log | cut -c1-260
