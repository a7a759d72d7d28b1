# round 3, call 4b: does FILL matter on the stretched O-grid?  round-1 RAS + ILU(k) blocks (host factorisation) at fill 0 / 1 vs node-block ILU(0), NACA 400 x 125 x 4
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 1200 python tools/adjoint_study.py --case naca --n 400 125 4 --span 0.4 --restart 1000 --maxit 1000 --pctype ras --block 4096 --fill 0 1 --overlap 1 --coarse-agg 0 2>&1 | grep -E "iters|hist" > $O/naca_ras_fill.log
timeout 600 python tools/adjoint_study.py --case naca --n 400 125 4 --span 0.4 --restart 1000 --maxit 1000 --pctype bilu --coarse-agg 0 2>&1 | grep -E "iters|hist" >> $O/naca_ras_fill.log
cat $O/naca_ras_fill.log | cut -c1-330
