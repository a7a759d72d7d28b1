# round 3, call w: why does the extruded NACA adjoint stall?  iterations vs spanwise layers / extent / state noise at 50 k cells per layer
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
for cfg in "400 125 1 0.1 0.02" "400 125 4 0.1 0.02" "400 125 4 0.4 0.02" "400 125 4 0.4 0.0" "400 125 8 0.8 0.02"; do
set -- $cfg
echo "== naca $1 x $2 x $3 span $4 perturb $5" >> $O/naca_span.log
timeout 300 python tools/adjoint_study.py --case naca --n $1 $2 $3 --span $4 --perturb $5 --restart 1000 --maxit 1000 2>&1 | grep -E "iters|hist" >> $O/naca_span.log
done
cat $O/naca_span.log | cut -c1-330
