# round 3, call 4a: hypothesis - the NACA plateau is the spanwise copies of the in-plane pressure modes (aggregates that span all layers cannot hold them)
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
for sc in "1,1,1" "1,1,1000"; do
echo "== DAS_COARSE_SCALE=$sc" >> $O/naca_zsplit.log
DAS_COARSE_SCALE=$sc timeout 900 python tools/adjoint_study.py --case naca --n 400 125 8 --span 0.8 --restart 1000 --maxit 1000 --combos 0:1:1560:additive 0:1:2048:additive 2>&1 | grep -E "iters|hist" >> $O/naca_zsplit.log
done
cat $O/naca_zsplit.log | cut -c1-330
