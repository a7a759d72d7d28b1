# round 4, call 5e: Newton-Krylov pseudo-time strategies on the 400 x 125 level
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 800 python tools/naca_newton_sweep.py --out $O > $O/newton.log 2> $O/newton.err
grep -v "^   hist" $O/newton.log
