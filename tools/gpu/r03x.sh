# round 3, call x: extruded NACA (400 x 125 x 8): does the pressure coarse space (aggregates, mode) move the plateau?
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout 900 python tools/adjoint_study.py --case naca --n 400 125 8 --span 0.8 --restart 1000 --maxit 1000 --combos 0:1:0:additive 0:1:1024:additive 0:1:2048:additive 0:1:2048:deflated 2>&1 | grep -E "iters|hist" > $O/naca_coarse.log
cat $O/naca_coarse.log | cut -c1-330
