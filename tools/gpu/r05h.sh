# round 4, call 5h: the 2 M-cell wing (converged 200 x 63 section x 160 layers) with amd.pcUpwindBlend 0.5
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python tools/naca_adjoint_sweep.py --nz 160 --dz 0.025 --blend 0.5 --combos a:additive:rcb:1 a:deflated:rcb:1 --maxit 1000 > $O/wing_dz0025.log 2> $O/wing_dz0025.err
grep "SWEEP\||R|\|extruded" $O/wing_dz0025.log; tail -2 $O/wing_dz0025.err
timeout 500 python tools/naca_adjoint_sweep.py --nz 160 --dz 0.1 --blend 0.5 --combos a:deflated:rcb:1 --maxit 1000 > $O/wing_dz01.log 2> $O/wing_dz01.err
grep "SWEEP\||R|\|extruded" $O/wing_dz01.log; tail -2 $O/wing_dz01.err
