# round 4, call 5g: pcUpwindBlend sweep about the converged section (2-D, 16 layers); BASELINE configs[1] / [2] as stated: 800 x 250 section
# converged by grid sequencing, adjoint in 2-D (200 k cells) and extruded to 10 layers (2 M cells)
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 400 python tools/naca_adjoint_sweep.py --blend 0 0.2 0.35 0.5 --combos a:additive:rcb:1 a:deflated:rcb:1 > $O/sweep.log 2> $O/sweep.err
grep "SWEEP\||R|" $O/sweep.log; tail -3 $O/sweep.err
timeout 900 python tools/naca_adjoint_sweep.py --section 800 250 --first-cell 2e-5 --nz 1 10 --dz 0.1 --blend 0 0.35 --combos a:additive:rcb:1 --maxit 1500 > $O/sweep800.log 2> $O/sweep800.err
grep "SWEEP\||R|\|primal\|extruded" $O/sweep800.log; tail -3 $O/sweep800.err
