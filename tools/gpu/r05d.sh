# round 4, call 5d: preconditioner option sweep about the converged NACA section (2-D and 16 layers)
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python tools/naca_adjoint_sweep.py > $O/sweep.log 2> $O/sweep.err
grep "SWEEP\|extruded\||R|" $O/sweep.log; tail -3 $O/sweep.err
