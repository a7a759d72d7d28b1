# round 4, call 5a: NACA0012 primal by grid sequencing (Newton-Krylov, CFL ramp) + adjoint about the converged state
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python tools/naca_primal_study.py --out $O --synthetic-too > $O/study_g15.log 2> $O/study_g15.err
tail -30 $O/study_g15.log
timeout 200 python tools/naca_primal_study.py --out $O/g125 --levels 100 32 200 63 --growth 1.25 --ser 1.5 --adjoint-levels --extrude > $O/study_g125.log 2> $O/study_g125.err
tail -8 $O/study_g125.log
grep -c "Newton primal step" $O/study_g15.err
