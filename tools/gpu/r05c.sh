# round 4, call 5c: NACA primal levels 0-2 with a 1000-iteration inner Krylov budget; adjoints about the converged section extruded to wings
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python tools/naca_primal_study.py --out $O --levels 100 32 200 63 400 125 --lin-iters 1000 --steps 60 --adjoint-levels 1 2 \
   --extrude 1 16 2 4 1 160 2 40 --dz 0.1 0.1 0.025 0.05 --polish 2 > $O/study.log 2> $O/study.err
grep -v "^   hist" $O/study.log | tail -40
grep -c "Newton primal step" $O/study.err; grep -i "fault\|error" $O/study.err | head -5
