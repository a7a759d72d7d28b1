# round 4, call 5b: NACA0012 primal by grid sequencing (pseudo-time on the transport rows only) + adjoint about the converged state
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 1100 python tools/naca_primal_study.py --out $O --synthetic-too --steps 120 > $O/study.log 2> $O/study.err
grep -v "^   hist" $O/study.log | tail -40
grep -c "Newton primal step" $O/study.err; grep -i "fault\|error" $O/study.err | head -5
