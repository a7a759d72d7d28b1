set -x
export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
true
true
timeout 600 python tools/adjoint_study.py --n 100 50 40 --pctype bilu --restart 1000 --maxit 1500 --coarse-agg 0 -1 512 --coarse-mode additive deflated > $O/study200k.log 2>&1
grep -E "^pc |hist|coarse space" $O/study200k.log
timeout 1500 python tools/adjoint_study.py --n 250 100 80 --pctype bilu --restart 700 --maxit 1400 --krylov-gb 95 --coarse-agg -1 --coarse-mode additive deflated > $O/study2M.log 2>&1
grep -E "^pc |hist|coarse space|runColoring|Error" $O/study2M.log
