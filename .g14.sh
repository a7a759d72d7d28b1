export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "newton_krylov" > $O/test_new.log 2>&1
tail -12 $O/test_new.log | cut -c1-300
cat > /tmp/nk5.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from dafoam_amd.meshgen import naca0012_case
from dafoam_amd.pyDAFoam import PYDAFOAM
case = naca0012_case(200, 60, 1, first_cell=1e-4)
for amd in ({"primalTau0": 0.05}, {"primalTau0": 0.01, "primalSERExponent": 1.0}):
    D = PYDAFOAM(options={"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "primalMinResTol": 1e-8, "amd": amd}, case=case)
    t = time.time(); fail = D.solvePrimal(maxSteps=150); dt = time.time() - t
    print("naca", amd, "fail", fail, {k: v for k, v in D.primalInfo.items() if k != "history"}, f"{dt:.1f} s", flush=True)
    print("   hist", " ".join(f"{v:.1e}" for v in D.primalInfo["history"][::5]), flush=True)
PY
timeout 900 python /tmp/nk5.py > $O/nk5.log 2>&1
grep -E "^naca|hist" $O/nk5.log | cut -c1-500
timeout 1700 python bench.py --converge-primal --cpu-solve > $O/bench_primal.log 2> $O/bench_primal.err
tail -1 $O/bench_primal.log | cut -c1-300
