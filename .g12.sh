export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
cat > /tmp/nk3.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from dafoam_amd.meshgen import channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
case = channel_case(10, 8, 6, perturb=0.0, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
for amd in ({"primalLinearTol": 1e-9, "primalPCLag": 1}, {}):
    D = PYDAFOAM(options={"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "primalMinResTol": 1e-10, "amd": amd}, case=case)
    print("=== amd", amd, flush=True)
    fail = D.solvePrimal(maxSteps=40)
PY
timeout 600 python /tmp/nk3.py > $O/nk3.log 2>&1
grep -E "===|Newton primal step|fail" $O/nk3.log | cut -c1-200
