export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
cat > /tmp/nk4.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
for dims in ((50, 25, 20), (100, 50, 40)):
    case = bench_channel_case(*dims)
    for amd in ({"primalSERExponent": 1.5}, {"primalSERExponent": 2.0}, {"primalSERExponent": 1.5, "primalTau0": 4.0}):
        D = PYDAFOAM(options={"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "primalMinResTol": 1e-8, "amd": amd}, case=case)
        t = time.time(); fail = D.solvePrimal(maxSteps=120); dt = time.time() - t
        print(dims, amd, "fail", fail, {k: v for k, v in D.primalInfo.items() if k != "history"}, f"{dt:.1f} s", flush=True)
PY
timeout 1500 python /tmp/nk4.py > $O/nk4.log 2>&1
grep -E "fail" $O/nk4.log | cut -c1-300
