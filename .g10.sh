set -x
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "newton_krylov or naca or device_coloring" > $O/test_new.log 2>&1
tail -30 $O/test_new.log
cat > /tmp/nk.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from dafoam_amd.meshgen import bench_channel_case, naca0012_case
from dafoam_amd.pyDAFoam import PYDAFOAM
for name, case in (("channel 50x25x20", bench_channel_case(50, 25, 20)), ("channel 100x50x40", bench_channel_case(100, 50, 40)), ("naca 200x60", naca0012_case(200, 60, 1, first_cell=1e-4))):
    D = PYDAFOAM(options={"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "primalMinResTol": 1e-8}, case=case)
    t = time.time(); fail = D.solvePrimal(maxSteps=60); dt = time.time() - t
    print(name, "fail", fail, {k: v for k, v in D.primalInfo.items() if k != "history"}, f"{dt:.1f} s", flush=True)
    print("   hist", " ".join(f"{v:.1e}" for v in D.primalInfo["history"]), flush=True)
PY
timeout 1500 python /tmp/nk.py > $O/nk.log 2>&1
grep -E "fail|hist|Error|error" $O/nk.log | cut -c1-400
