"""Dev tool (GPU): time runColoring for a list of in-flight workgroup counts of the device first-fit (DAS_COLOR_WGS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
dims = tuple(int(x) for x in sys.argv[1:4])
case = bench_channel_case(*dims)
for w in sys.argv[4:]:
    os.environ["DAS_COLOR_WGS"] = w
    D = PYDAFOAM(options={"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}}, case=case)
    t = time.time(); D.solver.runColoring(); dt = time.time() - t
    print(f"DAS_COLOR_WGS {w}: runColoring {dt:.2f} s, {D.solver.getColoring()[1]} colours", flush=True)
    del D
