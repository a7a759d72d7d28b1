"""Dev tool (GPU): pseudo-time strategies of the Newton-Krylov primal on the 400 x 125 NACA0012 level, started from the prolonged
converged 200 x 63 section (dafoam_amd/data/naca_primal_200x63.npz).  Round 4: profiles/r05e_*."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--sets", nargs="+", default=[
    '{"primalDampedSteps":"reject"}',
    '{"primalDampedSteps":"reject","primalTauGrowth":1.2,"primalSERExponent":0.5}',
    '{"primalDampedSteps":"reject","primalAcceptFactor":1.05}',
    '{"primalTauGrowth":1.2,"primalTauGrowthMax":1.2}',
    '{"primalDampedSteps":"reject","primalLinearTol":1e-3}',
    '{"primalPseudoTimeFields":"all","primalTauMode":"ser","primalTau0":1.0}',
])
ap.add_argument("--out", default="gpurun_out/newton")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import naca0012_case, prolong_naca_state
from dafoam_amd.workloads import NACA_PRIMAL_AMD
from dafoam_amd.pyDAFoam import PYDAFOAM
d = np.load(os.path.join(ROOT, "dafoam_amd", "data", "naca_primal_200x63.npz"))
nxc, nyc = [int(v) for v in d["dims"]]
fcc = float(d["first_cell"])
nx, ny, fc = 2 * nxc, 125, fcc / 2
case = naca0012_case(nx, ny, 1, first_cell=fc, perturb=0.0)
W0 = prolong_naca_state((nxc, nyc), d["states"], case, (nx, ny), first_cell=fc, coarse_first_cell=fcc)
best = None
for js in a.sets:
    amd = dict(NACA_PRIMAL_AMD, primalLinearIters=1000)
    amd.update(json.loads(js))
    case.states = W0.copy()
    D = PYDAFOAM(options={"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "amd": amd}, case=case)
    t = time.time()
    print("SET", js, flush=True); print("SET", js, file=sys.stderr, flush=True)
    try:
        fail, info = D.solver.solvePrimal(maxSteps=a.steps, relTol=1e-9, absTol=0.0)
    except Exception as e:  # noqa: BLE001
        print("   FAILED", e, flush=True)
        del D
        continue
    print(f"   fail {fail} steps {info['steps']} linear {info['linearIterations']} |R| {info['res0']:.3e} -> {info['res']:.3e} in {time.time() - t:.1f}s", flush=True)
    print("   hist", " ".join(f"{v:.1e}" for v in info["history"]), flush=True)
    if fail == 0 and best is None:
        best = js
        np.savez_compressed(os.path.join(a.out, f"naca_primal_{nx}x{ny}.npz"), states=D.getStates(), dims=np.array([nx, ny]), first_cell=fc, res0=info["res0"], res=info["res"])
    del D
print("first converging set:", best)
