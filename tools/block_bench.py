"""Dev tool (GPU): block (multi right-hand-side) GMRES vs the single-system solver on the bench channel - time per block
iteration and per right-hand side for s = 1 (single), 2, 4, 8 with a fixed iteration budget, kernel times from the library's
timers (spmv = SpMM for the block path, pc = one preconditioner apply)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[100, 50, 40])
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--nrhs", type=int, nargs="+", default=[2, 4, 8])
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
case = bench_channel_case(*a.n)
opts = {"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRestart": a.iters, "gmresMaxIters": a.iters, "gmresRelTol": 1e-30, "gmresAbsTol": 1e-300, "printInfo": 0},
        "amd": {"maxKrylovBytes": int(100 * 2**30)}}
D = PYDAFOAM(options=opts, case=case)
n = D.getNLocalAdjointStates()
D.solver.runColoring()
pc = Mat(); D.solver.calcdRdWT(1, pc)
ksp = KSP(); D.solverAD.createMLRKSPMatrixFree(pc, ksp)
D.solverAD.initializedRdWTMatrixFree()
L = _capi.lib()
N = case.mesh.n_cells
rng = np.random.default_rng(0)
base = np.zeros(n); base[0:3 * N:3] = 1.0 / N
def timers():
    return {k: (L.das_timer_avg_ms(D.solver._h, k.encode()), L.das_timer_count(D.solver._h, k.encode())) for k in ("spmv", "pc", "coarse")}
x = Vec(n); r = Vec(n); r.array[:] = base
D.solverAD.solveLinearEqn(ksp, r, x)  # warm-up (allocations)
L.das_timer_reset(D.solver._h); L.das_timer_enable(D.solver._h, 1)
t = time.time(); D.solverAD.solveLinearEqn(ksp, r, x); t1 = time.time() - t
it1 = ksp.info()["iters"]
print(f"single: {it1} iterations in {t1:.3f} s -> {t1 / it1 * 1e3:.2f} ms per iteration; timers {timers()}", flush=True)
for s in a.nrhs:
    B = np.stack([base * (1.0 + 0.1 * k) + 1e-3 * rng.standard_normal(n) * (base != 0) for k in range(s)], axis=1)
    X = np.zeros((n, s))
    D.solverAD.solveLinearEqnBlock(ksp, B, X)  # warm-up
    L.das_timer_reset(D.solver._h)
    t = time.time(); fail, r0, r1 = D.solverAD.solveLinearEqnBlock(ksp, B, X); ts = time.time() - t
    itb = a.iters // s if False else None
    print(f"block s={s}: {ts:.3f} s for the budget of {a.iters} block-vector products; relres {np.max(r1 / r0):.2e}; timers {timers()}", flush=True)
