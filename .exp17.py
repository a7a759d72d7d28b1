import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
dims = [int(v) for v in sys.argv[1:4]]
case = bench_channel_case(*dims)
res = {}
for orth in ("cgs", "dcgs2"):
    opts = {"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
            "adjEqnOption": {"printInfo": 0, "gmresRelTol": 1e-8, "gmresRestart": int(sys.argv[4]) if len(sys.argv) > 4 else 1000, "gmresMaxIters": 1500},
            "amd": {"maxKrylovBytes": int(160 * 2**30), "gmresOrthogonalization": orth}}
    D = PYDAFOAM(options=opts, case=case)
    n = D.getNLocalAdjointStates()
    D.solver.runColoring()
    pc = Mat(); D.solver.calcdRdWT(1, pc)
    ksp = KSP(); D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    D.solverAD.initializedRdWTMatrixFree()
    L = _capi.lib(); h = D.solver._h
    N = case.mesh.n_cells
    r = Vec(n); r.array[0:3 * N:3] = 1.0 / N
    xs = Vec(n)
    t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, xs); ts = time.time() - t
    info = ksp.info(); hist = np.array(ksp.history())
    # true residual through the operator
    y = Vec(n); D.solverAD.op_mult(xs, y) if hasattr(D.solverAD, "op_mult") else None
    print(f"{orth}: iters {info['iters']} fail {fail} relres {info['res']/info['res0']:.3e} {ts:.3f}s -> {info['iters']/ts:.1f} it/s", flush=True)
    res[orth] = (hist, xs.array.copy())
h1, x1 = res["cgs"]; h2, x2 = res["dcgs2"]
k = min(h1.size, h2.size) - 1   # the last entry is the true residual of the closing cycle
print("history rel diff (recurrence residuals)", np.max(np.abs(h1[:k] - h2[:k]) / h1[:k]), "lengths", h1.size, h2.size)
print("psi rel diff", np.linalg.norm(x1 - x2) / np.linalg.norm(x1))
