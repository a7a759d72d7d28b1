"""Dev tool: effect of the PC stencil levels (reference option maxResConLv4JacPCMat, pyDAFoam.py:568-582) and ILU fill on
preconditioner cost and GMRES convergence at bench size."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[100, 50, 40])
ap.add_argument("--cfg", type=str, nargs="+", default=["2,2,2,1,1", "1,1,1,1,1", "1,1,1,1,0", "1,1,1,0,1"],
                help="URes,pRes,nuTildaRes,phiRes levels, ILU fill")
ap.add_argument("--rtol", type=float, default=1e-6)
ap.add_argument("--restart", type=int, default=300)
ap.add_argument("--maxit", type=int, default=600)
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
L = _capi.lib()
case = bench_channel_case(*a.n)
N = case.mesh.n_cells
for cfg in a.cfg:
    u, p, nt, ph, fill = [int(x) for x in cfg.split(",")]
    opts = {"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
            "maxResConLv4JacPCMat": {"URes": u, "pRes": p, "nuTildaRes": nt, "phiRes": ph},
            "adjEqnOption": {"gmresRestart": a.restart, "gmresMaxIters": a.maxit, "gmresRelTol": a.rtol, "printInfo": 0, "pcFillLevel": fill}}
    D = PYDAFOAM(options=opts, case=case)
    n = D.getNLocalAdjointStates()
    t = time.time(); D.solver.runColoring(); tc = time.time() - t
    pc = Mat(); t = time.time(); D.solver.calcdRdWT(1, pc); tp = time.time() - t
    D.solverAD.initializedRdWTMatrixFree()
    rhs = np.zeros(n); rhs[0:3 * N:3] = 1.0 / N
    ksp = KSP(); t = time.time(); D.solverAD.createMLRKSPMatrixFree(pc, ksp); t_ilu = time.time() - t
    x = Vec(n); r = Vec(n); r.array[:] = rhs
    L.das_timer_reset(D.solver._h); L.das_timer_enable(D.solver._h, 1)
    t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, x); ts = time.time() - t
    info = ksp.info(); h = ksp.history()
    print(f"cfg {cfg}: pc nnz {pc.getInfo()['nz_used']:.3g} factor nnz {L.das_ksp_get_factor_nnz(ksp.handle):.3g} colouring {tc:.1f}s pcmat {tp:.2f}s ilu {t_ilu:.1f}s | "
          f"iters {info['iters']} fail {fail} relres {info['res']/info['res0']:.2e} solve {ts:.2f}s pc {L.das_timer_avg_ms(D.solver._h, b'pc'):.3f} ms "
          f"spmv {L.das_timer_avg_ms(D.solver._h, b'spmv'):.3f} ms", flush=True)
    print("    hist", " ".join(f"{v/h[0]:.1e}" for v in h[::max(1, len(h)//10)]), flush=True)
    L.das_timer_enable(D.solver._h, 0)
    del ksp, pc, D
