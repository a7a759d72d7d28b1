"""Dev tool (GPU): preconditioner option sweep for the adjoint about the CONVERGED NACA0012 section (dafoam_amd/data/naca_primal_200x63.npz,
produced by tools/naca_primal_study.py), extruded to --nz layers.  Round 4: profiles/r05d_*."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--nz", type=int, nargs="+", default=[1, 16])
ap.add_argument("--dz", type=float, default=0.1)
ap.add_argument("--section", type=int, nargs=2, default=None, help="converge this section (n_around n_normal) by grid sequencing instead of loading the 200 x 63 data file")
ap.add_argument("--first-cell", type=float, default=2e-5)
ap.add_argument("--combos", nargs="+", default=["-1:additive:rcb:1", "512:additive:rcb:1", "2048:additive:rcb:1", "-1:deflated:rcb:1", "2048:deflated:rcb:1", "-1:additive:strength:1",
                                                "2048:additive:strength:1", "-1:additive:rcb:2", "0:additive:rcb:1"], help="coarseAgg:coarseMode:aggregation:localPCIters (coarseAgg 'a' = automatic)")
ap.add_argument("--maxit", type=int, default=1500)
ap.add_argument("--blend", type=float, nargs="+", default=[0.0], help="amd.pcUpwindBlend values (the PC matrix is re-assembled per value)")
ap.add_argument("--polish", type=int, default=2)
ap.add_argument("--deflation", nargs="*", default=[], help="GMRES-DR runs after the sweep: pairs m:k (gmresRestart : amd.gmresDeflation), e.g. 300:100 200:70")
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import naca0012_case
from dafoam_amd.workloads import naca_extruded_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
L = _capi.lib()
if a.section:
    from dafoam_amd.workloads import naca_converged_primal
    nx, ny = a.section
    fc = a.first_cell
    t0 = time.time()
    case2, lv = naca_converged_primal(nx, ny, options={"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}}, first_cell=fc, verbose=True)
    print(f"primal by grid sequencing: {time.time() - t0:.1f} s", flush=True)
else:
    d = np.load(os.path.join(ROOT, "dafoam_amd", "data", "naca_primal_200x63.npz"))
    nx, ny = [int(v) for v in d["dims"]]
    fc = float(d["first_cell"])
    case2 = naca0012_case(nx, ny, 1, first_cell=fc, perturb=0.0)
    case2.states = d["states"].copy()
opts = {"solverName": "DASimpleFoam", "debug": False, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRestart": a.maxit, "gmresMaxIters": a.maxit, "gmresRelTol": 1e-6, "printInfo": 0}, "amd": {"maxKrylovBytes": int(140 * 2**30)}}
for nz in a.nz:
    if nz > 1:
        case, ex = naca_extruded_case(case2, (nx, ny), nz, dz=a.dz, first_cell=fc, options=opts, polish_steps=a.polish)
        print(f"extruded {nx}x{ny}x{nz}: polish {ex}", flush=True)
    else:
        case = case2
    N = case.mesh.n_cells
    D = PYDAFOAM(options=opts, case=case)
    n = D.getNLocalAdjointStates()
    R = np.zeros(n); D.solver.getResiduals(R); print(f"|R| {np.linalg.norm(R):.3e}", flush=True)
    D.solver.runColoring()
    D.solverAD.initializedRdWTMatrixFree()
    rhs = np.zeros(n); rhs[0:3 * N:3] = 1.0 / N
    for blend in a.blend:
      D.solver.updateDAOption({"amd": {"pcUpwindBlend": float(blend)}})
      pc = Mat(); D.solver.calcdRdWT(1, pc)
      for c in a.combos:
        cagg, cmode, cag, pit = c.split(":")
        cagg = -1 if cagg == "a" else cagg
        D.solver.updateDAOption({"amd": {"pcCoarseAggregates": int(cagg), "pcCoarseMode": cmode, "pcCoarseAggregation": cag}, "adjEqnOption": {"localPCIters": int(pit)}})
        ksp = KSP(); t = time.time(); D.solverAD.createMLRKSPMatrixFree(pc, ksp); t_ilu = time.time() - t
        x = Vec(n); r = Vec(n); r.array[:] = rhs
        L.das_timer_reset(D.solver._h); L.das_timer_enable(D.solver._h, 1)
        t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, x); ts = time.time() - t
        info = ksp.info(); h = ksp.history()
        plateau = int(np.argmax(h < 0.5 * h[0])) if len(h) else -1
        print(f"SWEEP {nx}x{ny}x{nz} blend {blend} combo {c:28s} ({L.das_ksp_get_coarse(ksp.handle, None)} agg): iters {info['iters']} plateau {plateau} fail {fail} rel {info['res'] / info['res0']:.2e} solve {ts:.2f}s ilu {t_ilu:.2f}s "
              f"pc {L.das_timer_avg_ms(D.solver._h, b'pc'):.3f} ms coarse {L.das_timer_avg_ms(D.solver._h, b'coarse'):.3f} ms", flush=True)
        L.das_timer_enable(D.solver._h, 0)
        ksp.destroy()
      pc.destroy()
    for mk in a.deflation:
        mm, kk = [int(v) for v in mk.split(":")]
        D.solver.updateDAOption({"amd": {"gmresDeflation": kk, "pcUpwindBlend": float(a.blend[-1])}, "adjEqnOption": {"gmresRestart": mm, "gmresMaxIters": a.maxit}})
        pc = Mat(); D.solver.calcdRdWT(1, pc)
        ksp = KSP(); D.solverAD.createMLRKSPMatrixFree(pc, ksp)
        x = Vec(n); r = Vec(n); r.array[:] = rhs
        t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, x); ts = time.time() - t
        info = ksp.info()
        print(f"SWEEP {nx}x{ny}x{nz} GMRES-DR(m={mm}, k={kk}) blend {a.blend[-1]}: iters {info['iters']} fail {fail} rel {info['res'] / info['res0']:.2e} solve {ts:.2f}s "
              f"({info['iters'] / ts:.1f} it/s, basis {mm + 2} vectors)", flush=True)
        D.solver.updateDAOption({"amd": {"gmresDeflation": 0}, "adjEqnOption": {"gmresRestart": a.maxit}})
        ksp.destroy(); pc.destroy()
    del D
