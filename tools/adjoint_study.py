"""Dev tool: full adjoint solve on a synthetic channel, prints setup/solve timings and convergence."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[100, 50, 40])
ap.add_argument("--block", type=int, nargs="+", default=[4096])
ap.add_argument("--rtol", type=float, default=1e-6)
ap.add_argument("--restart", type=int, default=200)
ap.add_argument("--maxit", type=int, default=600)
ap.add_argument("--grading", type=float, default=4.0)
ap.add_argument("--wf", action="store_true")
ap.add_argument("--overlap", type=int, nargs="+", default=[1])
ap.add_argument("--fill", type=int, nargs="+", default=[0])
ap.add_argument("--lx", type=float, default=2.0)
ap.add_argument("--fp32", type=int, nargs="+", default=[0])
ap.add_argument("--threads", type=int, nargs="+", default=[32])
ap.add_argument("--pctype", nargs="+", default=["bilu"])
ap.add_argument("--krylov-gb", type=float, default=32.0)
ap.add_argument("--case", default="channel", choices=["channel", "naca"], help="naca: --n = cells around, wall-normal, spanwise (BASELINE configs[1]: 800 250 1)")
ap.add_argument("--coarse-aggregation", nargs="+", default=["rcb"], help="amd.pcCoarseAggregation: rcb | strength (one run per value)")
ap.add_argument("--span", type=float, default=0.1, help="naca: spanwise extent of the extrusion")
ap.add_argument("--perturb", type=float, default=0.02, help="naca: amplitude of the seeded perturbation of the synthetic state")
ap.add_argument("--coarse-agg", type=int, nargs="+", default=[-1])
ap.add_argument("--coarse-mode", nargs="+", default=["additive"])
ap.add_argument("--pc-iters", type=int, nargs="+", default=[1], help="adjEqnOption.localPCIters (Richardson sweeps around the factorisation)")
ap.add_argument("--blend", type=float, default=0.0, help="amd.pcUpwindBlend")
ap.add_argument("--combos", nargs="+", default=None, help="explicit list fp32:pcIters:coarseAgg:coarseMode[:sweepDesign] instead of the full product")
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import channel_case, bench_channel_case, naca0012_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
case = naca0012_case(*a.n, wall_function=a.wf, span=a.span, perturb=a.perturb) if a.case == "naca" else bench_channel_case(*a.n, wall_function=a.wf)
opts = {"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRestart": a.restart, "gmresMaxIters": a.maxit, "gmresRelTol": a.rtol, "printInfo": 0}, "amd": {"pcUpwindBlend": a.blend}}
D = PYDAFOAM(options=opts, case=case)
n = D.getNLocalAdjointStates()
t = time.time(); D.solver.runColoring(); print(f"coloring {time.time()-t:.2f}s")
pc = Mat(); t = time.time(); D.solver.calcdRdWT(1, pc); print(f"dRdWTPC {time.time()-t:.2f}s nnz {pc.getInfo()['nz_used']:.3g}")
t = time.time(); D.solverAD.initializedRdWTMatrixFree(); print(f"dRdWT {time.time()-t:.2f}s")
N = case.mesh.n_cells
rhs = np.zeros(n); rhs[0:3 * N:3] = 1.0 / N
L = _capi.lib()
import itertools
runs = list(itertools.product(a.pctype, a.block, a.overlap, a.fill, a.fp32, a.threads, a.coarse_agg, a.coarse_mode, a.pc_iters))
if a.combos:
    runs = []
    for c in a.combos:
        f32, pit, cagg, cmode, *rest = c.split(":")
        runs.append((a.pctype[0], a.block[0], a.overlap[0], a.fill[0], int(f32), a.threads[0], int(cagg), cmode, int(pit), int(rest[0]) if rest else 1))
else:
    runs = [r + (1,) for r in runs]
runs = [r + (ag,) for r in runs for ag in a.coarse_aggregation]
for pct, b, ov, fl, f32, nth, cagg, cmode, pit, design, cag in runs:
    D.solver.updateDAOption({"amd": {"pcCoarseAggregation": cag, "pcSweepDesign": design, "pcType": pct, "pcCoarseAggregates": cagg, "pcCoarseMode": cmode, "maxKrylovBytes": int(a.krylov_gb * 2**30), "pcBlockCells": b, "pcFactorFP32": f32, "setupThreads": nth}, "adjEqnOption": {"asmOverlap": ov, "pcFillLevel": fl, "localPCIters": pit}})
    ksp = KSP(); t = time.time(); D.solverAD.createMLRKSPMatrixFree(pc, ksp); t_ilu = time.time() - t
    x = Vec(n); r = Vec(n); r.array[:] = rhs
    L.das_timer_reset(D.solver._h); L.das_timer_enable(D.solver._h, 1)
    t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, x); ts = time.time() - t
    info = ksp.info()
    h = ksp.history(); print("   hist", " ".join(f"{v/h[0]:.1e}" for v in h[::max(1,len(h)//12)]))
    nag, cms = L.das_ksp_get_coarse(ksp.handle, None), L.das_timer_avg_ms(D.solver._h, b"coarse")
    print(f"pc {pct} aggregation {cag} sweep-design {design} pcIters {pit} coarse {cagg}/{cmode} ({nag} aggregates, {cms:.3f} ms) block {b} overlap {ov} fill {fl} fp32 {f32} threads {nth} nblocks {L.das_ksp_get_n_blocks(ksp.handle)}: ilu {t_ilu:.2f}s  iters {info['iters']} fail {fail} relres {info['res']/info['res0']:.2e} solve {ts:.3f}s "
          f"-> {info['iters']/ts:.1f} it/s  spmv {L.das_timer_avg_ms(D.solver._h,b'spmv'):.3f} ms pc {L.das_timer_avg_ms(D.solver._h,b'pc'):.3f} ms")
    L.das_timer_enable(D.solver._h, 0)
    ksp.destroy() if hasattr(ksp, 'destroy') else None
