"""Dev tool (GPU): converge the DASimpleFoam + SA primal on the NACA0012 O-grid by grid sequencing with the Newton-Krylov
primal (CFL ramp), then solve the adjoint about the CONVERGED state (2-D levels and spanwise extrusions of it) for a matrix of
preconditioner options.  Writes the converged states to --out.  Round 4: profiles/r05a_*."""
import argparse, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="+", default=[100, 32, 200, 63, 400, 125, 800, 250], help="n_around n_normal per level")
ap.add_argument("--first-cell", type=float, default=2e-5, help="first cell height of the FINEST level (coarser levels: x2 per level)")
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--tol", type=float, default=1e-8)
ap.add_argument("--tau0", type=float, default=1.0)
ap.add_argument("--growth", type=float, default=1.5)
ap.add_argument("--ser", type=float, default=1.0)
ap.add_argument("--lin-iters", type=int, default=300)
ap.add_argument("--lin-tol", type=float, default=1e-2)
ap.add_argument("--out", default="gpurun_out/naca")
ap.add_argument("--adjoint-levels", type=int, nargs="*", default=[2, 3], help="level indices whose 2-D adjoint is solved")
ap.add_argument("--extrude", type=int, nargs="*", default=[2, 4, 2, 8], help="pairs (level index, nz) of extruded adjoints")
ap.add_argument("--dz", type=float, nargs="+", default=[0.1], help="spanwise layer thickness per --extrude pair (last value repeats)")
ap.add_argument("--orderings", nargs="+", default=["rcm"])
ap.add_argument("--coarse", type=int, nargs="+", default=[-1])
ap.add_argument("--synthetic-too", action="store_true", help="also solve the adjoint about the synthetic noisy state (round-3 workload)")
ap.add_argument("--polish", type=int, default=0, help="Newton steps on the extruded mesh before its adjoint")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import naca0012_case, prolong_naca_state, extrude_naca_state
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
L = _capi.lib()
NORM = {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}
levels = [(a.levels[2 * i], a.levels[2 * i + 1]) for i in range(len(a.levels) // 2)]
fcs = [a.first_cell * 2 ** (len(levels) - 1 - i) for i in range(len(levels))]


def opts(extra_amd=None, adj=None):
    amd = {"coloringAlgorithm": "speculative", "primalTauMode": "ramp", "primalTau0": a.tau0, "primalTauGrowth": a.growth, "primalSERExponent": a.ser,
           "primalLinearIters": a.lin_iters, "primalLinearTol": a.lin_tol, "maxKrylovBytes": int(140 * 2**30)}
    amd.update(extra_amd or {})
    o = {"solverName": "DASimpleFoam", "debug": True, "normalizeStates": dict(NORM), "primalMinResTol": a.tol,
         "adjEqnOption": dict({"gmresRestart": 1000, "gmresMaxIters": 1000, "gmresRelTol": 1e-6, "printInfo": 0}, **(adj or {})), "amd": amd}
    return o


def adjoint_matrix(case, tag, dims):
    N = case.mesh.n_cells
    for order in a.orderings:
        t0 = time.time()
        D = PYDAFOAM(options=opts(adj={"jacMatReOrdering": order}), case=case)
        n = D.getNLocalAdjointStates()
        D.solver.runColoring()
        pc = Mat(); D.solver.calcdRdWT(1, pc)
        D.solverAD.initializedRdWTMatrixFree()
        t_setup = time.time() - t0
        rhs = np.zeros(n); rhs[0:3 * N:3] = 1.0 / N
        for cagg in a.coarse:
            D.solver.updateDAOption({"amd": {"pcCoarseAggregates": cagg}})
            ksp = KSP(); t = time.time(); D.solverAD.createMLRKSPMatrixFree(pc, ksp); t_ilu = time.time() - t
            x = Vec(n); r = Vec(n); r.array[:] = rhs
            L.das_timer_reset(D.solver._h); L.das_timer_enable(D.solver._h, 1)
            t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, x); ts = time.time() - t
            info = ksp.info(); h = ksp.history()
            print(f"ADJOINT {tag} dims {dims} order {order} coarse {cagg} ({L.das_ksp_get_coarse(ksp.handle, None)} agg): iters {info['iters']} fail {fail} rel {info['res'] / info['res0']:.2e} "
                  f"solve {ts:.2f}s setup {t_setup:.1f}s ilu {t_ilu:.2f}s spmv {L.das_timer_avg_ms(D.solver._h, b'spmv'):.3f} pc {L.das_timer_avg_ms(D.solver._h, b'pc'):.3f} ms", flush=True)
            print("   hist", " ".join(f"{v / h[0]:.1e}" for v in h[::max(1, len(h) // 12)]), flush=True)
            L.das_timer_enable(D.solver._h, 0)
        del D


W_prev, conv = None, {}
for li, ((nx, ny), fc) in enumerate(zip(levels, fcs)):
    t0 = time.time()
    case = naca0012_case(nx, ny, 1, first_cell=fc, perturb=0.0)
    if W_prev is not None:
        case.states = prolong_naca_state(levels[li - 1], W_prev, case, (nx, ny), first_cell=fc, coarse_first_cell=fcs[li - 1])
    D = PYDAFOAM(options=opts(), case=case)
    R = np.zeros(D.getNLocalAdjointStates()); D.solver.getResiduals(R)
    print(f"LEVEL {li}: {nx} x {ny} first cell {fc:.1e}: |R0| {np.linalg.norm(R):.3e}  (case {time.time() - t0:.1f}s)", flush=True)
    t = time.time()
    try:
        _, info = D.solver.solvePrimal(maxSteps=a.steps, relTol=a.tol, absTol=0.0)
    except Exception as e:  # noqa: BLE001
        print("PRIMAL FAILED", e, flush=True)
        break
    W = D.getStates()
    print(f"LEVEL {li} primal: steps {info['steps']} linear iterations {info['linearIterations']} |R| {info['res0']:.3e} -> {info['res']:.3e} in {time.time() - t:.1f}s", flush=True)
    print("   hist", " ".join(f"{v:.2e}" for v in info["history"]), flush=True)
    np.savez_compressed(os.path.join(a.out, f"naca_primal_{nx}x{ny}.npz"), states=W, dims=np.array([nx, ny]), first_cell=fc, res0=info["res0"], res=info["res"])
    conv[li] = (case, W, info["res"] <= a.tol * info["res0"] * 100)
    W_prev = W
    del D

for li in a.adjoint_levels:
    if li in conv:
        case, W, ok = conv[li]
        case.states = W
        adjoint_matrix(case, f"2D-converged({ok})", levels[li])
        if a.synthetic_too:
            adjoint_matrix(naca0012_case(*levels[li], 1, first_cell=fcs[li]), "2D-synthetic-noisy", levels[li])
for q in range(len(a.extrude) // 2):
    li, nz = a.extrude[2 * q], a.extrude[2 * q + 1]
    if li not in conv:
        continue
    case2, W2, ok = conv[li]
    nx, ny = levels[li]
    dz = a.dz[min(q, len(a.dz) - 1)]
    case3 = naca0012_case(nx, ny, nz, span=dz * nz, first_cell=fcs[li], perturb=0.0)
    case3.states = extrude_naca_state(case2, W2, case3, (nx, ny, nz))
    if a.polish:
        D = PYDAFOAM(options=opts({"primalTau0": 1e3}), case=case3)
        _, info = D.solver.solvePrimal(maxSteps=a.polish, relTol=1e-12, absTol=0.0)
        print(f"POLISH {nx}x{ny}x{nz}: |R| {info['res0']:.3e} -> {info['res']:.3e} steps {info['steps']} lin {info['linearIterations']}", flush=True)
        case3.states = D.getStates()
        del D
    adjoint_matrix(case3, f"3D-extruded-converged({ok}) dz {dz}", (nx, ny, nz))
