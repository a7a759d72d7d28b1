"""Dev tool (GPU): launch-shape sweep of the node-block ILU sweeps (workgroups in flight, poll back-off)
on the bench channel, then a full adjoint solve with the best shape.  Prints the PC apply time per configuration."""
import argparse, itertools, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[100, 50, 40])
ap.add_argument("--wgs", type=int, nargs="+", default=[64, 128, 256, 512, 1024])
ap.add_argument("--sleep", type=int, nargs="+", default=[0])
ap.add_argument("--fp32", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--solve", type=int, default=1)
ap.add_argument("--restart", type=int, default=1000)
ap.add_argument("--maxit", type=int, default=1500)
ap.add_argument("--krylov-gb", type=float, default=100.0)
a = ap.parse_args()
import __graft_entry__ as ge
ge.build()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
from dafoam_amd.pyDASolvers import KSP, Mat, Vec
from dafoam_amd import _capi
case = bench_channel_case(*a.n)
opts = {"solverName": "DASimpleFoam", "debug": True, "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRestart": a.restart, "gmresMaxIters": a.maxit, "gmresRelTol": 1e-6, "printInfo": 0},
        "amd": {"pcFactorFP32": a.fp32, "maxKrylovBytes": int(a.krylov_gb * 2**30)}}
D = PYDAFOAM(options=opts, case=case)
n = D.getNLocalAdjointStates()
t = time.time(); D.solver.runColoring(); print(f"coloring {time.time()-t:.2f}s", flush=True)
pc = Mat(); D.solver.calcdRdWT(1, pc)
ksp = KSP(); t = time.time(); D.solverAD.createMLRKSPMatrixFree(pc, ksp); print(f"pc setup {time.time()-t:.2f}s", flush=True)
L = _capi.lib(); h = D.solver._h
x = np.random.default_rng(0).standard_normal(n)
def timed(reps):
    L.das_timer_reset(h); L.das_timer_enable(h, 1)
    for _ in range(reps):
        y = ksp.applyPC(D.solver, x)
    ms = L.das_timer_avg_ms(h, b"pc"); L.das_timer_enable(h, 0)
    return ms, y
KNOBS = ("DAS_BILU_WGS", "DAS_BILU_SLEEP")
for k in KNOBS:
    os.environ.pop(k, None)
ms0, y0 = timed(a.reps)
print(f"default launch shape: {ms0:.3f} ms", flush=True)
best = (ms0, None)
for sl, w in itertools.product(a.sleep, a.wgs):
    os.environ["DAS_BILU_WGS"] = str(w); os.environ["DAS_BILU_SLEEP"] = str(sl)
    ms, y = timed(2 if ms0 > 20 else a.reps)
    ok = np.array_equal(y, y0) or np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0)
    print(f"wgs {w:5d} sleep {sl}: {ms:8.3f} ms {'ok' if ok else 'MISMATCH'}", flush=True)
    if ms < best[0]:
        best = (ms, (w, sl))
print("best", best, flush=True)
for k in KNOBS:
    os.environ.pop(k, None)
if best[1] is not None:
    for k, v in zip(KNOBS, best[1]):
        os.environ[k] = str(v)
if a.solve:
    t = time.time(); D.solverAD.initializedRdWTMatrixFree(); print(f"dRdWT {time.time()-t:.2f}s", flush=True)
    N = case.mesh.n_cells
    r = Vec(n); r.array[0:3 * N:3] = 1.0 / N
    xs = Vec(n)
    L.das_timer_reset(h); L.das_timer_enable(h, 1)
    t = time.time(); fail = D.solverAD.solveLinearEqn(ksp, r, xs); ts = time.time() - t
    info = ksp.info(); hist = ksp.history()
    print("   hist", " ".join(f"{v/hist[0]:.1e}" for v in hist[::max(1, len(hist)//12)]))
    print(f"solve: iters {info['iters']} fail {fail} relres {info['res']/info['res0']:.2e} {ts:.3f}s -> {info['iters']/ts:.1f} it/s  spmv {L.das_timer_avg_ms(h,b'spmv'):.3f} ms "
          f"pc {L.das_timer_avg_ms(h,b'pc'):.3f} ms refinements {L.das_ksp_get_n_refine(ksp.handle)}", flush=True)
