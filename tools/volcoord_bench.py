"""Dev tool: cost of the full mesh-sensitivity product calcJacTVecProduct(volCoord -> residual | function) at bench sizes
(coloured central differences on the device, csrc/das_volcoord.hpp)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[100, 50, 40])
ap.add_argument("--check", type=int, default=1, help="contract with a displacement field and compare with the directional product")
a = ap.parse_args()
from dafoam_amd.meshgen import bench_channel_case
from dafoam_amd.pyDAFoam import PYDAFOAM
t = time.time()
case = bench_channel_case(*a.n)
print(f"case {a.n}: {case.mesh.n_cells} cells, {case.mesh.n_points} points, generated in {time.time() - t:.1f} s", flush=True)
opts = {"solverName": "DASimpleFoam", "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}, "adjEqnOption": {"printInfo": 0},
        "function": {"CD": {"type": "force", "source": "patchToFace", "patches": ["bottom"], "directionMode": "fixedDirection", "direction": [1.0, 0.0, 0.0], "scale": 1.0}}}
D = PYDAFOAM(options=opts, case=case)
S = D.solverAD
n, P3 = case.states.size, 3 * case.mesh.n_points
X0 = case.mesh.points.ravel().copy()
# a smooth seed field (an adjoint vector is smooth; random seeds make the contraction check below a sum of cancelling terms)
N = case.mesh.n_cells
xc = np.linspace(0.0, 1.0, n)
seeds = 1.0 + 0.5 * np.sin(7.0 * xc)
for out, nm, sd in (("residual", "residual", seeds), ("function", "CD", np.ones(1))):
    p = np.zeros(P3)
    t = time.time(); S.calcJacTVecProduct("x", "volCoord", X0, nm, out, sd, p); wall = time.time() - t
    i = S._volCoordInfo
    print(f"{out}: {i['colors']} point colours, {i['passes']} passes, device loop {i['seconds']:.2f} s ({1e3 * i['seconds'] / i['passes']:.2f} ms per pass), "
          f"influence + colouring build {i['build_seconds']:.2f} s (first call only), wall {wall:.2f} s, |product|max {np.abs(p).max():.3e}", flush=True)
    if a.check:
        Xr = case.mesh.points
        dX = np.stack([0.3 * np.sin(3 * Xr[:, 1]) * Xr[:, 0], 0.2 * Xr[:, 0] * (1 - Xr[:, 0]), 0.1 * np.cos(2 * Xr[:, 0])], axis=1).ravel()
        t = time.time(); dr = S.calcVolCoordDirectionalProduct(dX, nm, out, sd, eps=1e-6); td = time.time() - t
        print(f"   product . dX = {p @ dX:.10e}   directional (host metrics, 2 updates) = {dr:.10e}   diff / sum|terms| {abs(p @ dX - dr) / np.abs(p * dX).sum():.1e}   ({td:.2f} s)", flush=True)
