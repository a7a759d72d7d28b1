"""Dev tool (CPU only, oracle matrices): which coarse space removes the plateau of the NACA0012 adjoint?

Builds the exact dRdW^T of a small extruded NACA0012 O-grid with the oracle (complex-step coloured Jacobian), preconditions GMRES
with a scalar ILU(0) of the matrix plus an additive piecewise-constant PRESSURE coarse space  M^-1 = ILU^-1 + Z (Z^T A_pp Z)^-1 Z^T
and compares aggregate shapes: index-space blocks (isotropic, rays along the wall normal, rings), and an algebraic
strength-of-connection aggregation (repeated pairwise matching along the strongest |a_ij| of the p-p block - the aggregates
then follow the strong couplings of the stretched cells without knowing the mesh).  Also prints where the stagnating residual
lives (field / wall-normal layer).  Results of round 3: profiles/r04h_naca_coarse_study_cpu.log."""
import argparse, sys, os, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[96, 32, 1], help="cells around, wall-normal, spanwise")
ap.add_argument("--span", type=float, default=None)
ap.add_argument("--maxit", type=int, default=1000)
a = ap.parse_args()
from dafoam_amd.meshgen import naca0012_case
from oracle import jacobian as J, linear as OL
from oracle.foam_mesh import Geometry
from common import norm_states
na, nn, nz = a.n
t = time.time()
case = naca0012_case(na, nn, nz, span=a.span if a.span else 0.1 * nz)
g = Geometry(case.mesh)
sc = J.state_scales(case, g, norm_states(case))
con = J.connectivity(case, g)
col, _ = J.greedy_coloring(con)
A = J.jacobian_colored(case, g, case.states, con, col, sc, mode="cs", lower_bound=0).tocsr()
n, N = A.shape[0], g.nC
print(f"NACA {na} x {nn} x {nz}: {N} cells, n = {n}, nnz = {A.nnz}, {col.max() + 1} colours, assembled in {time.time() - t:.1f} s", flush=True)
rhs = np.zeros(n); rhs[0:3 * N:3] = g.V; rhs *= sc
ilu = OL.ILU(A, fill=0)
App = A[3 * N:4 * N, 3 * N:4 * N].tocsr()


def solve(agg, label):
    if agg is None:
        pc = ilu.solve
        nagg = 0
    else:
        nagg = agg.max() + 1
        Z = sp.csr_matrix((np.ones(N), (np.arange(N), agg)), shape=(N, nagg))
        Einv = np.linalg.inv((Z.T @ App @ Z).toarray())

        def pc(v):
            y = ilu.solve(v).copy()
            y[3 * N:4 * N] += Z @ (Einv @ (Z.T @ v[3 * N:4 * N]))
            return y
    x, info = OL.gmres(lambda v: A @ v, rhs, pc, restart=a.maxit, max_iters=a.maxit, rel_tol=1e-6)
    print(f"{label:46s} aggregates {nagg:5d}  iterations {info['iters']:4d}  rel {info['res'] / info['res0']:.1e}", flush=True)
    return x


x = solve(None, "scalar ILU(0), no coarse space")
xs, _ = OL.gmres(lambda v: A @ v, rhs, ilu.solve, restart=a.maxit, max_iters=min(300, a.maxit), rel_tol=1e-12)
r = rhs - A @ xs
tot = np.linalg.norm(r)
print("   residual after 300 iterations by field:", {b: round(float(np.linalg.norm(r[s]) / tot), 3) for b, s in
      (("U", slice(0, 3 * N)), ("p", slice(3 * N, 4 * N)), ("nuTilda", slice(4 * N, 5 * N)), ("phi", slice(5 * N, n)))})
kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(nn), np.arange(na), indexing="ij")  # cell = i + na (j + nn k)


def blocks(bi, bj, bk):
    nbi, nbj = (na + bi - 1) // bi, (nn + bj - 1) // bj
    return ((kk // bk) * (nbi * nbj) + (jj // bj) * nbi + (ii // bi)).ravel()


for bi, bj, bk, nm in ((6, 4, nz, "isotropic index blocks 6 x 4"), (3, 2, nz, "isotropic index blocks 3 x 2"), (na, 1, nz, "rings (all i, one j)"),
                       (1, nn, nz, "rays (one i, all j)"), (2, nn // 2, nz, "half rays 2 x nn/2")):
    solve(blocks(bi, bj, bk), nm)


def pairwise(M, passes):
    """aggregates of 2^passes cells: every node is matched with its strongest unmatched neighbour, the coarse matrix is the sum"""
    M = M.copy().tocsr()
    agg = np.arange(M.shape[0])
    for _ in range(passes):
        m = M.shape[0]
        Mc = abs(M).tocsr()
        match = -np.ones(m, int)
        cnt = 0
        for i in range(m):
            if match[i] >= 0:
                continue
            row = slice(Mc.indptr[i], Mc.indptr[i + 1])
            cols, vals = Mc.indices[row], Mc.data[row].copy()
            vals[cols == i] = -1
            vals[match[cols] >= 0] = -1
            if vals.size and vals.max() > 0:
                match[cols[vals.argmax()]] = cnt
            match[i] = cnt
            cnt += 1
        Zp = sp.csr_matrix((np.ones(m), (np.arange(m), match)), shape=(m, cnt))
        M = (Zp.T @ M @ Zp).tocsr()
        agg = match[agg]
    return agg


for passes in (4, 5, 6):
    solve(pairwise(App, passes), f"strength-based pairwise aggregation, {passes} passes")
