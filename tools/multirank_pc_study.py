"""Dev tool (CPU only; the oracle's host adjoint + the host twin of the node-block ILU(0)): GMRES iterations of the NACA0012 wing adjoint
when the preconditioner is cut into per-rank sub-domains - block-Jacobi (overlap 0, what rounds 3-5 ran across ranks) against restricted
additive Schwarz with `asmOverlap` rings of ghost cells (the reference's ASM, DALinearEqn.C:212-216), for the partitions bench.py offers:
spanwise slabs, sectors around the airfoil, and wall-normal-ray blocks in the (around, normal) index plane that keep every spanwise
column of cells on one rank.  The converged 200 x 63 section of dafoam_amd/data is extruded to nz layers of dz chords.
usage: multirank_pc_study.py nz dz world [partition:overlap ...]      e.g.  8 0.025 2 one:0 span:0 span:1 around:0 around:1"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import norm_states, options  # noqa: E402

from dafoam_amd.meshgen import extrude_naca_state, naca0012_case  # noqa: E402
from dafoam_amd.pyDASolvers import pyDASolvers  # noqa: E402
from oracle import jacobian as J  # noqa: E402
from oracle import linear as OL  # noqa: E402
from oracle.adjoint_host import HostAdjoint  # noqa: E402
from oracle.foam_mesh import Geometry  # noqa: E402

nz, dz, world = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
configs = [c.split(":") for c in (sys.argv[4:] or ["one:0", "span:0", "span:1", "around:0", "around:1"])]
blend = float(os.environ.get("STUDY_BLEND", "0.5"))
d = np.load(os.path.join(ROOT, "dafoam_amd", "data", "naca_primal_200x63.npz"))
na, nn = [int(v) for v in d["dims"]]
fc = float(d["first_cell"])
t0 = time.time()
case2 = naca0012_case(na, nn, 1, first_cell=fc, perturb=0.0)
case2.states = d["states"].copy()
if nz > 1:
    case = naca0012_case(na, nn, nz, span=dz * nz, first_cell=fc, y_wall_section=case2.y_wall, perturb=0.0)
    case.states = extrude_naca_state(case2, case2.states, case, (na, nn, nz))
else:
    case = case2
g = Geometry(case.mesh)
N = g.nC
H = HostAdjoint(case, g)
sc = J.state_scales(case, g, norm_states(case))
ncol = H.setup()
W = np.asarray(case.states, dtype=np.float64)
print(f"wing {na} x {nn} x {nz} (dz {dz}): {N} cells, |R| = {np.linalg.norm(H.residual(W)):.3e}, {ncol} colours, {time.time() - t0:.1f} s", flush=True)
P = H.assemble(W, sc, True, pc_blend=blend)
A = H.assemble(W, sc, False)
n = A[0].size - 1
print(f"assembled: nnz(A) {A[1].size}, nnz(P) {P[1].size}, {time.time() - t0:.1f} s", flush=True)
Ps = sp.csr_matrix((P[2], P[1], P[0]), shape=(n, n))
K = OL.OmpKrylov()
K.set_operator(A)
del A
s = pyDASolvers(b"DASimpleFoam -python", options(case, amd={"pcUpwindBlend": blend}), case=case)
S = s.pcStructure()
rhs = np.zeros(n)
rhs[0 : 3 * N : 3] = 1.0 / N

# anchor cell of every state (cell states: the cell; phi: the face's owner cell)
own = np.asarray(case.mesh.owner, dtype=np.int64)
cell_of = np.concatenate([np.repeat(np.arange(N), 3), np.arange(N), np.arange(N), own])
nIF = case.mesh.n_internal_faces
nei = np.asarray(case.mesh.neighbour, dtype=np.int64)
Adj = sp.coo_matrix((np.ones(nIF, np.int8), (own[:nIF], nei)), shape=(N, N)).tocsr()
Adj = ((Adj + Adj.T) > 0).astype(np.int8).tocsr()
cid = np.arange(N)
ci_, cj_, ck_ = cid % na, (cid // na) % nn, cid // (na * nn)


def partition(kind):
    if kind == "one":
        return np.zeros(N, np.int32)
    if kind == "span":
        return (ck_ * world // nz).astype(np.int32)
    if kind == "around":
        return (ci_ * world // na).astype(np.int32)
    raise ValueError(kind)


def gmres(pc, maxit=1000, rtol=1e-6):
    V = np.zeros((maxit + 1, n))
    Hm = np.zeros((maxit + 1, maxit))
    beta = np.linalg.norm(rhs)
    V[0] = rhs / beta
    gvec = np.zeros(maxit + 1)
    gvec[0] = beta
    cs, sn = np.zeros(maxit), np.zeros(maxit)
    hist = [1.0]
    for j in range(maxit):
        w = K.matvec(pc(V[j]))
        for _ in range(2):
            h = V[: j + 1] @ w
            w -= h @ V[: j + 1]
            Hm[: j + 1, j] += h
        Hm[j + 1, j] = np.linalg.norm(w)
        V[j + 1] = w / Hm[j + 1, j]
        for i in range(j):
            t = cs[i] * Hm[i, j] + sn[i] * Hm[i + 1, j]
            Hm[i + 1, j] = -sn[i] * Hm[i, j] + cs[i] * Hm[i + 1, j]
            Hm[i, j] = t
        r = np.hypot(Hm[j, j], Hm[j + 1, j])
        cs[j], sn[j] = Hm[j, j] / r, Hm[j + 1, j] / r
        Hm[j, j] = r
        gvec[j + 1] = -sn[j] * gvec[j]
        gvec[j] *= cs[j]
        hist.append(abs(gvec[j + 1]) / beta)
        if hist[-1] < rtol:
            break
    return len(hist) - 1, hist


for kind, ov in configs:
    ov = int(ov)
    part = partition(kind)
    nr = int(part.max()) + 1
    t1 = time.time()
    Ks, masks, owns = [], [], []
    for r in range(nr):
        inS = part == r
        for _ in range(ov):
            inS = inS | (Adj @ inS.astype(np.int8) > 0)
        mS = inS[cell_of]
        mO = (part == r)[cell_of]
        Dm = sp.diags(mS.astype(np.float64))
        Pr = (Dm @ Ps @ Dm + sp.diags((~mS).astype(np.float64))).tocsr()
        Pr.sort_indices()
        Kr = OL.OmpKrylov()
        Kr.set_pc_bilu((Pr.indptr.astype(np.int64), Pr.indices.astype(np.int32), Pr.data), S)
        Ks.append(Kr)
        masks.append(mS)
        owns.append(mO)

    def pc(v):
        z = np.zeros(n)
        for Kr, mS, mO in zip(Ks, masks, owns):
            z += np.where(mO, Kr.pc_solve(np.where(mS, v, 0.0)), 0.0)
        return z

    its, hist = gmres(pc)
    print(f"STUDY {na}x{nn}x{nz} dz {dz} blend {blend}: partition {kind:7s} ranks {nr} overlap {ov}: {its} iterations, rel {hist[-1]:.2e}, "
          f"every 100: {' '.join('%.1e' % v for v in hist[::100])}  ({time.time() - t1:.0f} s)", flush=True)
    del Ks
