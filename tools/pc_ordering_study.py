"""Dev tool (CPU only): how the unknown ordering inside a preconditioner block changes the ILU(1) level count (the length
of the dependent chain `k_ras_apply` walks), the fill, and the GMRES iteration count.  Block = a converged 12x10x8 channel
(oracle FD dRdWTPC, oracle GMRES); factor statistics from the product's own block factorisation (das_debug_factor_block)."""
import ctypes as C, os, sys, time
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import NORM_STATES
from dafoam_amd.meshgen import channel_case
from dafoam_amd.pyDASolvers import pyDASolvers
from dafoam_amd import _capi
from oracle import jacobian as J, linear as OL
from oracle.foam_mesh import Geometry
from oracle.primal import solve_primal

dims = (12, 10, 8)
case = channel_case(*dims, perturb=0.0, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
g = Geometry(case.mesh)
W, _ = solve_primal(case, g, max_iters=1500, tol=1e-11)
case.states = W
sc = J.state_scales(case, g, NORM_STATES)
con = J.connectivity(case, g); col, _ = J.greedy_coloring(con)
A = J.jacobian_colored(case, g, W, con, col, sc, mode="cs", lower_bound=0).tocsr()
conP = J.connectivity(case, g, isPC=True)
P = J.jacobian_colored(case, g, W, conP, col, sc, mode="fd", isPC=True, lower_bound=0).tocsr()
n = A.shape[0]
N, F = g.nC, g.nF
s = pyDASolvers(b"DASimpleFoam -python", {}, case=case)
cell_perm = s._cell_ordering_permutation()
rhs = np.zeros(n); rhs[0:3 * N:3] = g.V; rhs *= sc
L = _capi.lib()

def stats(perm, name):
    Pp = P[perm][:, perm].tocsr(); Pp.sort_indices()
    Ap = A[perm][:, perm].tocsr()
    tim = np.zeros(4); nnz = C.c_longlong(); l1 = C.c_int(); l2 = C.c_int()
    rp = Pp.indptr.astype(np.int64); ci = Pp.indices.astype(np.int32); v = Pp.data.astype(np.float64)
    _capi.check(L.das_debug_factor_block(n, rp.ctypes.data_as(_capi.c_ll_p), ci.ctypes.data_as(_capi.c_int_p), _capi.dptr(v), 1, _capi.dptr(tim), C.byref(nnz), C.byref(l1), C.byref(l2)))
    ilu = OL.ILU(Pp, fill=1)
    Ac = OL.CSR(Ap)
    x, info = OL.gmres(Ac.matvec, rhs[perm], ilu.solve, restart=300, max_iters=300, rel_tol=1e-8)
    print(f"{name:28s} levels L/U {l1.value:5d}/{l2.value:5d}  nnz(LU) {nnz.value:9d}  GMRES its {info['iters']:4d} relres {info['res']/info['res0']:.1e}", flush=True)

ident = np.arange(n)
stats(cell_perm, "cell-by-cell (default)")
stats(ident, "state ordering")
rcm = np.asarray(reverse_cuthill_mckee((P + P.T).tocsr(), symmetric_mode=True))
stats(rcm, "RCM of the PC graph")
# cell-by-cell with the cells in RCM order of the cell graph
cellG = g.cellCells.tocsr()
crcm = np.asarray(reverse_cuthill_mckee(cellG, symmetric_mode=True))
# position of each state in cell ordering is grouped by cell: rebuild by sorting cell blocks
pos_of_cell = np.empty(N, dtype=np.int64); pos_of_cell[crcm] = np.arange(N)
owner_of = np.empty(n, dtype=np.int64)
owner_of[:3 * N] = np.repeat(np.arange(N), 3); owner_of[3 * N:4 * N] = np.arange(N); owner_of[4 * N:5 * N] = np.arange(N)
owner_of[5 * N:] = g.own
order = np.lexsort((np.argsort(np.argsort(cell_perm)), pos_of_cell[owner_of]))  # cells in RCM order, inside a cell the default order
stats(order, "cell-by-cell, cells in RCM")
