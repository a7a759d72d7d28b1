"""Dev tool (CPU only): phase times of one preconditioner block factorisation on a synthetic block-sized pattern
(PC stencil of a 12x11x10 hex block in cell-by-cell ordering, diagonally dominant random values)."""
import ctypes as C, sys, os, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dafoam_amd.meshgen import channel_case
from dafoam_amd.pyDASolvers import pyDASolvers
from dafoam_amd import _capi
dims = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (12, 11, 10)
case = channel_case(*dims, perturb=0.0)
s = pyDASolvers(b"DASimpleFoam -python", {}, case=case)
s.runColoring()
P = s.getConnectivity(1).T.tocsr()  # dRdWTPC pattern
n = P.shape[0]
perm = s._cell_ordering_permutation()  # cell-by-cell ordering like inside a block
P = P[perm][:, perm].tocsr(); P.sort_indices()
rng = np.random.default_rng(0)
P.data = rng.standard_normal(P.nnz) * 0.1
P = (P + sp.diags(np.full(n, 50.0))).tocsr(); P.sort_indices()
L = _capi.lib()
for fill in (0, 1):
    tim = np.zeros(4); nnz = C.c_longlong(); l1 = C.c_int(); l2 = C.c_int()
    rp = P.indptr.astype(np.int64); ci = P.indices.astype(np.int32); v = P.data.astype(np.float64)
    t = time.time()
    _capi.check(L.das_debug_factor_block(n, rp.ctypes.data_as(_capi.c_ll_p), ci.ctypes.data_as(_capi.c_int_p), _capi.dptr(v), fill, _capi.dptr(tim), C.byref(nnz), C.byref(l1), C.byref(l2)))
    print(f"n {n} nnz(A) {P.nnz} ILU({fill}): nnz(LU) {nnz.value} levels {l1.value}/{l2.value}  total {time.time()-t:.2f}s  symbolic {tim[0]:.2f} numeric {tim[1]:.2f} schedules {tim[2]:.2f} streams {tim[3]:.2f}")
