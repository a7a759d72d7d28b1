"""PETSc binary Vec/Mat files without PETSc (hot-path "next" row).

The reference dumps its Jacobians, colour vectors and adjoint vectors with DAUtility::writeMatrixBinary /
writeVectorBinary (reference src/adjoint/DAUtility/DAUtility.C:282-441, PetscViewerBinaryOpen + MatView/VecView) and
compares them with dafoam/scripts/dafoam_vecdiff.py / dafoam_matdiff.py.  The on-disk format is PETSc's (not defined
in /root/reference): big-endian; Vec = int32 classid 1211214, int32 n, n float64; Mat (AIJ) = int32 classid 1211216,
int32 M, N, nnz, M int32 row lengths, nnz int32 column indices, nnz float64 values.  These readers/writers make the
GPU path's matrices and vectors exchangeable with a real DAFoam run (writeJacobians: ["dRdWT", "dRdWTPC", ...]).
"""
from __future__ import annotations

import numpy as np

VEC_CLASSID = 1211214
MAT_CLASSID = 1211216


def write_vec(path, x):
    x = np.asarray(x, dtype=np.float64)
    with open(path, "wb") as f:
        np.array([VEC_CLASSID, x.size], dtype=">i4").tofile(f)
        x.astype(">f8").tofile(f)


def read_vec(path):
    with open(path, "rb") as f:
        hdr = np.fromfile(f, dtype=">i4", count=2)
        if hdr.size != 2 or hdr[0] != VEC_CLASSID:
            raise ValueError(f"{path}: not a PETSc binary Vec (classid {hdr[:1]})")
        x = np.fromfile(f, dtype=">f8", count=int(hdr[1]))
        if x.size != hdr[1]:
            raise ValueError(f"{path}: truncated Vec")
    return x.astype(np.float64)


def write_mat(path, A):
    import scipy.sparse as sp

    A = sp.csr_matrix(A)
    A.sort_indices()
    M, N = A.shape
    with open(path, "wb") as f:
        np.array([MAT_CLASSID, M, N, A.nnz], dtype=">i4").tofile(f)
        np.diff(A.indptr).astype(">i4").tofile(f)
        A.indices.astype(">i4").tofile(f)
        A.data.astype(">f8").tofile(f)


def read_mat(path):
    import scipy.sparse as sp

    with open(path, "rb") as f:
        hdr = np.fromfile(f, dtype=">i4", count=4)
        if hdr.size != 4 or hdr[0] != MAT_CLASSID:
            raise ValueError(f"{path}: not a PETSc binary Mat (classid {hdr[:1]})")
        M, N, nnz = int(hdr[1]), int(hdr[2]), int(hdr[3])
        if nnz < 0:
            raise ValueError(f"{path}: dense/blocked PETSc Mat formats are not supported")
        rl = np.fromfile(f, dtype=">i4", count=M)
        ci = np.fromfile(f, dtype=">i4", count=nnz)
        v = np.fromfile(f, dtype=">f8", count=nnz)
        if rl.size != M or ci.size != nnz or v.size != nnz or rl.sum() != nnz:
            raise ValueError(f"{path}: truncated or inconsistent Mat")
    indptr = np.concatenate([[0], np.cumsum(rl.astype(np.int64))])
    return sp.csr_matrix((v.astype(np.float64), ci.astype(np.int32), indptr), shape=(M, N))


def vecdiff(path_a, path_b, rtol=1e-8, atol=1e-16, verbose=True):
    """dafoam_vecdiff.py semantics: report the largest absolute/relative difference; returns True if within tol."""
    a, b = read_vec(path_a), read_vec(path_b)
    if a.size != b.size:
        raise ValueError("vector sizes differ")
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), 1e-300)
    i = int(np.argmax(d))
    ok = bool(np.all((d <= atol) | (rel <= rtol)))
    if verbose:
        print(f"max abs diff {d[i]:.6e} at {i} (a={a[i]:.16e}, b={b[i]:.16e}); ||a-b||/||b|| = {np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300):.6e}; {'PASS' if ok else 'FAIL'}")
    return ok


def matdiff(path_a, path_b, rtol=1e-8, atol=1e-16, verbose=True):
    A, B = read_mat(path_a), read_mat(path_b)
    if A.shape != B.shape:
        raise ValueError("matrix sizes differ")
    D = (A - B).tocoo()
    mx = float(np.abs(D.data).max()) if D.nnz else 0.0
    ref = float(np.abs(B.data).max()) if B.nnz else 1.0
    ok = mx <= atol or mx <= rtol * ref
    if verbose:
        print(f"max abs entry diff {mx:.6e} (max |B| {ref:.6e}); nnz {A.nnz} vs {B.nnz}; {'PASS' if ok else 'FAIL'}")
    return ok


if __name__ == "__main__":
    import sys

    fn = {"vecdiff": vecdiff, "matdiff": matdiff}[sys.argv[1]]
    sys.exit(0 if fn(sys.argv[2], sys.argv[3], *(float(v) for v in sys.argv[4:6])) else 1)
