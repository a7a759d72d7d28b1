"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's Jacobian machinery:

  * stencil ("connectivity level") tables   reference src/adjoint/DAStateInfo/DAStateInfoSimpleFoam.C:78-128,
                                             DAStateInfoScalarTransportFoam.C:69-75,
                                             DASpalartAllmaras.C:364-383 (addModelResidualCon)
  * PC level reduction                       reference src/adjoint/DASolver/DASolver.C:576-705, dafoam/pyDAFoam.py:568-582
  * connectivity matrix dRdWCon              reference src/adjoint/DAJacCon/DAJacCon.C:304-667 (addStateConnections),
                                             :2039-2600 (setupdRdWCon; boundary-face levelCheck at :2459-2478)
  * distance-2 column colouring              reference src/adjoint/DAColoring/DAColoring.C:32-784 (single-rank path:
                                             sweep n colours all uncoloured columns, every row keeps its lowest-tiebreak
                                             column of colour n and un-colours the others; tiebreak srand(i);rand()%nCol :270-275)
  * colouring validity                       reference DAColoring.C:931-1037
  * calcColoredColumns                       reference DAJacCon.C:2691-2822
  * coloured finite-difference assembly      reference src/adjoint/DAPartDeriv/DAPartDeriv.C:42-107,109-208,210-315,350-473
  * operator/row scaling                     reference DASolver.C:1392-1401,2356-2455

PARITY UNPINNED (no reference test pins colours, dRdWT or psi - SURVEY.md section 4 / 8c).
Single-process restatement: the cross-rank parts (stateBoundaryCon, DAJacCon.C:800-1205) do
not exist here because the pattern is built on the global mesh.
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np
import scipy.sparse as sp

from .residual import residual

# stateResConInfo tables (levels 0..k -> list of connected states); 'nut' already renamed to
# the SA model state 'nuTilda' (DASpalartAllmaras.C:181-213)
STENCIL = {
    "DASimpleFoam": {
        "states": [("U", "vec"), ("p", "scl"), ("nuTilda", "scl"), ("phi", "face")],
        "URes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "pRes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "phiRes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "nuTildaRes": [["U", "nuTilda", "phi"], ["U", "nuTilda"], ["nuTilda"]],
    },
    # DASimpleFoam with the optional T field (DAStateInfoSimpleFoam.C:118-131).  TRes additionally lists U at level 0: the
    # boundary alphat = nut_b/Prt of a wall-function face depends on the cell velocity (the reference's table omits it).
    "DASimpleFoam+T": {
        "states": [("U", "vec"), ("p", "scl"), ("T", "scl"), ("nuTilda", "scl"), ("phi", "face")],
        "URes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "pRes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "TRes": [["U", "T", "nuTilda", "phi"], ["T", "nuTilda"], ["T"]],
        "phiRes": [["U", "p", "nuTilda", "phi"], ["U", "p", "nuTilda"], ["U"]],
        "nuTildaRes": [["U", "nuTilda", "phi"], ["U", "nuTilda"], ["nuTilda"]],
    },
    # DAStateInfoRhoSimpleFoam.C:40-47,79-116 + compressible SA (DASpalartAllmaras.C:364-383)
    "DARhoSimpleFoam": {
        "states": [("U", "vec"), ("p", "scl"), ("T", "scl"), ("nuTilda", "scl"), ("phi", "face")],
        "URes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "T"]],
        "pRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U"]],
        "TRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "p", "T"]],
        "nuTildaRes": [["U", "T", "p", "nuTilda", "phi"], ["U", "T", "p", "nuTilda"], ["T", "p", "nuTilda"]],
        "phiRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "T"]],
    },
    # DAStateInfoTurboFoam.C:44-49,82-119 (the level table equals the RhoSimpleFoam one) + compressible SA
    "DATurboFoam": {
        "states": [("U", "vec"), ("p", "scl"), ("T", "scl"), ("nuTilda", "scl"), ("phi", "face")],
        "URes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "T"]],
        "pRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U"]],
        "TRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "p", "T"]],
        "nuTildaRes": [["U", "T", "p", "nuTilda", "phi"], ["U", "T", "p", "nuTilda"], ["T", "p", "nuTilda"]],
        "phiRes": [["U", "p", "T", "nuTilda", "phi"], ["U", "p", "T", "nuTilda"], ["U", "T"]],
    },
    "DAScalarTransportFoam": {
        "states": [("T", "scl")],
        "TRes": [["T"], ["T"], ["T"]],
    },
}
MAX_RES_CON_LV_PC = {"pRes": 2, "phiRes": 1, "URes": 2, "TRes": 2, "nuTildaRes": 2}


def stencil_name(case):
    return "DASimpleFoam+T" if (case.solver_name == "DASimpleFoam" and getattr(case, "has_T", False)) else case.solver_name


def state_layout(solver, N, F):
    """DAIndex 'state' ordering offsets (reference DAIndex.C:188-258)."""
    off = {}
    size = {}
    o = 0
    for name, kind in STENCIL[solver]["states"]:
        off[name] = o
        size[name] = {"vec": 3 * N, "scl": N, "face": F}[kind]
        o += size[name]
    return off, size, o


def state_scales(case, g, normalize_states):
    """s_j of SURVEY Appendix C: normalizeStates[name] for cell states, |S_f| for phi
    (reference DAPartDeriv.C:210-315)."""
    N, F = g.nC, g.nF
    off, size, n = state_layout(stencil_name(case), N, F)
    s = np.ones(n)
    for name, kind in STENCIL[stencil_name(case)]["states"]:
        if name in normalize_states:
            if kind == "face":
                s[off[name] : off[name] + size[name]] = g.magSf
            else:
                s[off[name] : off[name] + size[name]] = normalize_states[name]
    return s


def connectivity(case, g, isPC=False):
    """dRdWCon as scipy CSR (rows = residuals, cols = states), values 1."""
    solver = stencil_name(case)
    N, F, nIF = g.nC, g.nF, g.nIF
    tbl = STENCIL[solver]
    C = g.cellCells.astype(np.int32)
    I = sp.identity(N, dtype=np.int32, format="csr")
    L = [I, C]
    L.append(((C @ C) > 0).astype(np.int32))
    L.append(((L[2] @ C) > 0).astype(np.int32))
    CF = g.cellFaces.astype(np.int32)
    kinds = dict(tbl["states"])
    # face -> adjacent cells incidence
    rows = np.concatenate([np.arange(nIF), np.arange(nIF), np.arange(nIF, F)])
    cols = np.concatenate([g.own[:nIF], g.nei, g.own[nIF:]])
    E = sp.coo_matrix((np.ones(rows.size, np.int32), (rows, cols)), shape=(F, N)).tocsr()
    E_int = E[:nIF]
    E_b = E[nIF:]

    def union(levels):
        if not levels:
            return sp.csr_matrix((N, N), dtype=np.int32)
        M = L[levels[0]]
        for k in levels[1:]:
            M = M + L[k]
        return (M > 0).astype(np.int32)

    blocks = []
    for rname, rkind in tbl["states"]:
        res = rname + "Res"
        lv = tbl[res]
        if isPC:
            lv = lv[: MAX_RES_CON_LV_PC[res] + 1]
        row_blocks = []
        for sname, skind in tbl["states"]:
            lev = [k for k, names in enumerate(lv) if sname in names]
            if skind == "face":
                if rkind == "face":
                    Pi = (E_int @ union(lev) @ CF) > 0
                    # boundary faces: phi faces added at level k if phi listed at level k (k=0)
                    # or at level k-1 (k>=1)  (DAJacCon.C levelCheck)
                    levb = [k for k in range(len(lv)) if ("phi" in lv[k] if k == 0 else "phi" in lv[k - 1])]
                    Pb = (E_b @ union(levb) @ CF) > 0
                    P = sp.vstack([Pi, Pb])
                else:
                    P = (union(lev) @ CF) > 0
            else:
                Pc = union(lev)
                if rkind == "face":
                    Pc = (E @ Pc) > 0
                P = Pc
                if skind == "vec":
                    P = sp.kron(P, np.ones((1, 3), np.int8))
            if rkind == "vec":
                P = sp.kron(P, np.ones((3, 1), np.int8))
            row_blocks.append(sp.csr_matrix(P, dtype=np.int8))
        blocks.append(row_blocks)
    con = sp.bmat(blocks, format="csr")
    con.data[:] = 1
    con.sort_indices()
    return con


# ----------------------------------------------------------------------------- colouring
_libc = ctypes.CDLL(ctypes.util.find_library("c"))
_libc.rand.restype = ctypes.c_int


def glibc_first_rand(seeds):
    """rand() right after srand(seed) - the reference's tie-breaker (DAColoring.C:270-275).
    Restated from glibc's TYPE_3 additive-feedback generator; checked against libc in tests."""
    seeds = np.asarray(seeds, dtype=np.int64)
    out = np.empty(seeds.size, dtype=np.int64)
    r = np.zeros((344, seeds.size), dtype=np.int64)
    s = seeds.copy()
    s[s == 0] = 1
    r[0] = s
    for i in range(1, 31):
        hi = r[i - 1] // 127773
        lo = r[i - 1] % 127773
        w = 16807 * lo - 2836 * hi
        w[w < 0] += 2147483647
        r[i] = w
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    o = (r[344 - 31] + r[344 - 3]) & 0xFFFFFFFF
    out[:] = o >> 1
    return out


def libc_first_rand(seed):
    _libc.srand(ctypes.c_uint(seed))
    return _libc.rand()


def d2_coloring(con):
    """Reference sweep algorithm, single rank (all columns 'strictly local').  Rows are
    processed in order within a sweep with in-place un-colouring, exactly like
    DAColoring.C:349-510."""
    con = con.tocsr()
    nR, nC = con.shape
    tb = glibc_first_rand(np.arange(nC)) % nC
    colors = np.full(nC, -1, dtype=np.int64)
    indptr, indices = con.indptr, con.indices
    active_rows = np.arange(nR)
    n = 0
    while True:
        colors[colors < 0] = n
        new_active = []
        for i in active_rows:
            cols = indices[indptr[i] : indptr[i + 1]]
            cand = cols[colors[cols] == n]
            if cand.size > 1:
                t = tb[cand]
                keep = cand[np.argmin(t)]  # first minimum == lowest column among ties
                colors[cand] = -1
                colors[keep] = n
            if cand.size > 1 or np.any(colors[cols] < 0):
                new_active.append(i)
        # rows whose columns are all coloured can never matter again
        active_rows = [i for i in new_active if np.any(colors[indices[indptr[i] : indptr[i + 1]]] < 0)]
        n += 1
        if not np.any(colors < 0):
            break
        if n > 10000:
            raise RuntimeError("more than 10000 colours")
    return colors, n


def greedy_coloring(con):
    """Fast first-fit distance-2 colouring (vectorised per column); used when the oracle
    only needs *a* valid colouring (the Jacobian does not depend on which one)."""
    csr = con.tocsr()
    csc = con.tocsc()
    nC = con.shape[1]
    colors = np.full(nC, -1, dtype=np.int64)
    for j in range(nC):
        rows = csc.indices[csc.indptr[j] : csc.indptr[j + 1]]
        nb = np.concatenate([csr.indices[csr.indptr[i] : csr.indptr[i + 1]] for i in rows]) if rows.size else np.empty(0, int)
        used = np.unique(colors[nb])
        used = used[used >= 0]
        c = 0
        for u in used:
            if u == c:
                c += 1
            elif u > c:
                break
        colors[j] = c
    return colors, int(colors.max()) + 1


def validate_coloring(con, colors):
    """No row may contain two columns of the same colour (reference DAColoring.C:931-1037)."""
    con = con.tocsr()
    rows = np.repeat(np.arange(con.shape[0]), np.diff(con.indptr))
    key = rows.astype(np.int64) * (int(colors.max()) + 2) + colors[con.indices]
    return np.unique(key).size == key.size and bool(np.all(colors >= 0))


def colored_columns(con, colors, color):
    """coloredColumn[i] = the unique column of `color` in row i, or -1 (DAJacCon.C:2691-2822)."""
    con = con.tocsr()
    rows = np.repeat(np.arange(con.shape[0]), np.diff(con.indptr))
    m = colors[con.indices] == color
    out = np.full(con.shape[0], -1, dtype=np.int64)
    out[rows[m]] = con.indices[m]
    return out


# ----------------------------------------------------------------------------- Jacobians
def jacobian_colored(case, g, W, con, colors, scales, mode="cs", isPC=False, delta=1e-6, lower_bound=1e-30):
    """dRdWT (transposed, rows = states j, cols = residuals i), entry s_j dR_i/dW_j.

    mode 'fd': the reference's one-sided coloured finite differences (DAPartDeriv.C:350-473,
               step delta*s_j); mode 'cs': complex-step on the same colouring (exact)."""
    n = W.size
    nColors = int(colors.max()) + 1
    R0 = residual(case, g, W, isPC=isPC) if mode == "fd" else None
    rows_l, cols_l, vals_l = [], [], []
    for c in range(nColors):
        m = colors == c
        if mode == "fd":
            Wp = W.copy()
            Wp[m] += delta * scales[m]
            dR = (residual(case, g, Wp, isPC=isPC) - R0) / delta
        else:
            h = 1e-40
            Wp = W.astype(np.complex128)
            Wp[m] += 1j * h * scales[m]
            dR = residual(case, g, Wp, isPC=isPC).imag / h
        cc = colored_columns(con, colors, c)
        i = np.nonzero(cc >= 0)[0]
        j = cc[i]
        v = dR[i]
        keep = (np.abs(v) > lower_bound) | (i == j) if lower_bound >= 1e-16 else np.ones(i.size, bool)
        if lower_bound < 1e-16:
            keep = np.ones(i.size, bool)
        rows_l.append(j[keep])
        cols_l.append(i[keep])
        vals_l.append(v[keep])
    A = sp.coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))), shape=(n, n))
    return A.tocsr()


def jacobian_bruteforce(case, g, W, scales, isPC=False):
    """Column-by-column complex-step dRdWT (no stencil, no colouring): ground truth for
    tiny meshes; any dependency outside the reference's stencil tables shows up here."""
    n = W.size
    h = 1e-40
    A = np.zeros((n, n))
    for j in range(n):
        Wp = W.astype(np.complex128)
        Wp[j] += 1j * h * scales[j]
        A[j, :] = residual(case, g, Wp, isPC=isPC).imag / h
    return A


def jac_t_vec(case, g, W, seed, scales):
    """(dRdW^T psi)_j * s_j by brute force is O(n) residuals; instead use the identity
    <psi, J v> for unit v?  Not needed: returns A @ seed for an assembled A elsewhere.
    Here: the matrix-free product via complex-step directional derivatives is only
    available for J v (forward).  Provided for the dot-product test."""
    h = 1e-40
    Wp = W.astype(np.complex128) + 1j * h * (scales * seed)
    return residual(case, g, Wp).imag / h
