"""Synthetic polyMesh + case generators (INPUT DATA for both the HIP path and the oracle).

The reference reads an OpenFOAM case directory (constant/polyMesh + 0/ fields); its
regression meshes (ConvergentChannel, NACA0012, pitzDaily...) are downloaded at
test time (reference tests/Allrun:8-18) and are not available here.  These
generators emit the *same data model* (points / faces / owner / neighbour /
patch table in OpenFOAM ordering) for hex blocks that are then treated as fully
unstructured by everything downstream.

Face ordering follows OpenFOAM's polyMesh contract: internal faces first in
upper-triangular order (by owner, then by neighbour), then boundary faces
patch by patch; every face normal points from owner to neighbour (or out of
the domain).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import os

import numpy as np

# boundary-condition type codes shared with the C-ABI (include/dafoam_amd.h)
BC_FIXED_VALUE = 0
BC_ZERO_GRADIENT = 1
BC_INLET_OUTLET = 2
BC_SYMMETRY = 3
BC_CYCLIC = 4  # coupled (cyclic) patch: no patch-field coefficients, the paired cell acts as a neighbour
# nut wall treatment codes
NUT_CALCULATED = 0
NUT_LOWRE_WALL = 1  # nut_w = 0 (nutLowReWallFunction, reference DAField.C:1155-1218)
NUT_SPALDING_WALL = 2  # nutUSpaldingWallFunction (reference DAMisc/nutUSpaldingWallFunctionDF)
NUT_SYMMETRY = 3


@dataclass
class Patch:
    name: str
    type: str  # "patch" | "wall" | "symmetry" | "cyclic"
    start: int
    size: int
    neighbour: Optional[str] = None  # cyclic: name of the paired patch (face k pairs with face k)
    rotation: Optional[np.ndarray] = None  # cyclic, rotational: 3x3 forwardT (neighbour-side vectors -> this side); None = translational


@dataclass
class PolyMesh:
    points: np.ndarray  # (P,3) float64
    face_ptr: np.ndarray  # (F+1,) int32 CSR into face_pts
    face_pts: np.ndarray  # int32 point ids
    owner: np.ndarray  # (F,) int32
    neighbour: np.ndarray  # (Fi,) int32
    patches: List[Patch]

    @property
    def n_cells(self) -> int:
        if not self.owner.size:
            return 0
        return int(max(self.owner.max(), self.neighbour.max() if self.neighbour.size else 0)) + 1

    @property
    def n_faces(self) -> int:
        return int(self.owner.size)

    @property
    def n_internal_faces(self) -> int:
        return int(self.neighbour.size)

    @property
    def n_points(self) -> int:
        return int(self.points.shape[0])


@dataclass
class FoamCase:
    """Everything the reference would read from the case directory for the hot path."""

    mesh: PolyMesh
    solver_name: str  # "DASimpleFoam" | "DAScalarTransportFoam"
    nu: float = 1.5e-5
    # per patch name -> {field: (bc_code, value)}; value scalar or 3-vector
    bcs: Dict[str, Dict[str, tuple]] = field(default_factory=dict)
    relax: Dict[str, float] = field(default_factory=lambda: {"U": 0.7, "nuTilda": 0.7, "T": 1.0})
    y_wall: Optional[np.ndarray] = None  # frozen wall distance (reference DASpalartAllmaras.C:94)
    states: Optional[np.ndarray] = None  # W in DAIndex "state" ordering
    # scalar transport extras
    DT: float = 0.01
    deltaT: float = 1.0
    phi: Optional[np.ndarray] = None  # frozen face flux (ScalarTransport)
    T_old: Optional[np.ndarray] = None
    # compressible thermo (hePsiThermo/perfectGas/hConst/const transport, reference DAResidual.C:179-293)
    thermo: Dict[str, float] = field(default_factory=lambda: {"Cp": 1005.0, "molWeight": 28.96, "mu": 1.8e-5, "Pr": 0.7, "Prt": 1.0})
    # constant/MRFProperties (one zone = the whole mesh): {"omega": (3,), "origin": (3,), "nonRotatingPatches": [names]}
    mrf: Optional[dict] = None
    # system/fvSolution SIMPLE.transonic and the reference option transonicPCOption (DATurboFoam)
    transonic: bool = False
    transonic_pc_option: int = 1
    # system/fvSolution SIMPLE.consistent (SIMPLEC form of the DASimpleFoam pressure equation, DAResidualSimpleFoam.C:187-194)
    simple_consistent: bool = False
    # DASimpleFoam with the optional passive T field (0/T present; Pr, Prt from transportProperties -> thermo["Pr"/"Prt"]):
    # states [U | p | T | nuTilda | phi]
    has_T: bool = False


def hex_block(
    nx: int,
    ny: int,
    nz: int,
    lengths=(1.0, 1.0, 1.0),
    mapping: Optional[Callable[[np.ndarray], np.ndarray]] = None,
    patch_names=("inlet", "outlet", "bottom", "top", "front", "back"),
    patch_types=("patch", "patch", "wall", "wall", "symmetry", "symmetry"),
    grading_y: float = 1.0,
    x_range=None,
) -> PolyMesh:
    """Structured nx*ny*nz hex block emitted as an unstructured polyMesh.
    x_range=(e0, e1, NX): emit only the cell columns e0 <= i < e1 of a global block with NX cells in x (nx must be
    e1-e0); point coordinates are those of the global lattice, so sub-blocks of different ranks coincide.

    mapping: optional function on the (P,3) unit-box point array (after scaling)
    used to bend/skew the block (creates non-orthogonality).
    grading_y: two-sided geometric clustering towards y=0 and y=Ly (ratio of
    centre/wall cell size).
    """
    Lx, Ly, Lz = lengths
    xs = np.linspace(0.0, Lx, nx + 1)
    if x_range is not None:
        e0, e1, NXg = x_range
        assert nx == e1 - e0
        xs = np.linspace(0.0, Lx, NXg + 1)[e0 : e1 + 1]
    if grading_y != 1.0:
        t = np.linspace(-1.0, 1.0, ny + 1)
        beta = np.log(grading_y)
        ys = 0.5 * Ly * (1.0 + np.tanh(beta * t) / np.tanh(beta))
    else:
        ys = np.linspace(0.0, Ly, ny + 1)
    zs = np.linspace(0.0, Lz, nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if mapping is not None:
        pts = np.ascontiguousarray(mapping(pts))

    def pid(i, j, k):
        return i + (nx + 1) * (j + (ny + 1) * k)

    def cid(i, j, k):
        return i + nx * (j + ny * k)

    I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    # cell-major order: cell index increasing -> iterate k, j, i
    order = np.argsort(cid(I, J, K).ravel())
    I, J, K = I.ravel()[order], J.ravel()[order], K.ravel()[order]
    c = cid(I, J, K)

    faces = []
    own = []
    nei = []
    # per cell, faces to +x, +y, +z neighbours (ascending neighbour id) -> upper-triangular order
    # build per-direction then interleave by (owner, neighbour) sort
    fx_mask = I < nx - 1
    fy_mask = J < ny - 1
    fz_mask = K < nz - 1

    def quad_x(i, j, k):  # face at x = i (normal +x)
        return np.stack([pid(i, j, k), pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i, j, k + 1)], axis=1)

    def quad_y(i, j, k):  # face at y = j (normal +y)
        return np.stack([pid(i, j, k), pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j, k)], axis=1)

    def quad_z(i, j, k):  # face at z = k (normal +z)
        return np.stack([pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i, j + 1, k)], axis=1)

    fo = np.concatenate([c[fx_mask], c[fy_mask], c[fz_mask]])
    fn = np.concatenate([c[fx_mask] + 1, c[fy_mask] + nx, c[fz_mask] + nx * ny])
    fq = np.concatenate(
        [
            quad_x(I[fx_mask] + 1, J[fx_mask], K[fx_mask]),
            quad_y(I[fy_mask], J[fy_mask] + 1, K[fy_mask]),
            quad_z(I[fz_mask], J[fz_mask], K[fz_mask] + 1),
        ]
    )
    o = np.lexsort((fn, fo))
    own.append(fo[o])
    nei.append(fn[o])
    faces.append(fq[o])
    n_int = fo.size

    patches: List[Patch] = []
    start = n_int

    def add_patch(idx, quads, cells):
        nonlocal start
        faces.append(quads)
        own.append(cells)
        patches.append(Patch(patch_names[idx], patch_types[idx], start, len(cells)))
        start += len(cells)

    JJ, KK = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    JJ, KK = JJ.ravel(), KK.ravel()
    o = np.argsort(cid(0, JJ, KK))
    # x- : outward normal -x -> reverse winding of quad_x
    add_patch(0, quad_x(np.zeros_like(JJ), JJ, KK)[o][:, ::-1], cid(0, JJ, KK)[o])
    o = np.argsort(cid(nx - 1, JJ, KK))
    add_patch(1, quad_x(np.full_like(JJ, nx), JJ, KK)[o], cid(nx - 1, JJ, KK)[o])
    II, KK = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    II, KK = II.ravel(), KK.ravel()
    o = np.argsort(cid(II, 0, KK))
    add_patch(2, quad_y(II, np.zeros_like(II), KK)[o][:, ::-1], cid(II, 0, KK)[o])
    o = np.argsort(cid(II, ny - 1, KK))
    add_patch(3, quad_y(II, np.full_like(II, ny), KK)[o], cid(II, ny - 1, KK)[o])
    II, JJ = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    II, JJ = II.ravel(), JJ.ravel()
    o = np.argsort(cid(II, JJ, 0))
    add_patch(4, quad_z(II, JJ, np.zeros_like(II))[o][:, ::-1], cid(II, JJ, 0)[o])
    o = np.argsort(cid(II, JJ, nz - 1))
    add_patch(5, quad_z(II, JJ, np.full_like(II, nz))[o], cid(II, JJ, nz - 1)[o])

    fq = np.concatenate(faces).astype(np.int32)
    F = fq.shape[0]
    return PolyMesh(
        points=pts.astype(np.float64),
        face_ptr=(4 * np.arange(F + 1)).astype(np.int32),
        face_pts=np.ascontiguousarray(fq.ravel()),
        owner=np.concatenate(own).astype(np.int32),
        neighbour=np.concatenate(nei).astype(np.int32),
        patches=patches,
    )


def bump_mapping(height: float = 0.1, skew: float = 0.15, Lx: float = 1.0, Ly: float = 1.0, z_periodic: bool = False):
    """Convergent-channel-like bump on the bottom wall plus a sinusoidal interior skew
    (gives non-orthogonal, non-uniform hexes like the reference's ConvergentChannel)."""

    def f(p):
        q = p.copy()
        x, y, z = p[:, 0] / Lx, p[:, 1] / Ly, p[:, 2]
        bump = height * Ly * np.sin(np.pi * x) ** 2
        q[:, 1] = p[:, 1] + bump * (1.0 - y)
        q[:, 0] = p[:, 0] + skew * Lx / 8.0 * np.sin(np.pi * y) * np.sin(2 * np.pi * x)
        if not z_periodic:  # (a z-translation-invariant block is needed for front/back cyclic pairs)
            q[:, 2] = z * (1.0 + 0.05 * np.sin(np.pi * x) * np.sin(np.pi * y))
        return q

    return f


class _InputGeometry:
    """Face/cell geometry used ONLY to build consistent synthetic input fields (phi from
    U, wall distance).  The product's geometry lives in csrc/das_mesh.cpp; the oracle's
    in oracle/foam_mesh.py - neither is imported here."""

    def __init__(self, mesh: PolyMesh):
        if mesh.n_faces >= 1_000_000 and not os.environ.get("DAS_MESHGEN_NUMPY") and self._native(mesh):
            return
        F = mesh.n_faces
        nv = np.diff(mesh.face_ptr)
        assert np.all(nv == nv[0]), "input generator handles uniform polygons"
        k = int(nv[0])
        # (component-wise over the k triangles of the fan: np.cross / np.roll on (F, k, 3) temporaries cost 4x as much at 6 M faces)
        fp = mesh.face_pts.reshape(F, k)
        X = [np.ascontiguousarray(mesh.points[:, d]) for d in range(3)]
        Pk = [[X[d][fp[:, i]] for d in range(3)] for i in range(k)]  # k x 3 arrays of length F
        fc = [sum(Pk[i][d] for i in range(k)) / k for d in range(3)]
        Sf = [np.zeros(F) for _ in range(3)]
        ac = [np.zeros(F) for _ in range(3)]
        asum = np.zeros(F)
        for i in range(k):
            p, q = Pk[i], Pk[(i + 1) % k]
            u = [q[d] - p[d] for d in range(3)]
            v = [fc[d] - p[d] for d in range(3)]
            n = [u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]]
            a = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
            asum += a
            for d in range(3):
                Sf[d] += n[d]
                ac[d] += a * (p[d] + q[d] + fc[d])
        self.Sf = 0.5 * np.stack(Sf, axis=1)
        self.Cf = np.stack(ac, axis=1) / (3.0 * asum)[:, None]
        N = mesh.n_cells
        nIF = mesh.n_internal_faces
        own, nei = mesh.owner, mesh.neighbour
        cnt = np.bincount(own, minlength=N) + np.bincount(nei, minlength=N)
        cE = np.zeros((N, 3))
        for d in range(3):
            cE[:, d] = (np.bincount(own, self.Cf[:, d], N) + np.bincount(nei, self.Cf[:nIF, d], N)) / cnt
        pv_o = np.einsum("ij,ij->i", self.Sf, self.Cf - cE[own])
        pv_n = np.einsum("ij,ij->i", self.Sf[:nIF], cE[nei] - self.Cf[:nIF])
        pc_o = 0.75 * self.Cf + 0.25 * cE[own]
        pc_n = 0.75 * self.Cf[:nIF] + 0.25 * cE[nei]
        V3 = np.bincount(own, pv_o, N) + np.bincount(nei, pv_n, N)
        self.C = np.zeros((N, 3))
        for d in range(3):
            self.C[:, d] = (np.bincount(own, pv_o * pc_o[:, d], N) + np.bincount(nei, pv_n * pc_n[:, d], N)) / V3
        self.V = V3 / 3.0
        so = np.abs(np.einsum("ij,ij->i", self.Sf[:nIF], self.Cf[:nIF] - self.C[own[:nIF]]))
        sn = np.abs(np.einsum("ij,ij->i", self.Sf[:nIF], self.C[nei] - self.Cf[:nIF]))
        self.w = sn / (so + sn)


def _native_metrics(self, mesh: PolyMesh) -> bool:
    """Bench sizes (>= 1 M faces): the library's host geometry bodies over all threads (das_mesh_metrics) instead of the numpy fan sums
    below - 30 s -> ~2 s at 2 M cells (bench hygiene).  Same formulas (fvMesh metrics), different summation order; the
    small meshes of the tests always take the numpy path.  False if the library is not built."""
    try:
        import ctypes as C

        from . import _capi

        L = _capi.lib()
    except Exception:
        return False
    F, N, nIF = mesh.n_faces, mesh.n_cells, mesh.n_internal_faces
    pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
    fptr, fpts = np.ascontiguousarray(mesh.face_ptr, dtype=np.int32), np.ascontiguousarray(mesh.face_pts, dtype=np.int32)
    own, nei = np.ascontiguousarray(mesh.owner, dtype=np.int32), np.ascontiguousarray(mesh.neighbour, dtype=np.int32)
    self.Sf, self.Cf, self.C, self.V, self.w = np.empty((F, 3)), np.empty((F, 3)), np.empty((N, 3)), np.empty(N), np.empty(nIF)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = L.das_mesh_metrics(pts.shape[0], pts.ctypes.data_as(dp), F, nIF, N, fptr.ctypes.data_as(ip), fpts.ctypes.data_as(ip), own.ctypes.data_as(ip),
                            nei.ctypes.data_as(ip), self.Sf.ctypes.data_as(dp), self.Cf.ctypes.data_as(dp), self.C.ctypes.data_as(dp), self.V.ctypes.data_as(dp),
                            self.w.ctypes.data_as(dp))
    return rc == 0


_InputGeometry._native = _native_metrics


def wall_distance(mesh: PolyMesh, cell_centres, face_centres, face_areas) -> np.ndarray:
    """Frozen wall distance y (reference: yWall from meshWaveFrozen, DASolver.C:4433-4482).

    Input data, not hot path: nearest wall-face centre, distance projected on that face's
    normal (exact for flat walls), floored at 1e-12.
    """
    from scipy.spatial import cKDTree

    idx = []
    for p in mesh.patches:
        if p.type == "wall":
            idx.append(np.arange(p.start, p.start + p.size))
    if not idx:
        return np.full(mesh.n_cells, 1.0)
    idx = np.concatenate(idx)
    tree = cKDTree(face_centres[idx])
    _, nn = tree.query(cell_centres, workers=-1)
    f = idx[nn]
    n = face_areas[f] / np.linalg.norm(face_areas[f], axis=1)[:, None]
    d = np.abs(np.einsum("ij,ij->i", cell_centres - face_centres[f], n))
    return np.maximum(d, 1e-12)


def wall_distance_exact(mesh: PolyMesh, cell_centres, face_centres, face_areas, k=6) -> np.ndarray:
    """True nearest-wall distance for ARBITRARY cases (reference: yWall from meshWave, DASolver.C:4433-4482): Euclidean
    distance from every cell centre to the nearest point of the wall-face polygons (the k wall faces with the nearest
    centres are tested exactly: inside the polygon -> normal distance, else distance to its nearest edge).  The
    synthetic channel generators keep `wall_distance` (walls span the whole domain there, the two coincide)."""
    from scipy.spatial import cKDTree

    idx = [np.arange(p.start, p.start + p.size) for p in mesh.patches if p.type == "wall"]
    if not idx:
        return np.full(mesh.n_cells, 1.0)
    idx = np.concatenate(idx)
    k = int(min(k, idx.size))
    _, nn = cKDTree(face_centres[idx]).query(cell_centres, k=k, workers=-1)
    nn = nn.reshape(len(cell_centres), k)
    best = np.full(len(cell_centres), np.inf)
    pts = mesh.points
    for j in range(k):
        f = idx[nn[:, j]]
        nv = mesh.face_ptr[f + 1] - mesh.face_ptr[f]
        for cnt in np.unique(nv):
            sel = np.nonzero(nv == cnt)[0]
            fv = mesh.face_pts[mesh.face_ptr[f[sel]][:, None] + np.arange(cnt)[None, :]]  # (m, cnt) vertex ids
            V = pts[fv]                                                                   # (m, cnt, 3)
            P = cell_centres[sel]
            nrm = face_areas[f[sel]] / np.linalg.norm(face_areas[f[sel]], axis=1)[:, None]
            dn = np.einsum("ij,ij->i", P - face_centres[f[sel]], nrm)
            foot = P - dn[:, None] * nrm
            inside = np.ones(sel.size, bool)
            dedge = np.full(sel.size, np.inf)
            for a in range(cnt):
                A, B = V[:, a], V[:, (a + 1) % cnt]
                e = B - A
                # the foot point is inside a convex polygon iff it lies left of every edge (w.r.t. the face normal)
                inside &= np.einsum("ij,ij->i", np.cross(e, foot - A), nrm) >= -1e-14
                t = np.clip(np.einsum("ij,ij->i", P - A, e) / np.maximum(np.einsum("ij,ij->i", e, e), 1e-300), 0.0, 1.0)
                dedge = np.minimum(dedge, np.linalg.norm(P - (A + t[:, None] * e), axis=1))
            d = np.where(inside, np.abs(dn), dedge)
            best[sel] = np.minimum(best[sel], d)
    return np.maximum(best, 1e-12)


def n_states(case: FoamCase) -> int:
    m = case.mesh
    if case.solver_name == "DAScalarTransportFoam":
        return m.n_cells
    if case.solver_name == "DASimpleFoam":
        return (6 if case.has_T else 5) * m.n_cells + m.n_faces
    if case.solver_name in ("DARhoSimpleFoam", "DATurboFoam"):
        return 6 * m.n_cells + m.n_faces
    raise ValueError(case.solver_name)


def channel_case(
    nx=7,
    ny=7,
    nz=7,
    lengths=(1.0, 0.2, 0.1),
    U0=10.0,
    nu=1.5e-5,
    nuTilda0=4.5e-5,
    bump=0.1,
    skew=0.15,
    grading_y=1.0,
    wall_function=False,
    side_walls=False,
    seed=0,
    perturb=0.02,
    x_range=None,
) -> FoamCase:
    """DASimpleFoam + SA channel (7x7x7 = 343 cells mirrors the reference's
    ConvergentChannel size, tests/runRegTests_DASimpleFoamForward.py:32; U0/nu/nuTilda0
    are the values of tests/runRegTests_AeroOpt.py:29-48).  The state is a smooth
    synthetic flow (not a converged primal - the primal solver is upstream of the hot
    path) with a seeded perturbation so that no term degenerates."""
    types = ("patch", "patch", "wall", "wall") + (("wall", "wall") if side_walls else ("symmetry", "symmetry"))
    mesh = hex_block(
        nx, ny, nz, lengths, bump_mapping(bump, skew, lengths[0], lengths[1]), patch_types=types, grading_y=grading_y, x_range=x_range
    )
    g = _InputGeometry(mesh)
    wall_nut = NUT_SPALDING_WALL if wall_function else NUT_LOWRE_WALL
    bcs = {
        "inlet": {
            "U": (BC_FIXED_VALUE, (U0, 0.0, 0.0)),
            "p": (BC_ZERO_GRADIENT, 0.0),
            "nuTilda": (BC_FIXED_VALUE, nuTilda0),
            "nut": (NUT_CALCULATED, 0.0),
        },
        "outlet": {
            "U": (BC_INLET_OUTLET, (0.0, 0.0, 0.0)),
            "p": (BC_FIXED_VALUE, 0.0),
            "nuTilda": (BC_INLET_OUTLET, nuTilda0),
            "nut": (NUT_CALCULATED, 0.0),
        },
    }
    for nm, tp in zip(("bottom", "top", "front", "back"), types[2:]):
        if tp == "wall":
            bcs[nm] = {
                "U": (BC_FIXED_VALUE, (0.0, 0.0, 0.0)),
                "p": (BC_ZERO_GRADIENT, 0.0),
                "nuTilda": (BC_FIXED_VALUE, 0.0),
                "nut": (wall_nut, 0.0),
            }
        else:
            bcs[nm] = {
                "U": (BC_SYMMETRY, (0.0, 0.0, 0.0)),
                "p": (BC_SYMMETRY, 0.0),
                "nuTilda": (BC_SYMMETRY, 0.0),
                "nut": (NUT_SYMMETRY, 0.0),
            }
    # artificial cut patches of a rank's extended sub-mesh (multi-GPU): everything zero-gradient; their values never
    # reach an owned residual row (ghost depth = stencil depth)
    cut = {"U": (BC_ZERO_GRADIENT, (0.0, 0.0, 0.0)), "p": (BC_ZERO_GRADIENT, 0.0), "nuTilda": (BC_ZERO_GRADIENT, 0.0), "nut": (NUT_CALCULATED, 0.0)}
    if x_range is not None:
        if x_range[0] > 0:
            bcs["inlet"] = dict(cut)
        if x_range[1] < x_range[2]:
            bcs["outlet"] = dict(cut)
    y = wall_distance(mesh, g.C, g.Cf, g.Sf)
    case = FoamCase(mesh=mesh, solver_name="DASimpleFoam", nu=nu, bcs=bcs, y_wall=y)
    case._input_geometry = (mesh, g)  # (reused by prolong_channel_state: the metrics of a 2 M-cell mesh take seconds in numpy)
    # smooth synthetic state: turbulent-like profile in wall distance + perturbation
    rng = np.random.default_rng(seed)
    N, F = mesh.n_cells, mesh.n_faces
    H = lengths[1]
    eta = np.clip(y / (0.5 * H), 0.0, 1.0)
    prof = eta ** (1.0 / 7.0)
    xh = g.C[:, 0] / lengths[0]
    U = np.zeros((N, 3))
    U[:, 0] = U0 * prof * (1.0 + 0.3 * np.sin(np.pi * xh) ** 2)
    U[:, 1] = 0.05 * U0 * prof * np.sin(2 * np.pi * xh + 0.3)
    U[:, 2] = 0.03 * U0 * prof * (0.3 + np.sin(np.pi * xh) * (g.C[:, 2] / lengths[2] - 0.3))
    U *= 1.0 + perturb * rng.standard_normal((N, 1))
    p = 0.5 * U0 * U0 * 0.2 * (1.0 - xh) * (1.0 + perturb * rng.standard_normal(N))
    nuT = nuTilda0 * (1.0 + 20.0 * eta * (1.0 - 0.5 * eta)) * (1.0 + perturb * rng.standard_normal(N))
    # face flux: linear interpolate of U dotted with Sf, boundary from BC-like values
    nIF = mesh.n_internal_faces
    own, nei = mesh.owner, mesh.neighbour
    Uf = g.w[:, None] * U[own[:nIF]] + (1 - g.w[:, None]) * U[nei]
    phi = np.zeros(F)
    phi[:nIF] = np.einsum("ij,ij->i", Uf, g.Sf[:nIF])
    for pt in mesh.patches:
        sl = slice(pt.start, pt.start + pt.size)
        if pt.name == "inlet" and bcs["inlet"]["U"][0] == BC_FIXED_VALUE:
            phi[sl] = U0 * g.Sf[sl, 0]
        elif pt.name in ("inlet", "outlet"):
            phi[sl] = np.einsum("ij,ij->i", U[own[sl]], g.Sf[sl])
        else:
            phi[sl] = 0.0
    phi[:nIF] *= 1.0 + perturb * rng.standard_normal(nIF)
    case.states = np.concatenate([U.ravel(), p, nuT, phi])
    return case


def _rot_x(a):
    return np.array([[1.0, 0.0, 0.0], [0.0, np.cos(a), -np.sin(a)], [0.0, np.sin(a), np.cos(a)]])


def periodic_channel_case(nx=6, ny=5, nz=5, lengths=(1.0, 0.2, 0.1), copies=1, wall_function=False, perturb=0.02, seed=0, sector=None,
                          solver_name="DASimpleFoam", mrf_omega=None, transonic=False, **kw) -> FoamCase:
    """SA channel that is periodic in its third direction: front/back are a cyclic patch pair (copies = 1).
    sector=None: translational periodicity in z.  sector=(r0, dtheta): the block is bent into an annular sector about
    the x axis (y -> radius r0 + y, z -> angle), i.e. ROTATIONAL periodicity - the cyclic pair carries the rotation
    tensors forwardT = Rx(-+dtheta), vector fields are periodic in their cylindrical components.
    solver_name: DASimpleFoam, or DARhoSimpleFoam / DATurboFoam (compressible states like rho_channel_case; mrf_omega =
    angular velocity about the x axis with the hub (bottom) rotating, as in a compressor passage).
    copies > 1 emits the same block repeated `copies` times in the periodic direction with ORDINARY front/back patches
    and the periodic state repeated (rotated copy by copy) - the non-periodic "unrolled" mesh the tests use to check the
    cyclic implementation with the unchanged oracle (the middle copy sees its periodic images as real neighbours; the
    single periodic block coincides with the MIDDLE copy)."""
    Lx, Ly, Lz = lengths
    mid = 1  # the single periodic block sits where the middle of three unrolled copies sits
    assert copies in (1, 3)
    bump, skew, gy = kw.get("bump", 0.1), kw.get("skew", 0.15), kw.get("grading_y", 1.0)

    def mapping(ncop, first):
        base_map = bump_mapping(bump, skew, Lx, Ly, z_periodic=True)
        if sector is None:
            def f(p):
                q = base_map(p)
                q[:, 2] = q[:, 2] + first * Lz
                return q
            return f
        r0, dth = sector

        def f(p):
            q = base_map(p)
            th = (q[:, 2] / Lz + first) * dth
            r = r0 + q[:, 1]
            return np.stack([q[:, 0], r * np.cos(th), r * np.sin(th)], axis=1)
        return f

    types = ("patch", "patch", "wall", "wall", "cyclic", "cyclic")
    mesh1 = hex_block(nx, ny, nz, lengths, mapping(1, mid), patch_types=types, grading_y=gy)
    for pt in mesh1.patches:
        if pt.name in ("front", "back"):
            pt.neighbour = "back" if pt.name == "front" else "front"
            if sector is not None:  # neighbour-side vectors seen from this side: the neighbour image is rotated by -+dtheta
                pt.rotation = _rot_x(-sector[1] if pt.name == "front" else sector[1])
    g = _InputGeometry(mesh1)
    y = wall_distance(mesh1, g.C, g.Cf, g.Sf)
    base = channel_case(2, 2, 2, lengths=lengths, wall_function=wall_function, perturb=0.0)  # BC table template
    bcs = {k: dict(v) for k, v in base.bcs.items() if k not in ("front", "back")}
    cyc = {"U": (BC_CYCLIC, (0.0, 0.0, 0.0)), "p": (BC_CYCLIC, 0.0), "nuTilda": (BC_CYCLIC, 0.0), "nut": (NUT_CALCULATED, 0.0), "T": (BC_CYCLIC, 0.0)}
    bcs["front"], bcs["back"] = dict(cyc), dict(cyc)
    rng = np.random.default_rng(seed)
    N, F, nIF = mesh1.n_cells, mesh1.n_faces, mesh1.n_internal_faces
    compressible = solver_name in ("DARhoSimpleFoam", "DATurboFoam")
    U0 = kw.get("U0", 50.0 if compressible else 10.0)
    nuTilda0 = kw.get("nuTilda0", 4.5e-5)
    # box coordinates of the cell centres: (x, y = wall-normal / radial offset, zh = periodic coordinate in [0,1))
    if sector is None:
        yb, zh = g.C[:, 1], g.C[:, 2] / Lz - mid
        e2 = np.tile([0.0, 1.0, 0.0], (N, 1))
        e3 = np.tile([0.0, 0.0, 1.0], (N, 1))
    else:
        r0, dth = sector
        rr, th = np.hypot(g.C[:, 1], g.C[:, 2]), np.arctan2(g.C[:, 2], g.C[:, 1])
        yb, zh = rr - r0, th / dth - mid
        e2 = np.stack([np.zeros(N), np.cos(th), np.sin(th)], axis=1)   # e_r
        e3 = np.stack([np.zeros(N), -np.sin(th), np.cos(th)], axis=1)  # e_theta
    eta = np.clip(y / (0.5 * Ly), 0.0, 1.0)
    prof = eta ** (1.0 / 7.0)
    xh = g.C[:, 0] / Lx
    u1 = U0 * prof * (1.0 + 0.3 * np.sin(np.pi * xh) ** 2) * (1.0 + 0.05 * np.sin(2 * np.pi * zh))
    u2 = 0.05 * U0 * prof * np.sin(2 * np.pi * xh + 0.3)
    u3 = 0.04 * U0 * prof * (0.5 + np.cos(2 * np.pi * zh + 0.4) * np.sin(np.pi * xh))
    U = u1[:, None] * np.array([1.0, 0.0, 0.0]) + u2[:, None] * e2 + u3[:, None] * e3
    U *= 1.0 + perturb * rng.standard_normal((N, 1))
    p = 0.5 * U0 * U0 * 0.2 * (1.0 - xh) * (1.0 + 0.1 * np.sin(2 * np.pi * zh)) * (1.0 + perturb * rng.standard_normal(N))
    nuT = nuTilda0 * (1.0 + 20.0 * eta * (1.0 - 0.5 * eta)) * (1.0 + perturb * rng.standard_normal(N))
    own, nei = mesh1.owner, mesh1.neighbour
    sl = {pt.name: slice(pt.start, pt.start + pt.size) for pt in mesh1.patches}
    phi = np.zeros(F)
    phi[:nIF] = np.einsum("ij,ij->i", g.w[:, None] * U[own[:nIF]] + (1 - g.w[:, None]) * U[nei], g.Sf[:nIF])
    phi[:nIF] *= 1.0 + perturb * rng.standard_normal(nIF)
    phi[sl["inlet"]] = U0 * g.Sf[sl["inlet"], 0]
    phi[sl["outlet"]] = np.einsum("ij,ij->i", U[own[sl["outlet"]]], g.Sf[sl["outlet"]])
    # periodic-consistent flux through the pair: phi_back(i,j) = -phi_front(i,j) (face k of front pairs with face k of back)
    phi[sl["back"]] = np.einsum("ij,ij->i", U[own[sl["back"]]], g.Sf[sl["back"]]) * (1.0 + perturb * rng.standard_normal(sl["back"].stop - sl["back"].start))
    phi[sl["front"]] = -phi[sl["back"]]
    case = FoamCase(mesh=mesh1, solver_name=solver_name, nu=base.nu, bcs=bcs, y_wall=y)
    if compressible:
        p0, T0 = kw.get("p0", 101325.0), kw.get("T0", 300.0)
        bcs["inlet"]["T"] = (BC_FIXED_VALUE, T0)
        bcs["outlet"]["T"] = (BC_INLET_OUTLET, T0)
        bcs["outlet"]["p"] = (BC_FIXED_VALUE, p0)
        for nm in ("bottom", "top"):
            bcs[nm]["T"] = (BC_ZERO_GRADIENT, 0.0)
        case.relax = {"U": 0.7, "nuTilda": 0.7, "T": 0.9}
        rho0 = p0 / (8314.47 / case.thermo["molWeight"] * T0)
        T = T0 * (1.0 + 0.01 * np.sin(np.pi * xh) * np.cos(2 * np.pi * zh))
        case.states = np.concatenate([U.ravel(), p0 + rho0 * p, T, nuT, rho0 * phi])
        case.transonic = bool(transonic)
        if mrf_omega is not None:
            case.mrf = {"omega": (float(mrf_omega), 0.0, 0.0), "origin": (0.0, 0.0, 0.0), "nonRotatingPatches": ["inlet", "outlet", "top"]}
    else:
        case.states = np.concatenate([U.ravel(), p, nuT, phi])
        if mrf_omega is not None:
            case.mrf = {"omega": (float(mrf_omega), 0.0, 0.0), "origin": (0.0, 0.0, 0.0), "nonRotatingPatches": ["inlet", "outlet", "top"]}
    if copies == 1:
        return case
    # ---- unrolled copies: ordinary patches at the two ends, periodic repetition of geometry and state
    import copy as _copy

    meshC = hex_block(nx, ny, nz * copies, (Lx, Ly, Lz * copies), mapping(copies, 0), patch_types=("patch", "patch", "wall", "wall", "patch", "patch"),
                      grading_y=gy)
    bcsC = {k: dict(v) for k, v in bcs.items()}
    zg = {"U": (BC_ZERO_GRADIENT, (0.0, 0.0, 0.0)), "p": (BC_ZERO_GRADIENT, 0.0), "nuTilda": (BC_ZERO_GRADIENT, 0.0), "nut": (NUT_CALCULATED, 0.0),
          "T": (BC_ZERO_GRADIENT, 0.0)}
    bcsC["front"], bcsC["back"] = dict(zg), dict(zg)
    caseC = _copy.copy(case)
    caseC.mesh, caseC.bcs, caseC.y_wall = meshC, bcsC, np.tile(y, copies)
    caseC.states = unroll_periodic_vector(mesh1, meshC, case.states, copies, dtheta=None if sector is None else sector[1])
    return caseC


def unroll_periodic_vector(mesh1: PolyMesh, meshC: PolyMesh, vec, copies, dtheta=None):
    """A state-like vector of the periodic block `mesh1` ([U|p|(T)|nuTilda|phi], front/back fluxes periodic-consistent:
    phi_front = -phi_back) repeated onto the `copies`-fold unrolled block `meshC` (mesh1 = its middle copy).  dtheta:
    sector angle of a rotationally periodic block - the vector block U of copy m is rotated by (m - middle) dtheta
    about the x axis."""
    N, F, nIF = mesh1.n_cells, mesh1.n_faces, mesh1.n_internal_faces
    NC, FC, nIFC = meshC.n_cells, meshC.n_faces, meshC.n_internal_faces
    own, nei = mesh1.owner, mesh1.neighbour
    sl = {pt.name: slice(pt.start, pt.start + pt.size) for pt in mesh1.patches}
    phi = vec[-F:]
    nsc = (vec.size - 3 * N - F) // N
    cell1 = np.arange(NC) % N  # cell (i,j,k) of the unrolled block <-> (i,j,k mod nz): cid = i + nx (j + ny k)
    copy_of = np.arange(NC) // N
    pair = {(int(own[f]), int(nei[f])): f for f in range(nIF)}
    back_of = {int(own[f]): f for f in range(sl["back"].start, sl["back"].stop)}
    front_of = {int(own[f]): f for f in range(sl["front"].start, sl["front"].stop)}
    phiC = np.zeros(FC, dtype=vec.dtype)
    for f in range(nIFC):
        a, b = int(meshC.owner[f]), int(meshC.neighbour[f])
        if copy_of[a] == copy_of[b]:
            phiC[f] = phi[pair[(int(cell1[a]), int(cell1[b]))]]
        else:  # interface between two copies = the back face of the lower cell (same orientation +z)
            phiC[f] = phi[back_of[int(cell1[a])]]
    slC = {pt.name: slice(pt.start, pt.start + pt.size) for pt in meshC.patches}
    for nm in ("inlet", "outlet", "bottom", "top"):
        by_cell = {int(own[f]): f for f in range(sl[nm].start, sl[nm].stop)}
        for f in range(slC[nm].start, slC[nm].stop):
            phiC[f] = phi[by_cell[int(cell1[meshC.owner[f]])]]
    for f in range(slC["front"].start, slC["front"].stop):
        phiC[f] = phi[front_of[int(cell1[meshC.owner[f]])]]
    for f in range(slC["back"].start, slC["back"].stop):
        phiC[f] = phi[back_of[int(cell1[meshC.owner[f]])]]
    Ublk = vec[: 3 * N].reshape(N, 3)
    if dtheta is not None:
        Uall = np.concatenate([Ublk @ _rot_x((m - copies // 2) * dtheta).T for m in range(copies)])
    else:
        Uall = np.tile(Ublk, (copies, 1))
    parts = [Uall.ravel()] + [np.tile(vec[(3 + b) * N : (4 + b) * N], copies) for b in range(nsc)] + [phiC]
    return np.concatenate(parts)


def renumber_case(case: FoamCase, seed=0) -> FoamCase:
    """Randomly renumber the cells (and therefore re-sort/re-orient the internal faces to keep OpenFOAM's
    upper-triangular order) - produces a genuinely unstructured numbering of the same mesh and state."""
    import copy

    m = case.mesh
    N, F, nIF = m.n_cells, m.n_faces, m.n_internal_faces
    rng = np.random.default_rng(seed)
    new_of_old = rng.permutation(N)
    o = new_of_old[m.owner[:nIF]]
    n = new_of_old[m.neighbour]
    flip = o > n
    no, nn = np.where(flip, n, o), np.where(flip, o, n)
    order = np.lexsort((nn, no))
    nv = np.diff(m.face_ptr)
    assert np.all(nv == nv[0])
    k = int(nv[0])
    fp = m.face_pts.reshape(F, k).copy()
    fp[:nIF][flip] = fp[:nIF][flip][:, ::-1]
    face_perm = np.concatenate([order, np.arange(nIF, F)])  # new face -> old face
    mesh = PolyMesh(points=m.points.copy(), face_ptr=m.face_ptr.copy(), face_pts=np.ascontiguousarray(fp[face_perm].ravel()),
                    owner=np.concatenate([no[order], new_of_old[m.owner[nIF:]]]).astype(np.int32), neighbour=nn[order].astype(np.int32),
                    patches=copy.deepcopy(m.patches))
    out = copy.copy(case)
    out.mesh = mesh
    old_of_new = np.argsort(new_of_old)
    if case.y_wall is not None:
        out.y_wall = case.y_wall[old_of_new]
    if case.T_old is not None:
        out.T_old = case.T_old[old_of_new]
    sign = np.ones(F)
    sign[:nIF] = np.where(flip[order], -1.0, 1.0)
    if case.phi is not None:
        out.phi = case.phi[face_perm] * sign
    W = case.states
    if case.solver_name == "DASimpleFoam":
        U = W[: 3 * N].reshape(N, 3)[old_of_new]
        out.states = np.concatenate([U.ravel(), W[3 * N : 4 * N][old_of_new], W[4 * N : 5 * N][old_of_new], W[5 * N :][face_perm] * sign])
    else:
        out.states = W[old_of_new]
    return out


def _logical_centres(nx, ny, nz, i0=0, i1=None, nx_global=None):
    """logical (xi,eta,zeta) in [0,1]^3 of the cell centres of an index sub-block (x range [i0,i1) of nx_global)."""
    nxg = nx if nx_global is None else nx_global
    i1 = nx if i1 is None else i1
    return (np.arange(i0, i1) + 0.5) / nxg, (np.arange(ny) + 0.5) / ny, (np.arange(nz) + 0.5) / nz


def prolong_channel_state(case: FoamCase, dims, coarse, i0=0, nx_global=None):
    """Replace the synthetic state of a channel case by the prolongation (tri-linear in logical block coordinates)
    of a converged coarse-mesh primal solution `coarse` = dict(dims=(cx,cy,cz), W=...) of the SAME channel geometry
    (fixture dafoam_amd/data/channel_primal_coarse.npz, produced with the oracle's SIMPLE solver by
    tests/golden/make_primal_fixture.py).  The face flux is rebuilt as interp(U).Sf (boundary: U_b.Sf).
    A nearly-converged state is what the adjoint is linearised about in practice (the primal solve is upstream of
    the hot path); Jacobians of arbitrary synthetic fields are not representative (unstable ILU pivots)."""
    from scipy.interpolate import RegularGridInterpolator

    nx, ny, nz = dims
    cx, cy, cz = [int(v) for v in coarse["dims"]]
    Wc = np.asarray(coarse["W"])
    Nc = cx * cy * cz
    xs, ys, zs = _logical_centres(cx, cy, cz)
    xf, yf, zf = _logical_centres(nx, ny, nz, i0, i0 + nx, nx_global or nx)
    X, Y, Z = np.meshgrid(xf, yf, zf, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)

    def field(v):
        a = v.reshape(cz, cy, cx).transpose(2, 1, 0)
        f = RegularGridInterpolator((xs, ys, zs), a, bounds_error=False, fill_value=None)
        return f(pts).reshape(nx, ny, nz).transpose(2, 1, 0).ravel()

    U = np.stack([field(Wc[k : 3 * Nc : 3]) for k in range(3)], 1)
    p = field(Wc[3 * Nc : 4 * Nc])
    nt = np.maximum(field(Wc[4 * Nc : 5 * Nc]), 1e-12)
    mesh = case.mesh
    cached = getattr(case, "_input_geometry", None)
    g = cached[1] if cached is not None and cached[0] is mesh else _InputGeometry(mesh)
    case._input_geometry = None  # one use: the case may be copied / pickled afterwards
    nIF = mesh.n_internal_faces
    own, nei = mesh.owner, mesh.neighbour
    Uf = g.w[:, None] * U[own[:nIF]] + (1 - g.w[:, None]) * U[nei]
    phi = np.zeros(mesh.n_faces)
    phi[:nIF] = np.einsum("ij,ij->i", Uf, g.Sf[:nIF])
    for pt in mesh.patches:
        sl = slice(pt.start, pt.start + pt.size)
        code, val = case.bcs[pt.name]["U"]
        if code == BC_FIXED_VALUE:
            phi[sl] = g.Sf[sl] @ np.asarray(val, dtype=float)
        elif code == BC_SYMMETRY:
            phi[sl] = 0.0
        else:  # zeroGradient / inletOutlet: extrapolated cell velocity (outflow)
            phi[sl] = np.einsum("ij,ij->i", U[own[sl]], g.Sf[sl])
    case.states = np.concatenate([U.ravel(), p, nt, phi])
    return case


def load_coarse_primal():
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "channel_primal_coarse.npz"))
    return {"dims": tuple(int(v) for v in z["dims"]), "W": z["W"], "lengths": tuple(z["lengths"]), "grading_y": float(z["grading_y"])}


def bench_channel_case(nx, ny, nz, wall_function=False):
    """DASimpleFoam+SA channel at bench scale with a nearly-converged state (see prolong_channel_state)."""
    c = load_coarse_primal()
    case = channel_case(nx, ny, nz, lengths=c["lengths"], grading_y=c["grading_y"], wall_function=wall_function, perturb=0.0)
    return prolong_channel_state(case, (nx, ny, nz), c)


def rho_channel_case(nx=7, ny=7, nz=7, lengths=(1.0, 0.2, 0.1), U0=50.0, p0=101325.0, T0=300.0, nuTilda0=4.5e-5, wall_function=False,
                     grading_y=1.0, bump=0.1, skew=0.15, perturb=0.0, seed=0) -> FoamCase:
    """DARhoSimpleFoam + SA channel (subsonic, perfect gas): states [U | p | T | nuTilda | phi(mass flux)].
    Free-stream values follow the reference's compressible tests (p 101325, T 300; tests/runRegTests_DARhoSimpleFoam*.py)."""
    base = channel_case(nx, ny, nz, lengths=lengths, U0=U0, nuTilda0=nuTilda0, wall_function=wall_function, grading_y=grading_y, bump=bump,
                        skew=skew, perturb=perturb, seed=seed)
    mesh = base.mesh
    N, F = mesh.n_cells, mesh.n_faces
    bcs = base.bcs
    bcs["inlet"]["T"] = (BC_FIXED_VALUE, T0)
    bcs["outlet"]["T"] = (BC_INLET_OUTLET, T0)
    bcs["outlet"]["p"] = (BC_FIXED_VALUE, p0)
    for nm in ("bottom", "top", "front", "back"):
        bcs[nm]["T"] = (BC_SYMMETRY, 0.0) if bcs[nm]["U"][0] == BC_SYMMETRY else (BC_ZERO_GRADIENT, 0.0)
    case = FoamCase(mesh=mesh, solver_name="DARhoSimpleFoam", nu=base.nu, bcs=bcs, y_wall=base.y_wall)
    case.relax = {"U": 0.7, "nuTilda": 0.7, "T": 0.9}
    W = base.states
    U = W[: 3 * N].reshape(N, 3)
    pk = W[3 * N : 4 * N]  # kinematic pressure perturbation of the incompressible synthetic field
    nuT = W[4 * N : 5 * N]
    phiv = W[5 * N :]
    R = 8314.47 / case.thermo["molWeight"]
    rho0 = p0 / (R * T0)
    g = _InputGeometry(mesh)
    p = p0 + rho0 * pk
    T = T0 * (1.0 + 0.01 * np.sin(np.pi * g.C[:, 0] / lengths[0]) * np.cos(np.pi * g.C[:, 1] / lengths[1]))
    case.states = np.concatenate([U.ravel(), p, T, nuT, rho0 * phiv])
    return case


def prolong_rho_channel_state(case: FoamCase, dims, coarse):
    """Compressible twin of prolong_channel_state: the cell fields [U | p | T | nuTilda] of a converged coarse DARhoSimpleFoam channel
    solution `coarse` = dict(dims, W) of the SAME geometry interpolated tri-linearly in logical block coordinates; the mass flux is rebuilt
    as interpolate(rho) interpolate(U).Sf with rho = p / (R T) (grid sequencing of the compressible primal, round 6)."""
    from scipy.interpolate import RegularGridInterpolator

    nx, ny, nz = dims
    cx, cy, cz = [int(v) for v in coarse["dims"]]
    Wc = np.asarray(coarse["W"])
    Nc = cx * cy * cz
    xs, ys, zs = _logical_centres(cx, cy, cz)
    xf, yf, zf = _logical_centres(nx, ny, nz)
    X, Y, Z = np.meshgrid(xf, yf, zf, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)

    def field(v):
        a = v.reshape(cz, cy, cx).transpose(2, 1, 0)
        f = RegularGridInterpolator((xs, ys, zs), a, bounds_error=False, fill_value=None)
        return f(pts).reshape(nx, ny, nz).transpose(2, 1, 0).ravel()

    U = np.stack([field(Wc[k : 3 * Nc : 3]) for k in range(3)], 1)
    p = field(Wc[3 * Nc : 4 * Nc])
    T = field(Wc[4 * Nc : 5 * Nc])
    nt = np.maximum(field(Wc[5 * Nc : 6 * Nc]), 1e-12)
    mesh = case.mesh
    g = _InputGeometry(mesh)
    R = 8314.47 / case.thermo["molWeight"]
    rho = p / (R * T)
    nIF = mesh.n_internal_faces
    own, nei = mesh.owner, mesh.neighbour
    w = g.w[:, None]
    Uf = w * U[own[:nIF]] + (1 - w) * U[nei]
    rhof = g.w * rho[own[:nIF]] + (1 - g.w) * rho[nei]
    phi = np.zeros(mesh.n_faces)
    phi[:nIF] = rhof * np.einsum("ij,ij->i", Uf, g.Sf[:nIF])
    for pt in mesh.patches:
        sl = slice(pt.start, pt.start + pt.size)
        code, val = case.bcs[pt.name]["U"]
        if code == BC_FIXED_VALUE:
            phi[sl] = rho[own[sl]] * (g.Sf[sl] @ np.asarray(val, dtype=float))
        elif code == BC_SYMMETRY:
            phi[sl] = 0.0
        else:
            phi[sl] = rho[own[sl]] * np.einsum("ij,ij->i", U[own[sl]], g.Sf[sl])
    case.states = np.concatenate([U.ravel(), p, T, nt, phi])
    return case


def simple_T_channel_case(nx=7, ny=7, nz=7, T0=300.0, **kw) -> FoamCase:
    """DASimpleFoam + SA with the optional passive T field (reference DAResidualSimpleFoam.C:215-235): the incompressible
    channel plus a smooth temperature field; fixedValue T at the inlet, inletOutlet at the outlet, hot bottom wall."""
    case = channel_case(nx, ny, nz, **kw)
    N, F = case.mesh.n_cells, case.mesh.n_faces
    g = _InputGeometry(case.mesh)
    lengths = kw.get("lengths", (1.0, 0.2, 0.1))
    bcs = case.bcs
    bcs["inlet"]["T"] = (BC_FIXED_VALUE, T0)
    bcs["outlet"]["T"] = (BC_INLET_OUTLET, T0)
    bcs["bottom"]["T"] = (BC_FIXED_VALUE, T0 + 20.0)
    bcs["top"]["T"] = (BC_ZERO_GRADIENT, 0.0)
    for nm in ("front", "back"):
        bcs[nm]["T"] = (BC_SYMMETRY, 0.0) if bcs[nm]["U"][0] == BC_SYMMETRY else (BC_ZERO_GRADIENT, 0.0)
    W = case.states
    T = T0 * (1.0 + 0.02 * np.sin(np.pi * g.C[:, 0] / lengths[0]) * np.cos(np.pi * g.C[:, 1] / lengths[1]))
    case.states = np.concatenate([W[: 4 * N], T, W[4 * N :]])
    case.has_T = True
    case.relax = dict(case.relax, T=1.0)
    return case


def turbo_channel_case(nx=7, ny=7, nz=7, omega=(60.0, 0.0, 0.0), origin=(0.0, -0.3, 0.0), transonic=False, mrf=True,
                       solver_name="DATurboFoam", **kw) -> FoamCase:
    """DATurboFoam (or DARhoSimpleFoam with MRF) on the compressible channel: one MRF zone covering the mesh, rotating
    about `omega` through `origin`; inlet, outlet and the top wall (a stationary shroud) are nonRotatingPatches, the
    bottom wall (hub) and the symmetry planes rotate with the zone (reference tests/runRegTests_DATurboFoam*.py,
    runUnitTests_DARhoSimpleFoamMRF.py use constant/MRFProperties of the CompressorFluid case)."""
    case = rho_channel_case(nx, ny, nz, **kw)
    case.solver_name = solver_name
    if mrf:
        case.mrf = {"omega": tuple(float(x) for x in omega), "origin": tuple(float(x) for x in origin),
                    "nonRotatingPatches": ["inlet", "outlet", "top"]}
    case.transonic = bool(transonic)
    return case


def scalar_transport_case(nx=18, ny=17, nz=16, lengths=(1.0, 0.5, 0.5), DT=0.01, deltaT=0.05, seed=0) -> FoamCase:
    """DAScalarTransportFoam box (BASELINE.json configs[0]: 18x17x16 = 4896 cells).
    phi from uniform U=(1,0,0); T = smooth blob advected 'a few steps' (T_old shifted)."""
    mesh = hex_block(
        nx, ny, nz, lengths, bump_mapping(0.05, 0.1, lengths[0], lengths[1]),
        patch_types=("patch", "patch", "wall", "wall", "wall", "wall"),
    )
    g = _InputGeometry(mesh)
    bcs = {
        "inlet": {"T": (BC_FIXED_VALUE, 1.0)},
        "outlet": {"T": (BC_ZERO_GRADIENT, 0.0)},
    }
    for nm in ("bottom", "top", "front", "back"):
        bcs[nm] = {"T": (BC_ZERO_GRADIENT, 0.0)}
    case = FoamCase(mesh=mesh, solver_name="DAScalarTransportFoam", bcs=bcs, DT=DT, deltaT=deltaT)
    rng = np.random.default_rng(seed)
    nIF = mesh.n_internal_faces
    Uc = np.zeros((mesh.n_cells, 3))
    Uc[:, 0] = 1.0
    Uc[:, 1] = 0.2 * np.sin(2 * np.pi * g.C[:, 0] / lengths[0])
    own, nei = mesh.owner, mesh.neighbour
    Uf = np.zeros((mesh.n_faces, 3))
    Uf[:nIF] = g.w[:, None] * Uc[own[:nIF]] + (1 - g.w[:, None]) * Uc[nei]
    Uf[nIF:] = Uc[own[nIF:]]
    phi = np.einsum("ij,ij->i", Uf, g.Sf)
    for pt in mesh.patches:
        if pt.type == "wall":
            phi[pt.start : pt.start + pt.size] = 0.0
    x = g.C / np.array(lengths)
    T = np.exp(-20 * ((x[:, 0] - 0.4) ** 2 + (x[:, 1] - 0.5) ** 2 + (x[:, 2] - 0.5) ** 2))
    T_old = np.exp(-20 * ((x[:, 0] - 0.35) ** 2 + (x[:, 1] - 0.5) ** 2 + (x[:, 2] - 0.5) ** 2))
    T *= 1.0 + 0.01 * rng.standard_normal(T.size)
    case.phi = phi
    case.T_old = T_old
    case.states = T.copy()
    return case


# =================================================================================================================
# NACA0012 O-grid (BASELINE configs[1]: DASimpleFoam + SA, ~200 k cells, wall-normal stretching)
# =================================================================================================================
def naca0012_xy(t):
    """Closed NACA0012 contour, chord 1: t in [0,1) runs from the trailing edge along the LOWER side to the leading edge and
    back along the upper side (closed-trailing-edge coefficient -0.1036, cosine spacing)."""
    t = np.asarray(t, dtype=float)
    s = np.where(t < 0.5, 1.0 - 2.0 * t, 2.0 * t - 1.0)          # 1 -> 0 -> 1 along the contour
    x = 0.5 * (1.0 - np.cos(np.pi * s))                            # cosine clustering at both ends
    yt = 0.6 * (0.2969 * np.sqrt(np.maximum(x, 0.0)) - 0.1260 * x - 0.3516 * x**2 + 0.2843 * x**3 - 0.1036 * x**4)
    return x, np.where(t < 0.5, -yt, yt)


def naca_normal_distribution(ny, first_cell=2.0e-5, radius=20.0):
    """Wall-normal point distribution of the O-grid: geometric growth from `first_cell` to `radius` over ny cells."""
    lo, hi = 1.0 + 1e-9, 2.0
    f = lambda r: first_cell * (r**ny - 1.0) / (r - 1.0) - radius  # noqa: E731
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if f(mid) < 0 else (lo, mid)
    r = 0.5 * (lo + hi)
    return first_cell * (r ** np.arange(ny + 1) - 1.0) / (r - 1.0)


def naca0012_case(n_around=800, n_normal=250, nz=1, radius=20.0, span=0.1, first_cell=2.0e-5, U0=10.0, aoa_deg=2.0, nu=1.5e-5, nuTilda0=4.5e-5,
                  wall_function=False, seed=0, perturb=0.02, y_wall_section=None, sweep_deg=0.0, taper=0.0, fold_seam=False) -> FoamCase:
    """DASimpleFoam + SA around a NACA0012 (chord 1) on a single-block O-grid of n_around x n_normal x nz hexahedra, extruded
    `span` in z with symmetry front/back (the reference's 2-D airfoil cases use one cell and `empty`/symmetry sides,
    tests/runRegTests_AeroOpt.py).  Wall-normal geometric stretching from `first_cell` (y+ ~ 1 at Re 6.7e5) to the far
    field at `radius` chords.  The branch cut behind the trailing edge is an ORDINARY set of internal faces (points merged),
    so the mesh needs no coupled patches.  Patches: airfoil (wall), farfield (patch: U / nuTilda inletOutlet about the
    free stream at `aoa_deg`, p fixedValue), front / back (symmetry).  States: a smooth synthetic boundary-layer-like flow
    (seeded perturbation) - the adjoint operator's conditioning does not depend on primal convergence.
    `sweep_deg` / `taper` (round 6; BASELINE.md config 3 names a swept-wing O-grid): a genuinely three-dimensional wing segment - layer k
    is the section scaled about the quarter chord to the chord 1 - taper z_k / span and shifted downstream by z_k tan(sweep); the
    transformation fades out with the distance from the wall (half at 3 chords), the far field and the two end planes stay where they
    are.  The spanwise copies of the section are then no longer identical: no degenerate spanwise modes.
    `fold_seam` (round 6): CELL NUMBERING only - the position around the section runs 0, n-1, 1, n-2, ... instead of 0, 1, ..., n-1, so that
    the two cells either side of the O-grid's seam are neighbours in the numbering too (what OpenFOAM's renumberMesh does for a user's mesh).
    In the plain numbering the last cells of a ring depend on the first cells of the same ring and the first cells of the NEXT ring on the
    last ones of this ring: every index-ordered data-flow computation (the first-fit colouring, das_color.hpp) is serialised ring after
    ring.  Geometry, patches and the state are the same mesh; `naca_ring_position` decodes the position from a cell id."""
    nx, ny = int(n_around), int(n_normal)
    d = naca_normal_distribution(ny, first_cell, radius)             # distance from the wall, d[0] = 0, d[ny] = radius
    sblend = d / d[-1]
    t = np.arange(nx) / nx
    xa, ya = naca0012_xy(t)
    ang = -2.0 * np.pi * t                                           # outer circle, same sense as the contour (lower side first)
    xo, yo = 0.5 + radius * np.cos(ang), radius * np.sin(ang)
    # normal offset near the wall blended into the circle far away: smooth, non-overlapping for a thin symmetric section
    tx, ty = np.gradient(xa, edge_order=2), np.gradient(ya, edge_order=2)
    tx[0], ty[0] = xa[1] - xa[-1], ya[1] - ya[-1]
    nrm = np.stack([ty, -tx], axis=1)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1), 1e-300)[:, None]
    # the contour runs clockwise (lower side first): the outward normal is (ty, -tx) up to sign - fix it with the circle
    sign = np.sign(np.einsum("ij,ij->i", nrm, np.stack([xo - xa, yo - ya], axis=1)))
    nrm *= sign[:, None]
    pts2 = np.zeros((ny + 1, nx, 2))
    for j in range(ny + 1):
        wgt = sblend[j] ** 0.5                                       # how much of the ray follows the circle
        px = (1.0 - wgt) * (xa + d[j] * nrm[:, 0]) + wgt * (xa + (xo - xa) * sblend[j])
        py = (1.0 - wgt) * (ya + d[j] * nrm[:, 1]) + wgt * (ya + (yo - ya) * sblend[j])
        pts2[j, :, 0], pts2[j, :, 1] = px, py
    zs = np.linspace(0.0, span, nz + 1)
    # point ids: (i, j, k) with i periodic
    def pid(i, j, k):
        return (np.asarray(i) % nx) + nx * (np.asarray(j) + (ny + 1) * np.asarray(k))

    fold = naca_fold_table(nx) if fold_seam else np.arange(nx)

    def cid(i, j, k):
        return fold[np.asarray(i) % nx] + nx * (np.asarray(j) + ny * np.asarray(k))

    points = np.zeros((nx * (ny + 1) * (nz + 1), 3))
    swept = (sweep_deg != 0.0) or (taper != 0.0)
    wfade = 1.0 / (1.0 + (sblend * radius / 3.0) ** 2)                # per ring of points: 1 at the wall, 1/2 at ~3 chords ...
    wfade = (wfade - wfade[-1]) / (1.0 - wfade[-1])                   # ... exactly 0 on the far-field ring
    tan_sw = np.tan(np.deg2rad(sweep_deg))
    for k in range(nz + 1):
        base = nx * (ny + 1) * k
        px, py = pts2[:, :, 0], pts2[:, :, 1]
        if swept:
            cr = 1.0 - taper * zs[k] / max(span, 1e-300)
            sc = 1.0 + (cr - 1.0) * wfade[:, None]
            px = 0.25 + (px - 0.25) * sc + zs[k] * tan_sw * wfade[:, None]
            py = py * sc
        points[base : base + nx * (ny + 1), 0] = px.ravel()
        points[base : base + nx * (ny + 1), 1] = py.ravel()
        points[base : base + nx * (ny + 1), 2] = zs[k]
    I, Jc, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    I, Jc, K = I.ravel(), Jc.ravel(), K.ravel()
    c = cid(I, Jc, K)
    # internal faces: +i (periodic), +j, +k, each as (cell a, cell b, quad with normal a -> b)
    qi = np.stack([pid(I + 1, Jc, K), pid(I + 1, Jc + 1, K), pid(I + 1, Jc + 1, K + 1), pid(I + 1, Jc, K + 1)], axis=1)
    fa, fb, fq = [c], [cid(I + 1, Jc, K)], [qi]
    mj = Jc < ny - 1
    fa.append(c[mj]); fb.append(cid(I[mj], Jc[mj] + 1, K[mj]))
    fq.append(np.stack([pid(I[mj], Jc[mj] + 1, K[mj]), pid(I[mj], Jc[mj] + 1, K[mj] + 1), pid(I[mj] + 1, Jc[mj] + 1, K[mj] + 1), pid(I[mj] + 1, Jc[mj] + 1, K[mj])], axis=1))
    mk = K < nz - 1
    fa.append(c[mk]); fb.append(cid(I[mk], Jc[mk], K[mk] + 1))
    fq.append(np.stack([pid(I[mk], Jc[mk], K[mk] + 1), pid(I[mk] + 1, Jc[mk], K[mk] + 1), pid(I[mk] + 1, Jc[mk] + 1, K[mk] + 1), pid(I[mk], Jc[mk] + 1, K[mk] + 1)], axis=1))
    fa, fb, fq = np.concatenate(fa), np.concatenate(fb), np.concatenate(fq)
    # the (i, j, k) -> (x, y, z) map built above is left-handed or right-handed depending on the contour sense: orient by
    # the geometry (normal must point from cell a to cell b)
    flip = fa > fb                                                   # owner must be the lower cell id (the branch cut)
    own = np.where(flip, fb, fa)
    nei = np.where(flip, fa, fb)
    o = np.lexsort((nei, own))
    own, nei, fq, flip = own[o], nei[o], fq[o], flip[o]
    fq = np.where(flip[:, None], fq[:, ::-1], fq)
    faces, owners, patches = [fq], [own], []
    start = own.size

    def add_patch(name, ptype, quads, cells):
        nonlocal start
        oo = np.argsort(cells, kind="stable")
        faces.append(quads[oo]); owners.append(cells[oo])
        patches.append(Patch(name, ptype, start, cells.size))
        start += cells.size

    II, KK = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    II, KK = II.ravel(), KK.ravel()
    add_patch("airfoil", "wall", np.stack([pid(II, 0, KK), pid(II + 1, 0, KK), pid(II + 1, 0, KK + 1), pid(II, 0, KK + 1)], axis=1), cid(II, 0, KK))
    add_patch("farfield", "patch", np.stack([pid(II, ny, KK), pid(II, ny, KK + 1), pid(II + 1, ny, KK + 1), pid(II + 1, ny, KK)], axis=1), cid(II, ny - 1, KK))
    II, JJ = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    II, JJ = II.ravel(), JJ.ravel()
    add_patch("front", "symmetry", np.stack([pid(II, JJ, 0), pid(II, JJ + 1, 0), pid(II + 1, JJ + 1, 0), pid(II + 1, JJ, 0)], axis=1), cid(II, JJ, 0))
    add_patch("back", "symmetry", np.stack([pid(II, JJ, nz), pid(II + 1, JJ, nz), pid(II + 1, JJ + 1, nz), pid(II, JJ + 1, nz)], axis=1), cid(II, JJ, nz - 1))
    allq = np.concatenate(faces).astype(np.int32)
    mesh = PolyMesh(points=points, face_ptr=(4 * np.arange(allq.shape[0] + 1)).astype(np.int32), face_pts=np.ascontiguousarray(allq.ravel()),
                    owner=np.concatenate(owners).astype(np.int32), neighbour=nei.astype(np.int32), patches=patches)
    g = _InputGeometry(mesh)
    # orientation check: the internal-face normals must point owner -> neighbour and the boundary normals outwards; the
    # (xi, eta, z) triad may be left-handed for this contour sense, in which case every quad is reversed
    nIF = mesh.n_internal_faces
    if np.einsum("ij,ij->i", g.Sf[:nIF], g.C[mesh.neighbour] - g.C[mesh.owner[:nIF]]).sum() < 0:
        mesh.face_pts = np.ascontiguousarray(allq[:, ::-1].ravel())
        g = _InputGeometry(mesh)
    assert np.all(g.V > 0) and np.all(np.einsum("ij,ij->i", g.Sf[:nIF], g.C[mesh.neighbour] - g.C[mesh.owner[:nIF]]) > 0)
    # frozen wall distance: of an extrusion it is the section's, layer by layer (the symmetry planes are no walls) - `y_wall_section`
    # (n_around * n_normal values of the one-layer case) saves the nearest-wall search over all cells of the extruded mesh
    if y_wall_section is not None:
        assert len(y_wall_section) == nx * ny  # (a section generated with the same fold_seam: the tile keeps the numbering)
        y = np.tile(np.asarray(y_wall_section, dtype=float), nz)
        if swept:  # the near field of layer k is the section scaled by its chord: so is the distance to the wall (frozen input of the SA model)
            zc = 0.5 * (zs[:-1] + zs[1:])
            wc = 0.5 * (wfade[:-1] + wfade[1:])
            y = (y.reshape(nz, ny, nx) * (1.0 + (-taper * zc / max(span, 1e-300))[:, None, None] * wc[None, :, None])).ravel()
    else:
        y = wall_distance_exact(mesh, g.C, g.Cf, g.Sf)
    a = np.deg2rad(aoa_deg)
    Uinf = np.array([U0 * np.cos(a), U0 * np.sin(a), 0.0])
    wall_nut = NUT_SPALDING_WALL if wall_function else NUT_LOWRE_WALL
    sym = {"U": (BC_SYMMETRY, (0.0, 0.0, 0.0)), "p": (BC_SYMMETRY, 0.0), "nuTilda": (BC_SYMMETRY, 0.0), "nut": (NUT_SYMMETRY, 0.0)}
    bcs = {
        "airfoil": {"U": (BC_FIXED_VALUE, (0.0, 0.0, 0.0)), "p": (BC_ZERO_GRADIENT, 0.0), "nuTilda": (BC_FIXED_VALUE, 0.0), "nut": (wall_nut, 0.0)},
        "farfield": {"U": (BC_INLET_OUTLET, tuple(Uinf)), "p": (BC_FIXED_VALUE, 0.0), "nuTilda": (BC_INLET_OUTLET, nuTilda0), "nut": (NUT_CALCULATED, 0.0)},
        "front": dict(sym), "back": dict(sym),
    }
    rng = np.random.default_rng(seed)
    N, F = mesh.n_cells, mesh.n_faces
    delta = 0.02                                                     # boundary-layer-like thickness of the synthetic profile
    prof = 1.0 - np.exp(-y / delta)
    rr = np.hypot(g.C[:, 0] - 0.25, g.C[:, 1]) + 0.1
    th = np.arctan2(g.C[:, 1], g.C[:, 0] - 0.25)
    # free stream + a doublet-like displacement that decays with the distance, damped to zero at the wall
    Ux = Uinf[0] * (1.0 - 0.05 * np.cos(2.0 * th) / (rr * rr) * 0.04)
    Uy = Uinf[1] - Uinf[0] * 0.05 * np.sin(2.0 * th) / (rr * rr) * 0.04
    U = np.stack([Ux * prof, Uy * prof, np.zeros(N)], axis=1) * (1.0 + perturb * rng.standard_normal((N, 1)))
    p = 0.5 * U0 * U0 * 0.2 * np.cos(th) / (1.0 + rr) * (1.0 + perturb * rng.standard_normal(N))
    nuT = nuTilda0 * (1.0 + 30.0 * prof * np.exp(-y / (10.0 * delta))) * (1.0 + perturb * rng.standard_normal(N))
    ownF, neiF = mesh.owner, mesh.neighbour
    phi = np.zeros(F)
    phi[:nIF] = np.einsum("ij,ij->i", g.w[:, None] * U[ownF[:nIF]] + (1 - g.w[:, None]) * U[neiF], g.Sf[:nIF]) * (1.0 + perturb * rng.standard_normal(nIF))
    sl = {pt.name: slice(pt.start, pt.start + pt.size) for pt in mesh.patches}
    phi[sl["farfield"]] = np.einsum("ij,ij->i", U[ownF[sl["farfield"]]], g.Sf[sl["farfield"]])
    case = FoamCase(mesh=mesh, solver_name="DASimpleFoam", nu=nu, bcs=bcs, y_wall=y)
    case.states = np.concatenate([U.ravel(), p, nuT, phi])
    case._input_geometry = g  # reused by extrude_naca_state / naca_fluxes_from_velocity (a second pass over 6 M faces costs seconds)
    return case


def naca_fold_table(nx):
    """fold[i] = number of ring position i in the seam-folded numbering 0, n-1, 1, n-2, ... (naca0012_case(fold_seam=True))."""
    i = np.arange(nx)
    return np.where(i < (nx + 1) // 2, 2 * i, 2 * (nx - 1 - i) + 1)


def naca_ring_position(cell_ids, nx, fold_seam=False):
    """Position around the section (0 ... nx-1, lower side first) of the cells with the given ids."""
    r = np.asarray(cell_ids) % nx
    if not fold_seam:
        return r
    inv = np.empty(nx, dtype=np.int64)
    inv[naca_fold_table(nx)] = np.arange(nx)
    return inv[r]


def naca_fluxes_from_velocity(case: FoamCase, U):
    """phi of every face from cell velocities: linear interpolation on internal faces, the boundary value of the patch
    condition on the rest (wall 0, far field: the owner cell / free stream by flow direction, symmetry 0)."""
    mesh = case.mesh
    g = getattr(case, "_input_geometry", None) or _InputGeometry(mesh)
    nIF, F = mesh.n_internal_faces, mesh.n_faces
    own, nei = mesh.owner, mesh.neighbour
    phi = np.zeros(F)
    phi[:nIF] = np.einsum("ij,ij->i", g.w[:, None] * U[own[:nIF]] + (1 - g.w[:, None]) * U[nei], g.Sf[:nIF])
    for pt in mesh.patches:
        if pt.name == "farfield":
            sl = slice(pt.start, pt.start + pt.size)
            Uinf = np.asarray(case.bcs["farfield"]["U"][1], dtype=float)
            out = np.einsum("ij,ij->i", U[own[sl]], g.Sf[sl])
            phi[sl] = np.where(out > 0.0, out, g.Sf[sl] @ Uinf)
    return phi


def prolong_naca_state(coarse_dims, coarse_states, fine_case: FoamCase, fine_dims, first_cell=2.0e-5, radius=20.0, coarse_first_cell=None, fold_seam=False):
    """Cell fields of a 2-D (one spanwise layer) NACA0012 O-grid solution interpolated to a finer O-grid of the same family
    (grid sequencing of the primal: the reference users start a fine case from `mapFields` of a coarse one).  Bilinear in
    (contour parameter, wall distance along the ray); phi is rebuilt from the interpolated velocity."""
    nxc, nyc = coarse_dims
    nxf, nyf = fine_dims
    Nc, Nf = nxc * nyc, nxf * nyf
    assert fine_case.mesh.n_cells == Nf, "prolong_naca_state works on one spanwise layer"
    dc = naca_normal_distribution(nyc, first_cell if coarse_first_cell is None else coarse_first_cell, radius)
    df = naca_normal_distribution(nyf, first_cell, radius)
    # logarithmic wall-distance coordinate of the cell centres (geometric stretching -> uniform in the index)
    sc, sf = np.log(0.5 * (dc[1:] + dc[:-1])), np.log(0.5 * (df[1:] + df[:-1]))
    jf = np.interp(sf, sc, np.arange(nyc))                              # fractional coarse j of every fine j (clamped)
    j0 = np.clip(np.floor(jf).astype(int), 0, nyc - 2) if nyc > 1 else np.zeros(nyf, int)
    wj = np.clip(jf - j0, 0.0, 1.0)
    tf = (np.arange(nxf) + 0.5) / nxf * nxc - 0.5                        # fractional coarse i (periodic)
    i0 = np.floor(tf).astype(int)
    wi = tf - i0
    i0m, i1m = i0 % nxc, (i0 + 1) % nxc
    j1 = np.minimum(j0 + 1, nyc - 1)

    foldc = naca_fold_table(nxc) if fold_seam else np.arange(nxc)      # ring position -> column of the coarse arrays
    foldf = naca_fold_table(nxf) if fold_seam else np.arange(nxf)

    def interp(fc):                                                     # fc[(nyc, nxc)] -> (nyf, nxf), both in their cell numbering
        fc = fc[:, foldc]                                               # columns in ring order
        a = fc[j0][:, i0m] * (1 - wi)[None, :] + fc[j0][:, i1m] * wi[None, :]
        b = fc[j1][:, i0m] * (1 - wi)[None, :] + fc[j1][:, i1m] * wi[None, :]
        out = np.empty((nyf, nxf))
        out[:, foldf] = a * (1 - wj)[:, None] + b * wj[:, None]
        return out

    Wc = np.asarray(coarse_states)
    Uc = Wc[: 3 * Nc].reshape(nyc, nxc, 3)
    U = np.stack([interp(Uc[:, :, k]) for k in range(3)], axis=2).reshape(Nf, 3)
    U[:, 2] = 0.0
    p = interp(Wc[3 * Nc : 4 * Nc].reshape(nyc, nxc)).ravel()
    nuT = np.maximum(interp(Wc[4 * Nc : 5 * Nc].reshape(nyc, nxc)).ravel(), 1e-14)
    phi = naca_fluxes_from_velocity(fine_case, U)
    return np.concatenate([U.ravel(), p, nuT, phi])


def extrude_naca_state(case2d: FoamCase, states2d, case3d: FoamCase, dims):
    """A one-layer solution copied to every spanwise layer of the extruded mesh (same n_around x n_normal): with symmetry planes
    front and back the 2-D solution IS the solution of the extruded case - cell fields repeat, in-plane face fluxes scale with
    the layer thickness, spanwise fluxes vanish."""
    nx, ny, nz = dims
    m2, m3 = case2d.mesh, case3d.mesh
    N2, N3 = m2.n_cells, m3.n_cells
    assert N2 == nx * ny and N3 == nx * ny * nz
    W2 = np.asarray(states2d)
    U = np.tile(W2[: 3 * N2].reshape(N2, 3), (nz, 1))
    p, nuT = np.tile(W2[3 * N2 : 4 * N2], nz), np.tile(W2[4 * N2 : 5 * N2], nz)
    phi2 = W2[5 * N2 :]
    g2 = getattr(case2d, "_input_geometry", None) or _InputGeometry(m2)
    g3 = getattr(case3d, "_input_geometry", None) or _InputGeometry(m3)
    a2, a3 = np.linalg.norm(g2.Sf, axis=1), np.linalg.norm(g3.Sf, axis=1)
    nIF2, nIF3 = m2.n_internal_faces, m3.n_internal_faces
    phi3 = np.zeros(m3.n_faces)
    # internal faces: in-plane faces of layer k pair (own % N2, nei % N2) within the same layer
    key2 = m2.owner[:nIF2].astype(np.int64) * N2 + m2.neighbour.astype(np.int64)
    o2 = np.argsort(key2)
    own3, nei3 = m3.owner[:nIF3].astype(np.int64), m3.neighbour.astype(np.int64)
    same = (own3 // N2) == (nei3 // N2)
    key3 = (own3[same] % N2) * N2 + (nei3[same] % N2)
    pos = np.searchsorted(key2[o2], key3)
    assert np.all(key2[o2][pos] == key3)
    f2 = o2[pos]
    idx3 = np.nonzero(same)[0]
    phi3[idx3] = phi2[f2] * a3[idx3] / a2[f2]
    sl2 = {pt.name: pt for pt in m2.patches}
    for pt in m3.patches:
        if pt.name in ("airfoil", "farfield"):
            q = sl2[pt.name]
            cell2face = np.full(N2, -1, dtype=np.int64)
            cell2face[m2.owner[q.start : q.start + q.size]] = np.arange(q.start, q.start + q.size)
            f3 = np.arange(pt.start, pt.start + pt.size)
            f2b = cell2face[m3.owner[f3] % N2]
            phi3[f3] = phi2[f2b] * a3[f3] / a2[f2b]
    return np.concatenate([U.ravel(), p, nuT, phi3])
