"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the finite-volume mesh geometry the reference obtains from
OpenFOAM's fvMesh (un-vendored dependency OpenFOAM-AD v2506, reference
src/adjoint/Make/options:1-60; evidence of version DAResidual.C:247).  Every
residual in reference src/adjoint/DAResidual/*.C consumes these through
mesh_.Sf(), mesh_.magSf(), mesh_.V(), mesh_.C(), surfaceInterpolation weights,
deltaCoeffs and nonOrthCorrectionVectors.

PARITY UNPINNED: the reference cannot be built in this container and its test
meshes are not vendored; the formulas below restate
OpenFOAM's published algorithms:
  * primitiveMesh::makeFaceCentresAndAreas (triangle fan about the vertex mean),
  * primitiveMesh::makeCellCentresAndVols (pyramid decomposition about the mean
    of face centres),
  * surfaceInterpolation::makeWeights / makeDeltaCoeffs / makeNonOrthDeltaCoeffs /
    makeNonOrthCorrectionVectors (ESI conventions; fvPatch::delta() of a non-coupled patch = nf (nf . (Cf - Cn))).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class Geometry:
    def __init__(self, mesh):
        self.mesh = mesh
        pts = mesh.points
        F = mesh.n_faces
        N = mesh.n_cells
        nIF = mesh.n_internal_faces
        own = mesh.owner.astype(np.int64)
        nei = mesh.neighbour.astype(np.int64)
        self.nC, self.nF, self.nIF, self.nBF = N, F, nIF, F - nIF
        self.own, self.nei = own, nei
        self.bcell = own[nIF:]

        # ---- faces: general polygons, per-face python-free loop over vertex slots ----
        fp = mesh.face_ptr.astype(np.int64)
        nv = np.diff(fp)
        kmax = int(nv.max())
        Sf = np.zeros((F, 3))
        Cf = np.zeros((F, 3))
        # vertex mean
        fc = np.zeros((F, 3))
        for s in range(kmax):
            m = nv > s
            fc[m] += pts[mesh.face_pts[fp[:-1][m] + s]]
        fc /= nv[:, None]
        sumN = np.zeros((F, 3))
        sumA = np.zeros(F)
        sumAc = np.zeros((F, 3))
        for s in range(kmax):
            m = nv > s
            a = pts[mesh.face_pts[fp[:-1][m] + s]]
            b = pts[mesh.face_pts[fp[:-1][m] + (s + 1) % nv[m]]]
            n = np.cross(b - a, fc[m] - a)
            an = np.sqrt((n * n).sum(1))
            sumN[m] += n
            sumA[m] += an
            sumAc[m] += an[:, None] * (a + b + fc[m])
        tri = nv == 3
        Cf[:] = sumAc / (3.0 * sumA[:, None])
        Sf[:] = 0.5 * sumN
        if tri.any():
            i0 = mesh.face_pts[fp[:-1][tri]]
            i1 = mesh.face_pts[fp[:-1][tri] + 1]
            i2 = mesh.face_pts[fp[:-1][tri] + 2]
            Cf[tri] = (pts[i0] + pts[i1] + pts[i2]) / 3.0
            Sf[tri] = 0.5 * np.cross(pts[i1] - pts[i0], pts[i2] - pts[i0])
        self.Sf, self.Cf = Sf, Cf
        self.magSf = np.sqrt((Sf * Sf).sum(1))
        self.nf = Sf / self.magSf[:, None]

        # ---- cells ----
        cnt = np.bincount(own, minlength=N) + np.bincount(nei, minlength=N)
        cEst = np.zeros((N, 3))
        np.add.at(cEst, own, Cf)
        np.add.at(cEst, nei, Cf[:nIF])
        cEst /= cnt[:, None]
        pvo = (Sf * (Cf - cEst[own])).sum(1)
        pvn = (Sf[:nIF] * (cEst[nei] - Cf[:nIF])).sum(1)
        pco = 0.75 * Cf + 0.25 * cEst[own]
        pcn = 0.75 * Cf[:nIF] + 0.25 * cEst[nei]
        V3 = np.zeros(N)
        C = np.zeros((N, 3))
        np.add.at(V3, own, pvo)
        np.add.at(V3, nei, pvn)
        np.add.at(C, own, pvo[:, None] * pco)
        np.add.at(C, nei, pvn[:, None] * pcn)
        self.C = C / V3[:, None]
        self.V = V3 / 3.0

        # ---- interpolation metrics (internal faces) ----
        o, n_ = own[:nIF], nei
        sfo = np.abs((Sf[:nIF] * (Cf[:nIF] - self.C[o])).sum(1))
        sfn = np.abs((Sf[:nIF] * (self.C[n_] - Cf[:nIF])).sum(1))
        self.w = sfn / (sfo + sfn)
        d = self.C[n_] - self.C[o]
        magd = np.sqrt((d * d).sum(1))
        self.deltaCoeffs = 1.0 / magd
        nd = (self.nf[:nIF] * d).sum(1)
        self.nonOrthDeltaCoeffs = 1.0 / np.maximum(nd, 0.05 * magd)
        self.nonOrthCorr = self.nf[:nIF] - d * self.nonOrthDeltaCoeffs[:, None]
        # ---- boundary faces ----
        # fvPatch::delta() of a non-coupled patch is the patch-normal part of Cf - Cn (OpenFOAM v1712+: "Use patch-normal delta for
        # all non-coupled BCs"): deltaCoeffs = 1 / |nf . (Cf - Cn)|
        db = Cf[nIF:] - self.C[self.bcell]
        self.bDeltaCoeffs = 1.0 / np.abs((self.nf[nIF:] * db).sum(1))
        self.bSf = Sf[nIF:]
        self.bMagSf = self.magSf[nIF:]
        self.bnf = self.nf[nIF:]

        # ---- cell-cell adjacency (face neighbours), used by the stencil/connectivity oracle
        A = sp.coo_matrix((np.ones(nIF), (o, n_)), shape=(N, N))
        self.cellCells = ((A + A.T) > 0).astype(np.int8).tocsr()
        # cell-face incidence (N x F) incl. boundary faces
        rows = np.concatenate([own, nei])
        cols = np.concatenate([np.arange(F), np.arange(nIF)])
        self.cellFaces = sp.coo_matrix((np.ones(rows.size, np.int8), (rows, cols)), shape=(N, F)).tocsr()

    def patch_slices(self):
        """name -> slice into the boundary-face arrays (0-based at first boundary face)."""
        return {p.name: slice(p.start - self.nIF, p.start - self.nIF + p.size) for p in self.mesh.patches}
