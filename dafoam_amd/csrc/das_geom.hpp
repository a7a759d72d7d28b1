// fvMesh metrics from the point coordinates, one entity per call: the SAME bodies run on the host (Mesh::compute_geometry) and in
// the device kernels of the mesh-sensitivity product (das_calc_dvolcoord_product re-evaluates the metrics for every coloured point
// perturbation without leaving the GPU).
//
// Restates OpenFOAM's primitiveMesh::makeFaceCentresAndAreas / makeCellCentresAndVols and surfaceInterpolation::makeWeights /
// makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors, cyclicFvPatch::makeWeights / delta() and fvPatch::delta() - what the
// reference's updateOFMesh (pyDASolvers.pyx:297-300 -> DASolver::updateOFMesh) triggers through fvMesh::movePoints.
#pragma once
#include "das_common.hpp"
#include "das_dual.hpp"

namespace das {

struct GeomTopo {  // point / face / cell addressing the three passes read
    int nC, nF, nIF;
    const int* face_ptr;
    const int* face_pts;
    const int* owner;
    const int* neigh;
    const int* cf_ptr;
    const int* cf_face;   // face id | side << 31 (side 1: the cell is the face's neighbour)
    const int* bpatch;    // boundary face -> patch
    const int* cyc;       // boundary face -> paired face, -1 otherwise
    const PatchBC* bc;
};

template <class S>
DAS_HD void geom_cross(const S* a, const S* b, S* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// (S: the scalar of points and metrics - double, or Dual<1> when the point coordinates carry tangents)
// pass 1, per face: area vector, magnitude, centre
template <class S>
DAS_HD void geom_face(int f, const GeomTopo& t, const S* __restrict__ P, FaceGeomT<S>& g) {
    const int b = t.face_ptr[f], e = t.face_ptr[f + 1], nv = e - b;
    if (nv == 3) {
        const S *p0 = P + 3 * (long long)t.face_pts[b], *p1 = P + 3 * (long long)t.face_pts[b + 1], *p2 = P + 3 * (long long)t.face_pts[b + 2];
        S a[3], c2[3], n[3];
        for (int k = 0; k < 3; k++) { a[k] = p1[k] - p0[k]; c2[k] = p2[k] - p0[k]; g.Cf[k] = (p0[k] + p1[k] + p2[k]) / 3.0; }
        geom_cross(a, c2, n);
        for (int k = 0; k < 3; k++) g.Sf[k] = 0.5 * n[k];
    } else {
        S fc[3] = {S(0.0), S(0.0), S(0.0)};
        for (int i = b; i < e; i++) for (int k = 0; k < 3; k++) fc[k] += P[3 * (long long)t.face_pts[i] + k];
        for (int k = 0; k < 3; k++) fc[k] = fc[k] / (double)nv;
        S sumN[3] = {S(0.0), S(0.0), S(0.0)}, sumA(0.0), sumAc[3] = {S(0.0), S(0.0), S(0.0)};
        for (int i = 0; i < nv; i++) {
            const S* p = P + 3 * (long long)t.face_pts[b + i];
            const S* q = P + 3 * (long long)t.face_pts[b + (i + 1 == nv ? 0 : i + 1)];
            S u[3], v[3], n[3];
            for (int k = 0; k < 3; k++) { u[k] = q[k] - p[k]; v[k] = fc[k] - p[k]; }
            geom_cross(u, v, n);
            const S a = dsqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; k++) { sumN[k] += n[k]; sumAc[k] += a * (p[k] + q[k] + fc[k]); }
            sumA += a;
        }
        for (int k = 0; k < 3; k++) { g.Cf[k] = sumAc[k] / (3.0 * sumA); g.Sf[k] = 0.5 * sumN[k]; }
    }
    g.magSf = dsqrt(g.Sf[0] * g.Sf[0] + g.Sf[1] * g.Sf[1] + g.Sf[2] * g.Sf[2]);
}

// pass 2, per cell: centre and volume from the pyramids over the cell's faces (ascending face id, like the face loop of
// makeCellCentresAndVols); the frozen wall distance y is not touched.  Returns false for a non-positive volume.
template <class S>
DAS_HD bool geom_cell(int c, const GeomTopo& t, const FaceGeomT<S>* __restrict__ fg, CellGeomT<S>& out) {
    S cEst[3] = {S(0.0), S(0.0), S(0.0)};
    const int s0 = t.cf_ptr[c], s1 = t.cf_ptr[c + 1];
    for (int s = s0; s < s1; s++) {
        const FaceGeomT<S>& g = fg[t.cf_face[s] & 0x7fffffff];
        for (int k = 0; k < 3; k++) cEst[k] += g.Cf[k];
    }
    for (int k = 0; k < 3; k++) cEst[k] = cEst[k] / (double)(s1 - s0);
    S V3(0.0), Cs[3] = {S(0.0), S(0.0), S(0.0)};
    for (int s = s0; s < s1; s++) {
        const int fe = t.cf_face[s];
        const FaceGeomT<S>& g = fg[fe & 0x7fffffff];
        S pv(0.0);
        if (fe >= 0) for (int k = 0; k < 3; k++) pv += g.Sf[k] * (g.Cf[k] - cEst[k]);
        else for (int k = 0; k < 3; k++) pv += g.Sf[k] * (cEst[k] - g.Cf[k]);
        for (int k = 0; k < 3; k++) Cs[k] += pv * (0.75 * g.Cf[k] + 0.25 * cEst[k]);
        V3 += pv;
    }
    for (int k = 0; k < 3; k++) out.C[k] = Cs[k] / V3;
    out.V = V3 / 3.0;
    return val(out.V) > 0;
}

// pass 3, per face: interpolation weight, non-orthogonal delta coefficient and correction vector
// (g is fg[f]: pass 3 writes w / nod / corr of its own face and reads Sf / Cf / magSf of the paired face - no restrict on fg)
template <class S>
DAS_HD void geom_weights(int f, const GeomTopo& t, const CellGeomT<S>* __restrict__ cg, const FaceGeomT<S>* fg, FaceGeomT<S>& g) {
    const S* Co = cg[t.owner[f]].C;
    if (f < t.nIF) {
        const S* Cn = cg[t.neigh[f]].C;
        S so(0.0), sn(0.0), d[3], md(0.0), nd(0.0);
        for (int k = 0; k < 3; k++) {
            so += g.Sf[k] * (g.Cf[k] - Co[k]);
            sn += g.Sf[k] * (Cn[k] - g.Cf[k]);
            d[k] = Cn[k] - Co[k];
            md += d[k] * d[k];
            nd += g.Sf[k] / g.magSf * d[k];
        }
        so = dabs(so); sn = dabs(sn); md = dsqrt(md);
        g.w = sn / (so + sn);
        g.nod = 1.0 / dmax(nd, 0.05 * md);
        for (int k = 0; k < 3; k++) g.corr[k] = g.Sf[k] / g.magSf - d[k] * g.nod;
    } else if (t.cyc[f - t.nIF] >= 0) {
        // cyclicFvPatch::makeWeights / delta(): the neighbour cell is the owner of the paired face, seen at
        // Cf - Q (Cf' - C')  (its image across the pair)
        const int f2 = t.cyc[f - t.nIF];
        const FaceGeomT<S>& g2 = fg[f2];
        const S* C2 = cg[t.owner[f2]].C;
        const double* Q = t.bc[t.bpatch[f - t.nIF]].Q;  // forwardT: neighbour-side vectors -> this side
        S dOwn(0.0), dNbr(0.0), d[3], md(0.0), nd(0.0);
        const S r2[3] = {g2.Cf[0] - C2[0], g2.Cf[1] - C2[1], g2.Cf[2] - C2[2]};
        for (int k = 0; k < 3; k++) {
            dOwn += g.Sf[k] / g.magSf * (g.Cf[k] - Co[k]);
            dNbr += g2.Sf[k] / g2.magSf * r2[k];
            d[k] = (g.Cf[k] - Co[k]) - (Q[3 * k] * r2[0] + Q[3 * k + 1] * r2[1] + Q[3 * k + 2] * r2[2]);
            md += d[k] * d[k];
        }
        for (int k = 0; k < 3; k++) nd += g.Sf[k] / g.magSf * d[k];
        md = dsqrt(md);
        g.w = dNbr / (dOwn + dNbr);
        g.nod = 1.0 / dmax(nd, 0.05 * md);
        for (int k = 0; k < 3; k++) g.corr[k] = g.Sf[k] / g.magSf - d[k] * g.nod;
    } else {
        // non-coupled patch: fvPatch::delta() is the PATCH-NORMAL part of Cf - Cn (OpenFOAM v1712+), so deltaCoeffs =
        // nonOrthDeltaCoeffs = 1 / |nf . (Cf - Cn)|; identical to 1/|Cf - Cn| on orthogonal wall cells, different on sheared ones
        S nd(0.0);
        for (int k = 0; k < 3; k++) nd += g.Sf[k] / g.magSf * (g.Cf[k] - Co[k]);
        g.w = S(1.0);
        g.nod = 1.0 / dabs(nd);
        g.corr[0] = g.corr[1] = g.corr[2] = S(0.0);
    }
}

}  // namespace das
