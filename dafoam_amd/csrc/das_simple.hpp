// SIMPLE sweeps of DASimpleFoam + Spalart-Allmaras on the device.
//
// Reference: DASimpleFoam::solvePrimal (src/adjoint/DASolver/DASimpleFoam/DASimpleFoam.C:123-185) = per iteration UEqnSimple.H (relaxed
// momentum predictor: solve(UEqn == -grad(p))), pEqnSimple.H (rAU, HbyA, constrainHbyA, the pressure equation with
// nNonOrthogonalCorrectors 1, phi = phiHbyA - pEqn.flux(), explicit p relaxation, U = HbyA - rAU grad(p)) and the SA transport solve +
// correctNut (DASpalartAllmaras.C:386-405, 407-488).  The Newton-Krylov primal of this library (run_newton_primal) reaches the same fixed
// point R(W) = 0 much faster; the sweeps exist because they ARE the reference's primal, iteration by iteration: k sweeps here equal k
// sweeps of the oracle's restatement (oracle/primal.py simple_iteration) to solver tolerance (tests).
//
// The per-entity bodies below are plain templates (DAS_HD) like the residual bodies: the HIP kernels of das_device.hip wrap them, and the
// test harness (tests/hostemu) runs the same bodies in host loops.  They reuse the face records of the cell-pass split (body_fcoef /
// body_bcoef, das_kernels.hpp): fc = [cd | cdn | F0 F1 F2 | FN] per internal face, brec = 13 scalars per boundary face.
// DASimpleFoam + SA without T field, MRF and cyclic pairs.
#pragma once
#include "das_kernels.hpp"

namespace das {

// upwind convection coefficient of the cell on side `nb` of internal face f: value multiplying the OTHER cell (off) and the cell itself (dcoef)
DAS_HD void simple_upwind(double phi, bool nb, double& dcoef, double& off) {
    const double wu = phi >= 0.0 ? 1.0 : 0.0;
    if (!nb) { dcoef = wu * phi; off = (1.0 - wu) * phi; }
    else { dcoef = -((1.0 - wu) * phi); off = -(wu * phi); }
}

// ---- momentum predictor: relaxed diagonal D, boundary diagonals bd[3], total source sb[3] (explicit fluxes + boundary sources + relaxation
// term (D - D0) U), right-hand side rhs[3] = sb - V grad(p)                                      (UEqnSimple.H; fvMatrix::relax)
DAS_HD void body_simple_ueqn(int c, const DevMeshT<double>& m, const ResParams& prm, const double* W, const double* gradP, const double* fc, const double* brec,
                             double* D, double* bd, double* sb, double* rhs) {
    const long long N = m.nC, nIF = m.nIF;
    double D0 = 0.0, sumOff = 0.0, sumPhi = 0.0, vmaxs = 0.0, vmins = 0.0;
    double src[3] = {0, 0, 0}, bdiag[3] = {0, 0, 0}, bsrc[3] = {0, 0, 0};
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        const bool nb = fe < 0;
        const double phi = W[prm.offPhi * N + f];
        if (f < nIF) {
            double dcoef, off;
            simple_upwind(phi, nb, dcoef, off);
            const double sg = nb ? -1.0 : 1.0, cd = fc[f];
            sumPhi += sg * phi;
            D0 += dcoef + cd;
            sumOff += fabs(off - cd);
            for (int k = 0; k < 3; k++) src[k] += sg * fc[(2 + k) * nIF + f];
        } else {
            const double* r = brec + (long long)DAS_BREC_N * (f - nIF);
            sumPhi += phi;
            for (int k = 0; k < 3; k++) { bdiag[k] += r[k]; bsrc[k] += r[3 + k]; src[k] += r[8 + k]; }
            vmaxs += r[6];
            vmins += r[7];
        }
    }
    D0 -= sumPhi;
    const double Dr = fmax(fabs(D0 + vmaxs), sumOff) * (1.0 / prm.alphaU) - vmins;
    D[c] = Dr;
    const double V = m.cg[c].V;
    for (int k = 0; k < 3; k++) {
        bd[3LL * c + k] = bdiag[k];
        const double t = src[k] + (Dr - D0) * W[3LL * c + k] + bsrc[k];
        sb[3LL * c + k] = t;
        rhs[3LL * c + k] = t - V * gradP[3LL * c + k];
    }
}
// off-diagonal coefficients of the convection-diffusion matrices per internal face: row owner gets `up` at the neighbour, row neighbour `lo`
// at the owner; cdIdx = 0 (momentum: cd) or 1 (nuTilda: cdn)
DAS_HD void body_simple_offdiag(int f, const DevMeshT<double>& m, const ResParams& prm, const double* W, const double* fc, int cdIdx, double* up, double* lo) {
    const double phi = W[prm.offPhi * (long long)m.nC + f], cd = fc[(long long)cdIdx * m.nIF + f];
    const double wu = phi >= 0.0 ? 1.0 : 0.0;
    up[f] = (1.0 - wu) * phi - cd;
    lo[f] = -(wu * phi) - cd;
}
// y = A x for an LDU matrix given as a per-cell diagonal and per-face off-diagonals (cell-wise gather: deterministic)
DAS_HD double body_ldu_row(int c, const DevMeshT<double>& m, const double* diag, const double* up, const double* lo, const double* x) {
    double a = diag[c] * x[c];
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        if (f >= m.nIF) continue;
        a += (fe < 0 ? lo[f] : up[f]) * x[m.cf_other[s]];
    }
    return a;
}

// ---- pEqnSimple.H: rAU = 1 / UEqn.A(), HbyA = rAU UEqn.H() with the predicted velocity U* (3N, cell-major)
DAS_HD void body_simple_hbya(int c, const DevMeshT<double>& m, const double* Us, const double* D, const double* bd, const double* sb, const double* up, const double* lo,
                             double* rAU, double* HbyA) {
    double offU[3] = {0, 0, 0};
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        if (f >= m.nIF) continue;
        const double co = fe < 0 ? lo[f] : up[f];
        const int o = m.cf_other[s];
        for (int k = 0; k < 3; k++) offU[k] += co * Us[3LL * o + k];
    }
    const double avgb = (bd[3LL * c] + bd[3LL * c + 1] + bd[3LL * c + 2]) * (1.0 / 3.0), rV = 1.0 / m.cg[c].V;
    const double rA = 1.0 / ((D[c] + avgb) * rV);
    rAU[c] = rA;
    for (int k = 0; k < 3; k++) HbyA[3LL * c + k] = rA * (((avgb - bd[3LL * c + k]) * Us[3LL * c + k] - offU[k] + sb[3LL * c + k]) * rV);
}
// per face: phiHbyA, the laplacian(rAU, p) coefficients (gpf = interpolate(rAU) |Sf|, cp = gpf nonOrthDeltaCoeff), and on boundary faces the
// p patch coefficients; Wn = the state with U replaced by U* (patch values follow the current field)
DAS_HD void body_simple_pface(int f, const DevMeshT<double>& m, const ResParams& prm, const double* Wn, const double* nut, const double* rAU, const double* HbyA,
                              double* phiH, double* gpf, double* cp, double* pbc /*4 per boundary face: vic vbc gic gbc*/) {
    const long long N = m.nC;
    const FaceGeomT<double>& g = m.fg[f];
    if (f < m.nIF) {
        const int o = m.owner[f], n = m.neigh[f];
        const double wl = g.w, wn = 1.0 - g.w;
        phiH[f] = g.Sf[0] * (wl * HbyA[3LL * o] + wn * HbyA[3LL * n]) + g.Sf[1] * (wl * HbyA[3LL * o + 1] + wn * HbyA[3LL * n + 1])
                  + g.Sf[2] * (wl * HbyA[3LL * o + 2] + wn * HbyA[3LL * n + 2]);
        const double gp = (wl * rAU[o] + wn * rAU[n]) * g.magSf;
        gpf[f] = gp;
        cp[f] = gp * g.nod;
    } else {
        const int c = m.owner[f], b = f - m.nIF;
        const PatchBC& bc = m.bc[m.bpatch[b]];
        const double Uc[3] = {Wn[3LL * c], Wn[3LL * c + 1], Wn[3LL * c + 2]};
        BFace<double, double> bf;
        eval_bface<double, false>(bc, g, m.cg[c], prm, Uc, Wn[prm.offP * N + c], 0.0, Wn[prm.offN * N + c], nut[c], Wn[prm.offPhi * N + f], bf);
        double Hb[3] = {HbyA[3LL * c], HbyA[3LL * c + 1], HbyA[3LL * c + 2]};
        if (bc.U_code == DAS_BC_SYMMETRY) {
            const double hn = bf.nrm[0] * Hb[0] + bf.nrm[1] * Hb[1] + bf.nrm[2] * Hb[2];
            for (int k = 0; k < 3; k++) Hb[k] -= bf.nrm[k] * hn;
        }
        if (prm.constrainHbyA && bc.U_code == DAS_BC_FIXED_VALUE) for (int k = 0; k < 3; k++) Hb[k] = bf.U.xb[k];
        phiH[f] = g.Sf[0] * Hb[0] + g.Sf[1] * Hb[1] + g.Sf[2] * Hb[2];
        gpf[f] = rAU[c] * g.magSf;
        pbc[4LL * b] = bf.p.vic; pbc[4LL * b + 1] = bf.p.vbc; pbc[4LL * b + 2] = bf.p.gic; pbc[4LL * b + 3] = bf.p.gbc;
    }
}
// pressure equation row of cell c: diagonal dp and right-hand side (explicit non-orthogonal correction with the current grad(p))
//   laplacian(rAU, p) == div(phiHbyA):  sum_f cp (p_n - p_c) + boundary = div(phiHbyA) - div(correction)
DAS_HD void body_simple_peqn(int c, const DevMeshT<double>& m, const double* phiH, const double* gpf, const double* cp, const double* pbc, const double* gradP,
                             double* dp, double* rhs) {
    double d = 0.0, r = 0.0;
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        if (f < m.nIF) {
            const double sg = fe < 0 ? -1.0 : 1.0;
            const FaceGeomT<double>& g = m.fg[f];
            const int o = m.owner[f], n = m.neigh[f];
            const double wl = g.w, wn = 1.0 - g.w;
            const double corr = gpf[f] * (g.corr[0] * (wl * gradP[3LL * o] + wn * gradP[3LL * n]) + g.corr[1] * (wl * gradP[3LL * o + 1] + wn * gradP[3LL * n + 1])
                                          + g.corr[2] * (wl * gradP[3LL * o + 2] + wn * gradP[3LL * n + 2]));
            d -= cp[f];
            r += sg * (phiH[f] - corr);
        } else {
            const long long b = f - m.nIF;
            d += gpf[f] * pbc[4 * b + 2];
            r += phiH[f] - gpf[f] * pbc[4 * b + 3];
        }
    }
    dp[c] = d;
    rhs[c] = r;
}
// Gauss-linear gradient of a scalar cell field with patch values x_b = vic x_c + vbc
DAS_HD void body_simple_gradp(int c, const DevMeshT<double>& m, const double* p, const double* pbc, double* gradP) {
    double a[3] = {0, 0, 0};
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        const FaceGeomT<double>& g = m.fg[f];
        double pf, sg = 1.0;
        if (f < m.nIF) {
            pf = g.w * p[m.owner[f]] + (1.0 - g.w) * p[m.neigh[f]];
            sg = fe < 0 ? -1.0 : 1.0;
        } else {
            const long long b = f - m.nIF;
            pf = pbc[4 * b] * p[c] + pbc[4 * b + 1];
        }
        for (int k = 0; k < 3; k++) a[k] += sg * g.Sf[k] * pf;
    }
    const double rV = 1.0 / m.cg[c].V;
    for (int k = 0; k < 3; k++) gradP[3LL * c + k] = a[k] * rV;
}
// phi = phiHbyA - pEqn.flux()
DAS_HD void body_simple_flux(int f, const DevMeshT<double>& m, const double* pn, const double* gradP, const double* phiH, const double* gpf, const double* cp,
                             const double* pbc, double* phiOut) {
    const FaceGeomT<double>& g = m.fg[f];
    double flux;
    if (f < m.nIF) {
        const int o = m.owner[f], n = m.neigh[f];
        const double wl = g.w, wn = 1.0 - g.w;
        flux = cp[f] * (pn[n] - pn[o]) + gpf[f] * (g.corr[0] * (wl * gradP[3LL * o] + wn * gradP[3LL * n]) + g.corr[1] * (wl * gradP[3LL * o + 1] + wn * gradP[3LL * n + 1])
                                                    + g.corr[2] * (wl * gradP[3LL * o + 2] + wn * gradP[3LL * n + 2]));
    } else {
        const long long b = f - m.nIF;
        flux = gpf[f] * (pbc[4 * b + 2] * pn[m.owner[f]] + pbc[4 * b + 3]);
    }
    phiOut[f] = phiH[f] - flux;
}

// ---- SA transport (DASpalartAllmaras.C:386-405): relaxed diagonal (+ boundary diagonal) and right-hand side of the nuTilda equation at the
// updated U / phi; gradU / gradN / the face records belong to that updated state
DAS_HD void body_simple_saeqn(int c, const DevMeshT<double>& m, const ResParams& prm, const double* W, const double* gradU, const double* gradN, const double* fc,
                              const double* brec, double* diag, double* rhs) {
    const long long N = m.nC, nIF = m.nIF;
    const double nc = W[prm.offN * N + c];
    double dN = 0.0, sumOff = 0.0, sN = 0.0, bdN = 0.0, bsN = 0.0, sumPhi = 0.0, vmax = 0.0, vmin = 0.0;
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s], f = fe & 0x7fffffff;
        const bool nb = fe < 0;
        const double phi = W[prm.offPhi * N + f];
        if (f < nIF) {
            double dcoef, off;
            simple_upwind(phi, nb, dcoef, off);
            const double sg = nb ? -1.0 : 1.0, cdn = fc[nIF + f];
            sumPhi += sg * phi;
            dN += dcoef + cdn;
            sumOff += fabs(off - cdn);
            sN += sg * fc[5 * nIF + f];
        } else {
            const double* r = brec + (long long)DAS_BREC_N * (f - nIF);
            sumPhi += phi;
            bdN += r[11];
            bsN += r[12];
            vmax += fabs(r[11]);
            vmin += r[11];
        }
    }
    dN -= sumPhi;
    const CellGeomT<double>& cg = m.cg[c];
    const double y = cg.y, k2y2 = (SA_KAPPA * y) * (SA_KAPPA * y), V = cg.V;
    const double chi = nc / prm.nu;
    const double fv1 = fv1_of<double>(chi);
    const double fv2 = 1.0 - chi / (1.0 + chi * fv1);
    const double w01 = 0.5 * (gradU[9LL * c + 1] - gradU[9LL * c + 3]), w02 = 0.5 * (gradU[9LL * c + 2] - gradU[9LL * c + 6]), w12 = 0.5 * (gradU[9LL * c + 5] - gradU[9LL * c + 7]);
    const double Omega = 1.4142135623730951 * sqrt(2.0 * (w01 * w01 + w02 * w02 + w12 * w12));
    const double Stilda = fmax(Omega + fv2 * nc / k2y2, SA_CS * Omega);
    const double r = fmin(nc / (fmax(Stilda, DAS_SMALL) * k2y2), 10.0);
    const double r2 = r * r, r6 = r2 * r2 * r2;
    const double gg = r + SA_CW2 * (r6 - r);
    const double g2 = gg * gg, g6 = g2 * g2 * g2;
    const double cw36 = SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3;
    const double fw = gg * pow((1.0 + cw36) / (g6 + cw36), 1.0 / 6.0);
    const double gN2 = gradN[3LL * c] * gradN[3LL * c] + gradN[3LL * c + 1] * gradN[3LL * c + 1] + gradN[3LL * c + 2] * gradN[3LL * c + 2];
    dN += V * SA_CW1 * fw * nc / (y * y);                                    // fvm::Sp(Cw1 fw nuTilda / y^2)
    sN += V * ((SA_CB2 / SA_SIGMA) * gN2 + SA_CB1 * Stilda * nc);            // Cb2 / sigma |grad nuTilda|^2 + Cb1 Stilda nuTilda
    const double Dr = fmax(fabs(dN + vmax), sumOff) * (1.0 / prm.alphaN) - vmin;
    sN += (Dr - dN) * nc;
    diag[c] = Dr + bdN;
    rhs[c] = sN + bsN;
}

}  // namespace das
