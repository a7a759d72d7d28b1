// Residual kernel bodies R(W) for the MI355X adjoint hot path, templated on the scalar type
// (double, or Dual<K> for the coloured forward-mode Jacobian assembly).
//
// What is evaluated (reference file:line):
//   DAResidualSimpleFoam::calcResiduals        src/adjoint/DAResidual/DAResidualSimpleFoam.C:106-237
//   DASpalartAllmaras::calcResiduals/correctNut src/adjoint/DAModel/DATurbulenceModel/DASpalartAllmaras.C:124-178,215-233,407-488
//   DATurbulenceModel::divDevRhoReff            src/adjoint/DAModel/DATurbulenceModel/DATurbulenceModel.C:378-408
//   nutUSpaldingWallFunction calcNut/calcUTau   src/adjoint/DAMisc/nutUSpaldingWallFunctionDF/...DF.C:42-150
//   DAResidualScalarTransportFoam::calcResiduals src/adjoint/DAResidual/DAResidualScalarTransportFoam.C:57-84
//   residual normalisation macros               src/include/DAMacroFunctions.H:28-51
//   DAResidualRhoSimpleFoam::calcResiduals      src/adjoint/DAResidual/DAResidualRhoSimpleFoam.C:84-211   (RHO = true)
//   DAResidualTurboFoam::calcResiduals          src/adjoint/DAResidual/DAResidualTurboFoam.C:66-233       (RHO = true, prm.turbo)
//   MRF (OpenFOAM MRFZone: addCoriolis, makeRelativeRhoFlux, correctBoundaryVelocity; one zone = the mesh)
// The OpenFOAM operators behind those lines (fvm::div/laplacian, fvc::grad, fvMatrix::A/H/flux/&/relax,
// constrainHbyA, patch-field coefficients) are un-vendored; their semantics for the canonical scheme set are
// documented in DESIGN.md and restated independently by oracle/residual.py.
//
// Design (MI355X): every kernel is cell- or face-centric *gather* (no atomics, deterministic): a cell walks
// its face list, reads the neighbour cell's record through L2, and recomputes face quantities on the fly
// instead of materialising face-coefficient arrays in HBM.  Boundary conditions are re-evaluated inline from
// the owner cell's values, so no boundary-field arrays exist.  One residual evaluation = 4 launches:
//   k_grad (Gauss gradients, nut) -> k_cell (U-eqn A/H/URes + SA) -> k_face (phiHbyA, p-flux, phiRes)
//   -> k_pres (pRes).
#pragma once
#include "das_common.hpp"
#include "das_dual.hpp"

namespace das {

template <class G>
struct DevMeshT {  // G: the scalar of the metrics (double; Dual<1> in the mesh-sensitivity pass)
    int nC, nF, nIF;
    const FaceGeomT<G>* fg;
    const CellGeomT<G>* cg;
    const int* cf_ptr;
    const int* cf_face;
    const int* cf_other;
    const int* owner;
    const int* neigh;
    const int* bpatch;  // boundary face -> patch
    const PatchBC* bc;
    const int* cyc;     // boundary face -> paired face of a cyclic pair, -1 otherwise
};
typedef DevMeshT<double> DevMesh;

struct ResParams {
    double nu, alphaU, alphaN, DT, deltaT;
    int isPC, constrainHbyA;
    // weight of the explicit linearUpwindV correction: 1 in the operator residual, amd.pcUpwindBlend (default 0 = div(pc) upwind, the
    // reference's PC scheme) in the PC residual - a PC matrix between the first-order and the operator's discretisation
    double convBlend;
    int normU, normP, normN, normPhi, normT;  // 1 = residual listed in normalizeResiduals
    // state-block offsets in units of nC (DAIndex "state" ordering): SimpleFoam [U|p|nuTilda|phi] = 3,-,4,5;
    // RhoSimpleFoam [U|p|T|nuTilda|phi] = 3,4,5,6
    int offP, offT, offN, offPhi;
    // perfect-gas / hConst / const-transport thermo (reference DAResidual.C:179-293)
    double Cp, Rgas, mu, Pr, Prt;
    // transport: const (mu, alpha = mu / Pr) or sutherland (mu(T), alpha = mu * alphaFac), DAResidual.C:264-293
    int sutherland;
    double As, Ts, alphaFac;
    // DATurboFoam switches and the MRF zone (angular velocity, origin)
    int turbo, transonic, transonicPC, mrf;
    int hasT;  // DASimpleFoam with the optional passive T field (RHO = false kernels)
    int cellFaceSplit;     // amd.cellFaceSplit and the case allows it: k_fcoef + k_bcoef + k_cell2 instead of k_cell (host-side launch switch)
    int gradFaceParallel;  // amd.gradFaceParallel: the face-parallel LDS-staged gradient kernel k_grad_fp where it applies (host-side launch switch)
    double om[3], org[3];
    // DATurboFoam work array of the launch (typed by the kernel's scalar type): Teff.U per cell (3N)
    void* wTU;
    // `field` input betaFINuTilda (per cell; null = 1) and its tangent (null = 0)
    const double* betaFI;
    const double* dBetaFI;
};
#define DAS_TREF 298.15

// laminar viscosity of the compressible solvers: constant, or Sutherland's law mu = As sqrt(T) / (1 + Ts / T)
template <class T>
DAS_HD T mu_of(const ResParams& prm, const T& Tk) {
    if (!prm.sutherland) return T(prm.mu);
    return prm.As * dsqrt(Tk) / (1.0 + prm.Ts / Tk);
}

// Omega x (x - origin)
template <class G>
DAS_HD void mrf_velocity(const ResParams& prm, const G* x, G* v) {
    const G r0 = x[0] - prm.org[0], r1 = x[1] - prm.org[1], r2 = x[2] - prm.org[2];
    v[0] = prm.om[1] * r2 - prm.om[2] * r1;
    v[1] = prm.om[2] * r0 - prm.om[0] * r2;
    v[2] = prm.om[0] * r1 - prm.om[1] * r0;
}
// (Omega x (Cf - origin)) . Sf : what makeRelative subtracts per unit density
template <class G>
DAS_HD G mrf_face_flux(const ResParams& prm, const FaceGeomT<G>& g) {
    G v[3];
    mrf_velocity(prm, g.Cf, v);
    return v[0] * g.Sf[0] + v[1] * g.Sf[1] + v[2] * g.Sf[2];
}

// SA constants (reference DASpalartAllmaras.C:47-80)
#define SA_SIGMA 0.66666
#define SA_KAPPA 0.41
#define SA_CB1 0.1355
#define SA_CB2 0.622
#define SA_CW2 0.3
#define SA_CW3 2.0
#define SA_CV1 7.1
#define SA_CS 0.3
#define SA_CW1 (SA_CB1 / (SA_KAPPA * SA_KAPPA) + (1.0 + SA_CB2) / SA_SIGMA)
#define DAS_VSMALL 1e-300
#define DAS_SMALL 1e-15
#define DAS_ROOTVSMALL 1e-150

// rotational cyclic pairs: neighbour-side vectors / gradient tensors seen in this side's frame (forwardT = Q)
template <class T>
DAS_HD void rot_vec(const double* Q, T* v) {
    T a = Q[0] * v[0] + Q[1] * v[1] + Q[2] * v[2];
    T b = Q[3] * v[0] + Q[4] * v[1] + Q[5] * v[2];
    T c = Q[6] * v[0] + Q[7] * v[1] + Q[8] * v[2];
    v[0] = a; v[1] = b; v[2] = c;
}
template <class T>
DAS_HD void rot_ten(const double* Q, T* g) {  // g[3i+j] = d_i U_j  ->  Q g Q^T
    T t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = Q[3 * i] * g[j] + Q[3 * i + 1] * g[3 + j] + Q[3 * i + 2] * g[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) g[3 * i + j] = t[3 * i] * Q[3 * j] + t[3 * i + 1] * Q[3 * j + 1] + t[3 * i + 2] * Q[3 * j + 2];
}
// the rotation of the cyclic patch of boundary face f, or null (internal faces, translational pairs)
template <class G>
DAS_HD const double* cyclic_rotation(const DevMeshT<G>& m, int f) {
    if (f < m.nIF) return nullptr;
    const PatchBC& pb = m.bc[m.bpatch[f - m.nIF]];
    return pb.rot ? pb.Q : nullptr;
}

template <class T>
DAS_HD T fv1_of(const T& chi) {
    T chi3 = chi * chi * chi;
    return chi3 / (chi3 + SA_CV1 * SA_CV1 * SA_CV1);
}

// ---- patch-field coefficients: x_b = vic*x_c + vbc ; snGrad_b = gic*x_c + gbc ---------------------
template <class T, class G = double>
struct ScalarBC {
    T xb;
    G vic, gic;
    T vbc, gbc;  // carry the tangent of the patch value (dR/d(BC value) seeds)
};
template <class T, class G>
DAS_HD void bc_scalar(int code, double value, double dvalue, const G& delta, double phib, const T& xc, ScalarBC<T, G>& o) {
    double f = 0.0;
    if (code == DAS_BC_FIXED_VALUE) f = 1.0;
    else if (code == DAS_BC_INLET_OUTLET) f = phib >= 0.0 ? 0.0 : 1.0;
    const T val = MkSeed<T>::make(value, dvalue);
    o.vic = 1.0 - f;
    o.vbc = f * val;
    o.gic = -f * delta;
    o.gbc = (f * delta) * val;
    o.xb = o.vic * xc + o.vbc;
}
template <class T, class G = double>
struct VectorBC {
    T xb[3];
    G vic[3], gic[3];
    T vbc[3], gbc[3];
};
template <class T, class G, class V>
DAS_HD void bc_vector(int code, const V* value, const double* dvalue, const G& delta, double phib, const G* n, const T* Xc, VectorBC<T, G>& o) {
    if (code == DAS_BC_SYMMETRY) {
        // basicSymmetry/transformFvPatchField: x_b = X - n (n.X); snGradTransformDiag = |n|
        T nX = n[0] * Xc[0] + n[1] * Xc[1] + n[2] * Xc[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            G sd = dabs(n[k]);
            o.xb[k] = Xc[k] - n[k] * nX;
            o.vic[k] = 1.0 - sd;
            o.vbc[k] = o.xb[k] - o.vic[k] * Xc[k];
            o.gic[k] = -delta * sd;
            o.gbc[k] = (-n[k] * delta) * nX - o.gic[k] * Xc[k];
        }
        return;
    }
    double f = 0.0;
    if (code == DAS_BC_FIXED_VALUE) f = 1.0;
    else if (code == DAS_BC_INLET_OUTLET) f = phib >= 0.0 ? 0.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const T val = MkSeed<T>::make(value[k], dvalue[k]);
        o.vic[k] = 1.0 - f;
        o.vbc[k] = f * val;
        o.gic[k] = -f * delta;
        o.gbc[k] = (f * delta) * val;
        o.xb[k] = o.vic[k] * Xc[k] + o.vbc[k];
    }
}

// nutUSpaldingWallFunctionDF (reference ...DF.C:42-150): Newton iteration to the root - maxIter 1000, tolerance 1e-14 are the
// reference's defaults (...DF.C:180-181, ...DF.H:32-37; the stock OpenFOAM field stops after 10) - from the laminar seed
// (:117-118; at the root the seed drops out).  The dual part rides along: at the root one Newton step of the dual number IS
// the implicit-function derivative.
#define DAS_SPALDING_MAXITER 1000
template <class T>
DAS_HD T spalding_nut(const T& magUp, const T& magGradU, const T& y, const T& nu) {
    const double kappa = 0.41, E = 9.8;
    T ut = dsqrt(nu * magGradU);
    if (!(val(ut) > DAS_ROOTVSMALL)) return T(0.0);
    for (int it = 0; it < DAS_SPALDING_MAXITER; it++) {
        T kUu = dmin(kappa * magUp / ut, 50.0);
        T fkUu = dexp(kUu) - 1.0 - kUu * (1.0 + 0.5 * kUu);
        T f = -(ut * (y / nu)) + magUp / ut + (1.0 / E) * (fkUu - (1.0 / 6.0) * kUu * kUu * kUu);
        T df = (y / nu) + magUp / (ut * ut) + (1.0 / E) * kUu * fkUu / ut;
        T utn = ut + f / df;
        double err = fabs((val(ut) - val(utn)) / val(ut));
        ut = utn;
        if (val(ut) <= DAS_ROOTVSMALL || err <= 1e-14) break;
    }
    ut = dmax(ut, 0.0);
    return dmax(ut * ut / (magGradU + DAS_ROOTVSMALL) - nu, 0.0);
}

template <class T, class G = double>
struct BFace {
    VectorBC<T, G> U;
    ScalarBC<T, G> p, n;
    ScalarBC<T, G> Tt, he;  // compressible only
    T rho_b, nu_b, mu_b;  // compressible only (incompressible: rho_b = 1, nu_b = nu)
    T nut_b;
    G nrm[3];
};

// evaluate all patch fields of one boundary face from the owner cell's values
template <class T, bool RHO, class G>
DAS_HD void eval_bface(const PatchBC& bc, const FaceGeomT<G>& g, const CellGeomT<G>& cgc, const ResParams& prm, const T* Uc, const T& pc,
                       const T& Tc, const T& nc, const T& nut_c, double phib, BFace<T, G>& o) {
#pragma unroll
    for (int k = 0; k < 3; k++) o.nrm[k] = g.Sf[k] / g.magSf;
    if (prm.mrf && bc.mrf_included && bc.U_code == DAS_BC_FIXED_VALUE) {
        // MRFZone::correctBoundaryVelocity: fixedValue patches that rotate with the zone carry Omega x r
        G uw[3];
        const double zero[3] = {0.0, 0.0, 0.0};
        mrf_velocity(prm, g.Cf, uw);
        bc_vector<T>(bc.U_code, uw, zero, g.nod, phib, o.nrm, Uc, o.U);
    } else {
        bc_vector<T>(bc.U_code, bc.U_val, bc.dU_val, g.nod, phib, o.nrm, Uc, o.U);
    }
    bc_scalar<T>(bc.p_code, bc.p_val, bc.dp_val, g.nod, phib, pc, o.p);
    bc_scalar<T>(bc.nuTilda_code, bc.nuTilda_val, bc.dnuTilda_val, g.nod, phib, nc, o.n);
    if (RHO) {
        bc_scalar<T>(bc.T_code, bc.T_val, bc.dT_val, g.nod, phib, Tc, o.Tt);
        T hec = prm.Cp * (Tc - DAS_TREF);
        bc_scalar<T>(bc.T_code, prm.Cp * (bc.T_val - DAS_TREF), prm.Cp * bc.dT_val, g.nod, phib, hec, o.he);
        o.rho_b = o.p.xb / (prm.Rgas * o.Tt.xb);
        o.mu_b = mu_of<T>(prm, o.Tt.xb);
        o.nu_b = o.mu_b / o.rho_b;
    } else {
        o.rho_b = T(1.0);
        o.nu_b = T(prm.nu);
        if (prm.hasT) {  // passive T of DASimpleFoam: "he" is T itself
            bc_scalar<T>(bc.T_code, bc.T_val, bc.dT_val, g.nod, phib, Tc, o.Tt);
            o.he = o.Tt;
        }
    }
    if (bc.nut_code == DAS_NUT_LOWRE_WALL) o.nut_b = T(0.0);
    else if (bc.nut_code == DAS_NUT_SYMMETRY) o.nut_b = nut_c;
    else if (bc.nut_code == DAS_NUT_SPALDING_WALL) {
        T d0 = Uc[0] - o.U.xb[0], d1 = Uc[1] - o.U.xb[1], d2 = Uc[2] - o.U.xb[2];
        T magUp = dsqrt(d0 * d0 + d1 * d1 + d2 * d2);
        T magGradU = magUp * g.nod;
        const T yw = dabs((g.Cf[0] - cgc.C[0]) * o.nrm[0] + (g.Cf[1] - cgc.C[1]) * o.nrm[1] + (g.Cf[2] - cgc.C[2]) * o.nrm[2]);
        o.nut_b = spalding_nut<T>(magUp, magGradU, yw, o.nu_b);
    } else {
        o.nut_b = o.n.xb * fv1_of<T>(o.n.xb / o.nu_b);
    }
}

// q = Teff . U with Teff = muEff dev(twoSymm(grad U))  (= -devRhoReff, symmetric); g[3*i+j] = d_i U_j
template <class T>
DAS_HD void teff_dot_u(const T* g, const T& muEff, const T* U, T* q) {
    T tr3 = (2.0 / 3.0) * (g[0] + g[4] + g[8]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T acc(0.0);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T sij = g[3 * i + j] + g[3 * j + i];
            if (i == j) sij = sij - tr3;
            acc += sij * U[j];
        }
        q[i] = muEff * acc;
    }
}

// ================================================================================ k_grad
// per cell: nut = nuTilda*fv1 ; Gauss-linear gradients of U, p, nuTilda (and he = Cp (T - Tref) when RHO)
//
// One face slot s of cell c: S_f (x) (face values) added to the Gauss sums gU / gP / gN / gH.  `tile`: optional LDS copy of the cell
// states [U0 U1 U2 p nuTilda][tileN] of the cells tile0 .. tile0 + tileN - 1 (k_grad_fp stages its own cells there: a neighbour inside
// the tile is read from LDS, everyone else from W); null in the one-thread-per-cell body.
template <class T, bool RHO, class G>
DAS_HD void grad_face(int c, int s, const DevMeshT<G>& m, const ResParams& prm, const T* W, const T* Uc, const T& pc, const T& Tc, const T& nc,
                      const T& nut_c, bool energy, T* gU, T* gP, T* gN, T* gH, const T* tile = nullptr, int tile0 = 0, int tileN = 0) {
    const long long N = m.nC;
    const CellGeomT<G>& cgc = m.cg[c];
    int fe = m.cf_face[s];
    int f = fe & 0x7fffffff;
    bool nb = fe < 0;
    const FaceGeomT<G>& g = m.fg[f];
    T Uf[3], pf, nf, hf(0.0);
    double sg = nb ? -1.0 : 1.0;
    if (m.cf_other[s] >= 0) {  // internal face, or a cyclic boundary face (the paired cell acts as the neighbour)
        int o = m.cf_other[s];
        G wc = nb ? G(1.0 - g.w) : g.w;
        T Uo[3], po, no;
        const int ol = o - tile0;
        if (tile && ol >= 0 && ol < tileN) {
            Uo[0] = tile[ol]; Uo[1] = tile[tileN + ol]; Uo[2] = tile[2 * tileN + ol]; po = tile[3 * tileN + ol]; no = tile[4 * tileN + ol];
        } else {
            Uo[0] = W[3LL * o]; Uo[1] = W[3LL * o + 1]; Uo[2] = W[3LL * o + 2]; po = W[prm.offP * N + o]; no = W[prm.offN * N + o];
        }
        const double* Qr = cyclic_rotation(m, f);
        if (Qr) rot_vec<T>(Qr, Uo);
#pragma unroll
        for (int k = 0; k < 3; k++) Uf[k] = wc * Uc[k] + (1.0 - wc) * Uo[k];
        pf = wc * pc + (1.0 - wc) * po;
        nf = wc * nc + (1.0 - wc) * no;
        if (RHO) hf = prm.Cp * (wc * Tc + (1.0 - wc) * W[prm.offT * N + o] - DAS_TREF);
        else if (energy) hf = wc * Tc + (1.0 - wc) * W[prm.offT * N + o];
    } else {
        BFace<T, G> b;
        eval_bface<T, RHO>(m.bc[m.bpatch[f - m.nIF]], g, cgc, prm, Uc, pc, Tc, nc, nut_c, val(W[prm.offPhi * N + f]), b);
#pragma unroll
        for (int k = 0; k < 3; k++) Uf[k] = b.U.xb[k];
        pf = b.p.xb;
        nf = b.n.xb;
        if (energy) hf = b.he.xb;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        G S = sg * g.Sf[i];
#pragma unroll
        for (int j = 0; j < 3; j++) gU[3 * i + j] += S * Uf[j];
        gP[i] += S * pf;
        gN[i] += S * nf;
        if (energy) gH[i] += S * hf;
    }
}
template <class T, bool RHO, class G>
DAS_HD void body_grad(int c, const DevMeshT<G>& m, const ResParams& prm, const T* W, T* nut, T* gradU, T* gradP, T* gradN, T* gradH,
                      T* TU = nullptr) {
    const long long N = m.nC;
    const CellGeomT<G>& cgc = m.cg[c];
    T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c];
    const bool energy = RHO || prm.hasT;  // an energy-like scalar with a gradient (he, or the passive T)
    T Tc = energy ? W[prm.offT * N + c] : T(0.0);
    T nu_c = RHO ? mu_of<T>(prm, Tc) * (prm.Rgas * Tc) / pc : T(prm.nu);
    T nut_c = nc * fv1_of<T>(nc / nu_c);
    nut[c] = nut_c;
    T gU[9], gP[3], gN[3], gH[3];
#pragma unroll
    for (int k = 0; k < 9; k++) gU[k] = T(0.0);
#pragma unroll
    for (int k = 0; k < 3; k++) { gP[k] = T(0.0); gN[k] = T(0.0); gH[k] = T(0.0); }
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) grad_face<T, RHO, G>(c, s, m, prm, W, Uc, pc, Tc, nc, nut_c, energy, gU, gP, gN, gH);
    G rV = 1.0 / cgc.V;
#pragma unroll
    for (int k = 0; k < 9; k++) gradU[9LL * c + k] = gU[k] * rV;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        gradP[3LL * c + k] = gP[k] * rV;
        gradN[3LL * c + k] = gN[k] * rV;
        if (energy) gradH[3LL * c + k] = gH[k] * rV;
    }
    if (RHO && prm.turbo) {  // viscous-work vector Teff.U of the cell (EEqn: - fvc::div(Teff.T() & U))
#pragma unroll
        for (int k = 0; k < 9; k++) gU[k] = gU[k] * rV;
        T rho_c = pc / (prm.Rgas * Tc);
        T q[3];
        teff_dot_u<T>(gU, rho_c * (nu_c + nut_c), Uc, q);
#pragma unroll
        for (int k = 0; k < 3; k++) TU[3LL * c + k] = q[k];
    }
}

// tau = nuEff * dev2(T(gradU)) ; g[3*i+j] = d_i U_j
template <class T>
DAS_HD void dev2T_scaled(const T* g, const T& nuEff, T* tau) {
    T tr23 = (2.0 / 3.0) * (g[0] + g[4] + g[8]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T t = g[3 * j + i];
            if (i == j) t = t - tr23;
            tau[3 * i + j] = nuEff * t;
        }
}

// ================================================================================ k_cell
// per cell: U-equation (diag/off-diag/source incl. boundary coeffs), relax, URes, rAU, HbyA, the SA residual and
// (RHO) the energy residual TRes = EEqn & he.  RHO: phi is the mass flux, muEff = mu + rho nut replaces nuEff.
template <class T, bool RHO, class G>
DAS_HD void body_cell(int c, const DevMeshT<G>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, const T* gradP,
                      const T* gradN, const T* gradH, T* R, T* rAU, T* HbyA, const T* TU = nullptr) {
    const long long N = m.nC;
    const CellGeomT<G>& cgc = m.cg[c];
    T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c];
    const bool energy = RHO || prm.hasT;  // energy equation (RHO) or the passive T equation of DASimpleFoam
    T Tc = energy ? W[prm.offT * N + c] : T(0.0);
    T rho_c = RHO ? pc / (prm.Rgas * Tc) : T(1.0);
    T mu_c = RHO ? mu_of<T>(prm, Tc) : T(0.0);
    T nu_c = RHO ? mu_c / rho_c : T(prm.nu);
    T nut_c = nut[c];
    T muEff_c = rho_c * (nu_c + nut_c);
    T gUc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) gUc[k] = gradU[9LL * c + k];
    T gNc[3] = {gradN[3LL * c], gradN[3LL * c + 1], gradN[3LL * c + 2]};
    T tau_c[9];
    dev2T_scaled<T>(gUc, muEff_c, tau_c);
    T Dn_c = rho_c * (nc + nu_c) * (1.0 / SA_SIGMA);
    // energy (RHO)
    T he_c = RHO ? prm.Cp * (Tc - DAS_TREF) : Tc;
    // alphaEff: compressible mu/Pr + rho nut/Prt ; DASimpleFoam T field nu/Pr + nut/Prt (DAResidualSimpleFoam.C:226)
    T aEff_c = RHO ? mu_c * prm.alphaFac + rho_c * nut_c * (1.0 / prm.Prt) : prm.nu / prm.Pr + nut_c * (1.0 / prm.Prt);
    T K_c = RHO ? 0.5 * (Uc[0] * Uc[0] + Uc[1] * Uc[1] + Uc[2] * Uc[2]) : T(0.0);

    const bool turbo = RHO && prm.turbo;
    const bool mrf = prm.mrf != 0;
    G vC_c[3] = {G(0.0), G(0.0), G(0.0)};
    if (mrf) mrf_velocity(prm, cgc.C, vC_c);
    T D0(0.0), sumOff(0.0), sumPhi(0.0), vmaxs(0.0), vmins(0.0);
    T offU[3], src[3], bdiag[3], bsrc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { offU[k] = T(0.0); src[k] = T(0.0); bdiag[k] = T(0.0); bsrc[k] = T(0.0); }
    T dN(0.0), offN(0.0), sN(0.0), bdN(0.0), bsN(0.0);
    T dE(0.0), offE(0.0), sE(0.0), bdE(0.0), bsE(0.0);

    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        int fe = m.cf_face[s];
        int f = fe & 0x7fffffff;
        bool nb = fe < 0;
        const FaceGeomT<G>& g = m.fg[f];
        T phi = W[prm.offPhi * N + f];
        double sg = nb ? -1.0 : 1.0;
        if (m.cf_other[s] >= 0) {  // internal face, or a cyclic boundary face (the paired cell acts as the neighbour)
            int o = m.cf_other[s];
            const CellGeomT<G>& cgo = m.cg[o];
            double pv = val(phi);
            double wu = pv >= 0.0 ? 1.0 : 0.0;  // upwind weight of the owner value = pos0(flux)
            T dcoef, off;
            if (!nb) { dcoef = wu * phi; off = (1.0 - wu) * phi; }
            else { dcoef = -((1.0 - wu) * phi); off = -(wu * phi); }
            sumPhi += sg * phi;
            T Uo[3] = {W[3LL * o], W[3LL * o + 1], W[3LL * o + 2]};
            const double* Qr = cyclic_rotation(m, f);
            if (Qr) rot_vec<T>(Qr, Uo);
            T nuT_o = W[prm.offN * N + o];
            T rho_o(1.0), nu_o(prm.nu), T_o(0.0);
            if (energy) T_o = W[prm.offT * N + o];
            if (RHO) {
                rho_o = W[prm.offP * N + o] / (prm.Rgas * T_o);
                nu_o = mu_of<T>(prm, T_o) / rho_o;
            }
            T nut_o = nut[o];
            T muEff_o = rho_o * (nu_o + nut_o);
            T gUo[9];
#pragma unroll
            for (int k = 0; k < 9; k++) gUo[k] = gradU[9LL * o + k];
            T gNo[3] = {gradN[3LL * o], gradN[3LL * o + 1], gradN[3LL * o + 2]};
            if (Qr) { rot_ten<T>(Qr, gUo); rot_vec<T>(Qr, gNo); }
            const G wl = g.w;
            const G wc = nb ? G(1.0 - wl) : wl, wo = 1.0 - wc;
            // ---- momentum diffusion  -fvm::laplacian(rho nuEff,U)  (Gauss linear corrected)
            T gam = (wc * muEff_c + wo * muEff_o) * g.magSf;
            T cd = gam * g.nod;
            T offTot = off - cd;
            D0 += dcoef + cd;
            sumOff += dabs(offTot);
#pragma unroll
            for (int k = 0; k < 3; k++) offU[k] += offTot * Uo[k];
            // ---- linearUpwindV explicit correction (PC residual: div(pc) = upwind, i.e. weight 0, unless amd.pcUpwindBlend > 0)
            if (prm.convBlend > 0.0) {
                bool pos = pv > 0.0;
                bool upIsC = (pos != nb);  // upwind cell is the owner when flux > 0
                const T* gUp = upIsC ? gUc : gUo;
                const G* Cup = upIsC ? cgc.C : cgo.C;
                G d[3] = {g.Cf[0] - Cup[0], g.Cf[1] - Cup[1], g.Cf[2] - Cup[2]};
                if (f >= m.nIF && !upIsC) {  // cyclic: the neighbour's face-centre offset lives at the paired face
                    const FaceGeomT<G>& g2 = m.fg[m.cyc[f - m.nIF]];
                    d[0] = g2.Cf[0] - cgo.C[0]; d[1] = g2.Cf[1] - cgo.C[1]; d[2] = g2.Cf[2] - cgo.C[2];
                    if (Qr) rot_vec<G>(Qr, d);
                }
                T corr[3], mx[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    corr[j] = d[0] * gUp[j] + d[1] * gUp[3 + j] + d[2] * gUp[6 + j];
                    T UO = nb ? Uo[j] : Uc[j];
                    T UN = nb ? Uc[j] : Uo[j];
                    mx[j] = pos ? (1.0 - wl) * (UN - UO) : wl * (UO - UN);
                }
                T sfc = corr[0] * corr[0] + corr[1] * corr[1] + corr[2] * corr[2];
                T mxc = corr[0] * mx[0] + corr[1] * mx[1] + corr[2] * mx[2];
                if (val(sfc) > 0.0) {
                    if (val(mxc) < 0.0) { corr[0] = T(0.0); corr[1] = T(0.0); corr[2] = T(0.0); }
                    else if (val(sfc) > val(mxc)) {
                        T sc = mxc / (sfc + DAS_VSMALL);
                        corr[0] = corr[0] * sc; corr[1] = corr[1] * sc; corr[2] = corr[2] * sc;
                    }
                }
                T pout = sg * phi;
#pragma unroll
                for (int j = 0; j < 3; j++) src[j] -= prm.convBlend * (pout * corr[j]);
            }
            // ---- non-orthogonal correction of the laplacian and the explicit dev2 stress term
            T tau_o[9];
            dev2T_scaled<T>(gUo, muEff_o, tau_o);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                T cvg = g.corr[0] * (wc * gUc[j] + wo * gUo[j]) + g.corr[1] * (wc * gUc[3 + j] + wo * gUo[3 + j])
                        + g.corr[2] * (wc * gUc[6 + j] + wo * gUo[6 + j]);
                T tf = g.Sf[0] * (wc * tau_c[j] + wo * tau_o[j]) + g.Sf[1] * (wc * tau_c[3 + j] + wo * tau_o[3 + j])
                       + g.Sf[2] * (wc * tau_c[6 + j] + wo * tau_o[6 + j]);
                src[j] += sg * (gam * cvg + tf);
            }
            // ---- SA convection (bounded upwind) + diffusion
            T Dn_o = rho_o * (nuT_o + nu_o) * (1.0 / SA_SIGMA);
            T gn = (wc * Dn_c + wo * Dn_o) * g.magSf;
            T cdn = gn * g.nod;
            dN += dcoef + cdn;
            offN += (off - cdn) * nuT_o;
            T cvn = g.corr[0] * (wc * gNc[0] + wo * gNo[0]) + g.corr[1] * (wc * gNc[1] + wo * gNo[1]) + g.corr[2] * (wc * gNc[2] + wo * gNo[2]);
            sN += sg * (gn * cvn);
            // ---- energy: div(phi,he) upwind + fvc::div(phi,K) upwind - laplacian(alphaEff, he)
            //      (DASimpleFoam T field: div(phi,T) bounded upwind - laplacian(alphaEff, T), no K)
            if (energy) {
                T he_o = RHO ? prm.Cp * (T_o - DAS_TREF) : T_o;
                T aEff_o = RHO ? (rho_o * nu_o) * prm.alphaFac + rho_o * nut_o * (1.0 / prm.Prt) : prm.nu / prm.Pr + nut_o * (1.0 / prm.Prt);
                T ga = (wc * aEff_c + wo * aEff_o) * g.magSf;
                T cde = ga * g.nod;
                dE += dcoef + cde;
                offE += (off - cde) * he_o;
                T gHo[3] = {gradH[3LL * o], gradH[3LL * o + 1], gradH[3LL * o + 2]};
                if (Qr) rot_vec<T>(Qr, gHo);
                T cve = g.corr[0] * (wc * gradH[3LL * c] + wo * gHo[0]) + g.corr[1] * (wc * gradH[3LL * c + 1] + wo * gHo[1])
                        + g.corr[2] * (wc * gradH[3LL * c + 2] + wo * gHo[2]);
                sE += sg * (ga * cve);
                if (RHO) {
                    T K_o = 0.5 * (Uo[0] * Uo[0] + Uo[1] * Uo[1] + Uo[2] * Uo[2]);
                    // upwind face value of K: owner value if flux >= 0
                    bool ownerIsC = !nb;
                    T Kf = ((pv >= 0.0) == ownerIsC) ? K_c : K_o;
                    sE -= (sg * phi) * Kf;
                }
                if (turbo) {
                    // - fvc::div(Teff.T() & U) (Gauss linear) and + fvc::div(p (U - URel)), U - URel = Omega x r
                    T TUo[3] = {TU[3LL * o], TU[3LL * o + 1], TU[3LL * o + 2]};
                    if (Qr) rot_vec<T>(Qr, TUo);
                    T qf = g.Sf[0] * (wc * TU[3LL * c] + wo * TUo[0]) + g.Sf[1] * (wc * TU[3LL * c + 1] + wo * TUo[1])
                           + g.Sf[2] * (wc * TU[3LL * c + 2] + wo * TUo[2]);
                    sE += sg * qf;
                    if (mrf) {
                        G vC_o[3];
                        mrf_velocity(prm, cgo.C, vC_o);
                        if (Qr) rot_vec<G>(Qr, vC_o);
                        T p_o = W[prm.offP * N + o];
                        T wf = g.Sf[0] * (wc * (pc * vC_c[0]) + wo * (p_o * vC_o[0])) + g.Sf[1] * (wc * (pc * vC_c[1]) + wo * (p_o * vC_o[1]))
                               + g.Sf[2] * (wc * (pc * vC_c[2]) + wo * (p_o * vC_o[2]));
                        sE -= sg * wf;
                    }
                }
            }
        } else {
            BFace<T, G> b;
            double pv = val(phi);
            eval_bface<T, RHO>(m.bc[m.bpatch[f - m.nIF]], g, cgc, prm, Uc, pc, Tc, nc, nut_c, pv, b);
            sumPhi += phi;
            T muEff_b = b.rho_b * (b.nu_b + b.nut_b);
            T gam_b = muEff_b * g.magSf;
            T iC[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                iC[k] = phi * b.U.vic[k] - gam_b * b.U.gic[k];
                bdiag[k] += iC[k];
                bsrc[k] += gam_b * b.U.gbc[k] - phi * b.U.vbc[k];
            }
            T a0 = dabs(iC[0]), a1 = dabs(iC[1]), a2 = dabs(iC[2]);
            T vmx = a0;
            if (val(a1) > val(vmx)) vmx = a1;
            if (val(a2) > val(vmx)) vmx = a2;
            T vmn = iC[0];
            if (val(iC[1]) < val(vmn)) vmn = iC[1];
            if (val(iC[2]) < val(vmn)) vmn = iC[2];
            vmaxs += vmx;
            vmins += vmn;
            // boundary gradient (GaussGrad::correctBoundaryConditions) for the dev2 term
            T gUb[9], dsn[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                T snG = b.U.gic[j] * Uc[j] + b.U.gbc[j];
                T ngU = b.nrm[0] * gUc[j] + b.nrm[1] * gUc[3 + j] + b.nrm[2] * gUc[6 + j];
                dsn[j] = snG - ngU;
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) gUb[3 * i + j] = gUc[3 * i + j] + b.nrm[i] * dsn[j];
            T tau_b[9];
            dev2T_scaled<T>(gUb, muEff_b, tau_b);
#pragma unroll
            for (int j = 0; j < 3; j++) src[j] += g.Sf[0] * tau_b[j] + g.Sf[1] * tau_b[3 + j] + g.Sf[2] * tau_b[6 + j];
            // SA boundary coefficients
            T gn_b = b.rho_b * (b.n.xb + b.nu_b) * (g.magSf / SA_SIGMA);
            bdN += phi * b.n.vic - gn_b * b.n.gic;
            bsN += gn_b * b.n.gbc - phi * b.n.vbc;
            if (energy) {
                T ga_b = (RHO ? b.mu_b * prm.alphaFac + b.rho_b * b.nut_b * (1.0 / prm.Prt) : prm.nu / prm.Pr + b.nut_b * (1.0 / prm.Prt)) * g.magSf;
                bdE += phi * b.he.vic - ga_b * b.he.gic;
                bsE += ga_b * b.he.gbc - phi * b.he.vbc;
                T Kb = 0.5 * (b.U.xb[0] * b.U.xb[0] + b.U.xb[1] * b.U.xb[1] + b.U.xb[2] * b.U.xb[2]);
                if (RHO) sE -= phi * Kb;
                if (turbo) {
                    T qb[3];
                    teff_dot_u<T>(gUb, muEff_b, b.U.xb, qb);
                    sE += g.Sf[0] * qb[0] + g.Sf[1] * qb[1] + g.Sf[2] * qb[2];
                    if (mrf) {
                        const PatchBC& pb_ = m.bc[m.bpatch[f - m.nIF]];
                        T wb[3];
                        if (pb_.mrf_included) { wb[0] = b.U.xb[0]; wb[1] = b.U.xb[1]; wb[2] = b.U.xb[2]; }
                        else {
                            G vF[3];
                            mrf_velocity(prm, g.Cf, vF);
                            wb[0] = T(vF[0]); wb[1] = T(vF[1]); wb[2] = T(vF[2]);
                        }
                        sE -= b.p.xb * (g.Sf[0] * wb[0] + g.Sf[1] * wb[1] + g.Sf[2] * wb[2]);
                    }
                }
            }
        }
    }
    // bounded Gauss: - fvm::Sp(div(phi))
    D0 -= sumPhi;
    dN -= sumPhi;
    if (mrf) {  // + MRF.DDt(rho, U): source -= V rho (Omega x U)   (MRFZone::addCoriolis)
        const T rv = cgc.V * rho_c;
        src[0] -= rv * (prm.om[1] * Uc[2] - prm.om[2] * Uc[1]);
        src[1] -= rv * (prm.om[2] * Uc[0] - prm.om[0] * Uc[2]);
        src[2] -= rv * (prm.om[0] * Uc[1] - prm.om[1] * Uc[0]);
    }
    // fvMatrix::relax (see DESIGN.md "relax"): diagonal dominance fix-up with boundary max/min contributions
    T D = dmax(dabs(D0 + vmaxs), sumOff) * (1.0 / prm.alphaU) - vmins;
    T dD = D - D0;
    const G rV = 1.0 / cgc.V;
    T avgb = (bdiag[0] + bdiag[1] + bdiag[2]) * (1.0 / 3.0);
    T A = (D + avgb) * rV;
    T rA = 1.0 / A;
    rAU[c] = rA;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T sk = src[k] + dD * Uc[k];
        T ures = ((D + bdiag[k]) * Uc[k] + offU[k] - sk - bsrc[k]) * rV + gradP[3LL * c + k];
        if (!prm.normU) ures = ures * cgc.V;
        R[3LL * c + k] = ures;
        T H = ((avgb - bdiag[k]) * Uc[k] - offU[k] + sk + bsrc[k]) * rV;
        HbyA[3LL * c + k] = rA * H;
    }
    if (energy) {
        dE -= sumPhi;
        T tres = ((dE + bdE) * he_c + offE - sE - bsE) * rV;
        if (!prm.normT) tres = tres * cgc.V;
        R[prm.offT * N + c] = tres;
    }
    // ---- SA source terms (DASpalartAllmaras.C:124-178,445-485)
    const G y = cgc.y;
    const G k2y2 = (SA_KAPPA * y) * (SA_KAPPA * y);
    T chi = nc / nu_c;
    T fv1 = fv1_of<T>(chi);
    T fv2 = 1.0 - chi / (1.0 + chi * fv1);
    T w01 = 0.5 * (gUc[1] - gUc[3]), w02 = 0.5 * (gUc[2] - gUc[6]), w12 = 0.5 * (gUc[5] - gUc[7]);
    T Omega = 1.4142135623730951 * dsqrt(2.0 * (w01 * w01 + w02 * w02 + w12 * w12));
    T Stilda = dmax(Omega + fv2 * nc / k2y2, SA_CS * Omega);
    T r = dmin(nc / (dmax(Stilda, DAS_SMALL) * k2y2), 10.0);
    T r2 = r * r, r6 = r2 * r2 * r2;
    T gg = r + SA_CW2 * (r6 - r);
    T g2 = gg * gg, g6 = g2 * g2 * g2;
    const double cw36 = SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3;
    T fw = gg * dpow((1.0 + cw36) / (g6 + cw36), 1.0 / 6.0);
    T convdiff = ((dN + bdN) * nc + offN - sN - bsN) * rV;
    // betaFINuTilda: the field-inversion multiplier of the production term (reference DASpalartAllmaras.C:445-485, betaFINuTilda_;
    // 1 unless a `field` input sets it, DAInputField.C); its tangent is the seed of calcJacTVecProduct(field -> residual)
    const T betaFI = prm.betaFI ? MkSeed<T>::make(prm.betaFI[c], prm.dBetaFI ? prm.dBetaFI[c] : 0.0) : T(1.0);
    T nres = convdiff - rho_c * ((SA_CB2 / SA_SIGMA) * (gNc[0] * gNc[0] + gNc[1] * gNc[1] + gNc[2] * gNc[2]) + SA_CB1 * Stilda * nc * betaFI)
             + SA_CW1 * rho_c * fw * nc / (y * y) * nc;
    if (!prm.normN) nres = nres * cgc.V;
    R[prm.offN * N + c] = nres;
}

// ================================================================================ k_fcoef / k_bcoef / k_cell2
// Round 6: body_cell split the classical finite-volume way for the benchmark path (DASimpleFoam + SA without the T field, MRF and cyclic
// pairs; double metrics).  body_cell evaluates every internal face TWICE (once from each side) inside a serial six-trip loop whose
// dependent loads (face slot -> face record -> neighbour gradients) nothing hides at one wave per SIMD.  Here:
//   body_fcoef  one thread per INTERNAL face: everything that is symmetric in the two cells, once - the diffusion coefficients of the U
//               and nuTilda equations and the explicit face fluxes (non-orthogonal correction, dev2 stress, linearUpwindV correction);
//               SIX scalars per face, stored quantity-major (fc[q nIF + f]);
//   body_bcoef  one thread per BOUNDARY face: the patch coefficients of its cell (13 scalars, brec[13 b + q]);
//   body_cell2  one thread per cell: gathers the records of its faces (the upwind coefficients come from phi), then relax, URes, rAU,
//               HbyA and the SA residual exactly as body_cell.
// Same arithmetic as body_cell up to the association of the face-interpolation weights (1 - (1 - w) vs w); reference arithmetic:
// DAResidualSimpleFoam.C:106-237, DASpalartAllmaras.C:407-488.
#define DAS_FC_N 6    // cd, cdn, F[3], FN
#define DAS_BREC_N 13  // iC[3], bsrc[3], vmx, vmn, srcb[3], bdN, bsN
template <class T>
DAS_HD void body_fcoef(int f, const DevMeshT<double>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, const T* gradN, T* fc) {
    const long long N = m.nC, nIF = m.nIF;
    const int o = m.owner[f], n = m.neigh[f];
    const FaceGeomT<double>& g = m.fg[f];
    const T phi = W[prm.offPhi * N + f];
    const double wl = g.w, wn = 1.0 - g.w;
    const T nuT_o = W[prm.offN * N + o], nuT_n = W[prm.offN * N + n];
    const T muEff_o = prm.nu + nut[o], muEff_n = prm.nu + nut[n];
    const T gam = (wl * muEff_o + wn * muEff_n) * g.magSf;
    fc[0 * nIF + f] = gam * g.nod;
    const T gn = (wl * ((nuT_o + prm.nu) * (1.0 / SA_SIGMA)) + wn * ((nuT_n + prm.nu) * (1.0 / SA_SIGMA))) * g.magSf;
    fc[1 * nIF + f] = gn * g.nod;
    T F[3] = {T(0.0), T(0.0), T(0.0)};
    // owner side, then neighbour side: one gradient tensor live at a time
    T cvg[3] = {T(0.0), T(0.0), T(0.0)};
    T corrv[3] = {T(0.0), T(0.0), T(0.0)};
    const double pv = val(phi);
    const bool pos = pv > 0.0;  // the upwind cell is the owner when the flux is positive
    T cvn(0.0);
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int c = side == 0 ? o : n;
        const double wc = side == 0 ? wl : wn;
        T gU[9];
#pragma unroll
        for (int k = 0; k < 9; k++) gU[k] = gradU[9LL * c + k];
        T tau[9];
        dev2T_scaled<T>(gU, side == 0 ? muEff_o : muEff_n, tau);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cvg[j] += wc * (g.corr[0] * gU[j] + g.corr[1] * gU[3 + j] + g.corr[2] * gU[6 + j]);
            F[j] += wc * (g.Sf[0] * tau[j] + g.Sf[1] * tau[3 + j] + g.Sf[2] * tau[6 + j]);
        }
        if (prm.convBlend > 0.0 && (pos == (side == 0))) {
            const CellGeomT<double>& cgu = m.cg[c];
            const double d[3] = {g.Cf[0] - cgu.C[0], g.Cf[1] - cgu.C[1], g.Cf[2] - cgu.C[2]};
#pragma unroll
            for (int j = 0; j < 3; j++) corrv[j] = d[0] * gU[j] + d[1] * gU[3 + j] + d[2] * gU[6 + j];
        }
        cvn += wc * (g.corr[0] * gradN[3LL * c] + g.corr[1] * gradN[3LL * c + 1] + g.corr[2] * gradN[3LL * c + 2]);
    }
    if (prm.convBlend > 0.0) {
        T mx[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const T UO = W[3LL * o + j], UN = W[3LL * n + j];
            mx[j] = pos ? wn * (UN - UO) : wl * (UO - UN);
        }
        T sfc = corrv[0] * corrv[0] + corrv[1] * corrv[1] + corrv[2] * corrv[2];
        T mxc = corrv[0] * mx[0] + corrv[1] * mx[1] + corrv[2] * mx[2];
        if (val(sfc) > 0.0) {
            if (val(mxc) < 0.0) { corrv[0] = T(0.0); corrv[1] = T(0.0); corrv[2] = T(0.0); }
            else if (val(sfc) > val(mxc)) {
                T sc = mxc / (sfc + DAS_VSMALL);
                corrv[0] = corrv[0] * sc; corrv[1] = corrv[1] * sc; corrv[2] = corrv[2] * sc;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) F[j] -= prm.convBlend * (phi * corrv[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) fc[(2 + j) * nIF + f] = F[j] + gam * cvg[j];
    fc[5 * nIF + f] = gn * cvn;
}

template <class T>
DAS_HD void body_bcoef(int b, const DevMeshT<double>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, T* brec) {
    const long long N = m.nC;
    const int f = m.nIF + b, c = m.owner[f];
    const FaceGeomT<double>& g = m.fg[f];
    const CellGeomT<double>& cgc = m.cg[c];
    const T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    const T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c], nut_c = nut[c];
    const T phi = W[prm.offPhi * N + f];
    T gUc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) gUc[k] = gradU[9LL * c + k];
    BFace<T, double> bf;
    eval_bface<T, false>(m.bc[m.bpatch[b]], g, cgc, prm, Uc, pc, T(0.0), nc, nut_c, val(phi), bf);
    T* r = brec + (long long)DAS_BREC_N * b;
    const T muEff_b = bf.rho_b * (bf.nu_b + bf.nut_b);
    const T gam_b = muEff_b * g.magSf;
    T iC[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        iC[k] = phi * bf.U.vic[k] - gam_b * bf.U.gic[k];
        r[k] = iC[k];
        r[3 + k] = gam_b * bf.U.gbc[k] - phi * bf.U.vbc[k];
    }
    T a0 = dabs(iC[0]), a1 = dabs(iC[1]), a2 = dabs(iC[2]);
    T vmx = a0;
    if (val(a1) > val(vmx)) vmx = a1;
    if (val(a2) > val(vmx)) vmx = a2;
    T vmn = iC[0];
    if (val(iC[1]) < val(vmn)) vmn = iC[1];
    if (val(iC[2]) < val(vmn)) vmn = iC[2];
    r[6] = vmx;
    r[7] = vmn;
    T gUb[9], dsn[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T snG = bf.U.gic[j] * Uc[j] + bf.U.gbc[j];
        T ngU = bf.nrm[0] * gUc[j] + bf.nrm[1] * gUc[3 + j] + bf.nrm[2] * gUc[6 + j];
        dsn[j] = snG - ngU;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) gUb[3 * i + j] = gUc[3 * i + j] + bf.nrm[i] * dsn[j];
    T tau_b[9];
    dev2T_scaled<T>(gUb, muEff_b, tau_b);
#pragma unroll
    for (int j = 0; j < 3; j++) r[8 + j] = g.Sf[0] * tau_b[j] + g.Sf[1] * tau_b[3 + j] + g.Sf[2] * tau_b[6 + j];
    const T gn_b = bf.rho_b * (bf.n.xb + bf.nu_b) * (g.magSf / SA_SIGMA);
    r[11] = phi * bf.n.vic - gn_b * bf.n.gic;
    r[12] = gn_b * bf.n.gbc - phi * bf.n.vbc;
}

template <class T>
DAS_HD void body_cell2(int c, const DevMeshT<double>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, const T* gradP, const T* gradN,
                       const T* fc, const T* brec, T* R, T* rAU, T* HbyA) {
    const long long N = m.nC, nIF = m.nIF;
    const CellGeomT<double>& cgc = m.cg[c];
    const T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    const T nc = W[prm.offN * N + c];
    T D0(0.0), sumOff(0.0), sumPhi(0.0), vmaxs(0.0), vmins(0.0);
    T offU[3], src[3], bdiag[3], bsrc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { offU[k] = T(0.0); src[k] = T(0.0); bdiag[k] = T(0.0); bsrc[k] = T(0.0); }
    T dN(0.0), offN(0.0), sN(0.0), bdN(0.0), bsN(0.0);
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        const int fe = m.cf_face[s];
        const int f = fe & 0x7fffffff;
        const bool nb = fe < 0;
        const T phi = W[prm.offPhi * N + f];
        if (f < nIF) {
            const int o = m.cf_other[s];
            const double sg = nb ? -1.0 : 1.0;
            const double wu = val(phi) >= 0.0 ? 1.0 : 0.0;  // upwind weight of the owner value = pos0(flux)
            T dcoef, off;
            if (!nb) { dcoef = wu * phi; off = (1.0 - wu) * phi; }
            else { dcoef = -((1.0 - wu) * phi); off = -(wu * phi); }
            sumPhi += sg * phi;
            const T cd = fc[f], cdn = fc[nIF + f];
            const T offTot = off - cd;
            D0 += dcoef + cd;
            sumOff += dabs(offTot);
#pragma unroll
            for (int k = 0; k < 3; k++) { offU[k] += offTot * W[3LL * o + k]; src[k] += sg * fc[(2 + k) * nIF + f]; }
            dN += dcoef + cdn;
            offN += (off - cdn) * W[prm.offN * N + o];
            sN += sg * fc[5 * nIF + f];
        } else {
            const T* r = brec + (long long)DAS_BREC_N * (f - nIF);
            sumPhi += phi;
#pragma unroll
            for (int k = 0; k < 3; k++) { bdiag[k] += r[k]; bsrc[k] += r[3 + k]; src[k] += r[8 + k]; }
            vmaxs += r[6];
            vmins += r[7];
            bdN += r[11];
            bsN += r[12];
        }
    }
    D0 -= sumPhi;
    dN -= sumPhi;
    T D = dmax(dabs(D0 + vmaxs), sumOff) * (1.0 / prm.alphaU) - vmins;
    T dD = D - D0;
    const double rV = 1.0 / cgc.V;
    T avgb = (bdiag[0] + bdiag[1] + bdiag[2]) * (1.0 / 3.0);
    T A = (D + avgb) * rV;
    T rA = 1.0 / A;
    rAU[c] = rA;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T sk = src[k] + dD * Uc[k];
        T ures = ((D + bdiag[k]) * Uc[k] + offU[k] - sk - bsrc[k]) * rV + gradP[3LL * c + k];
        if (!prm.normU) ures = ures * cgc.V;
        R[3LL * c + k] = ures;
        T H = ((avgb - bdiag[k]) * Uc[k] - offU[k] + sk + bsrc[k]) * rV;
        HbyA[3LL * c + k] = rA * H;
    }
    // ---- SA source terms (as body_cell)
    const T gNc[3] = {gradN[3LL * c], gradN[3LL * c + 1], gradN[3LL * c + 2]};
    const double y = cgc.y;
    const double k2y2 = (SA_KAPPA * y) * (SA_KAPPA * y);
    T chi = nc / T(prm.nu);
    T fv1 = fv1_of<T>(chi);
    T fv2 = 1.0 - chi / (1.0 + chi * fv1);
    T w01 = 0.5 * (gradU[9LL * c + 1] - gradU[9LL * c + 3]), w02 = 0.5 * (gradU[9LL * c + 2] - gradU[9LL * c + 6]), w12 = 0.5 * (gradU[9LL * c + 5] - gradU[9LL * c + 7]);
    T Omega = 1.4142135623730951 * dsqrt(2.0 * (w01 * w01 + w02 * w02 + w12 * w12));
    T Stilda = dmax(Omega + fv2 * nc / k2y2, SA_CS * Omega);
    T r = dmin(nc / (dmax(Stilda, DAS_SMALL) * k2y2), 10.0);
    T r2 = r * r, r6 = r2 * r2 * r2;
    T gg = r + SA_CW2 * (r6 - r);
    T g2 = gg * gg, g6 = g2 * g2 * g2;
    const double cw36 = SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3 * SA_CW3;
    T fw = gg * dpow((1.0 + cw36) / (g6 + cw36), 1.0 / 6.0);
    T convdiff = ((dN + bdN) * nc + offN - sN - bsN) * rV;
    const T betaFI = prm.betaFI ? MkSeed<T>::make(prm.betaFI[c], prm.dBetaFI ? prm.dBetaFI[c] : 0.0) : T(1.0);
    T nres = convdiff - ((SA_CB2 / SA_SIGMA) * (gNc[0] * gNc[0] + gNc[1] * gNc[1] + gNc[2] * gNc[2]) + SA_CB1 * Stilda * nc * betaFI)
             + SA_CW1 * fw * nc / (y * y) * nc;
    if (!prm.normN) nres = nres * cgc.V;
    R[prm.offN * N + c] = nres;
}

// ================================================================================ k_face
// per face: phiHbyA, pressure flux, q = flux - phiHbyA (consumed by k_pres) and phiRes
template <class T, bool RHO, class G>
DAS_HD void body_face(int f, const DevMeshT<G>& m, const ResParams& prm, const T* W, const T* nut, const T* gradP, const T* rAU,
                      const T* HbyA, T* q, T* R) {
    const long long N = m.nC;
    const FaceGeomT<G>& g = m.fg[f];
    T phiHbyA, flux;
    const bool turbo = RHO && prm.turbo;
    const G rel = prm.mrf ? mrf_face_flux(prm, g) : G(0.0);
    if (turbo) {
        // DAResidualTurboFoam.C:146-212.  "phiHbyA" below is what enters div(): phiHbyA (+ SIMPLEC correction), or the
        // convective flux phid_f p_upwind of fvm::div(phid,p) in the transonic form.
        T snGradP, rho_f;
        if (f < m.nIF || m.cyc[f - m.nIF] >= 0) {  // internal or cyclic face
            int o = m.owner[f], n = f < m.nIF ? m.neigh[f] : m.owner[m.cyc[f - m.nIF]];
            const G wl = g.w, wn = 1.0 - g.w;
            T po = W[prm.offP * N + o], pn = W[prm.offP * N + n];
            T psio = 1.0 / (prm.Rgas * W[prm.offT * N + o]), psin = 1.0 / (prm.Rgas * W[prm.offT * N + n]);
            T ro = po * psio, rn = pn * psin;
            T gPn[3] = {gradP[3LL * n], gradP[3LL * n + 1], gradP[3LL * n + 2]};
            T Hn[3] = {HbyA[3LL * n], HbyA[3LL * n + 1], HbyA[3LL * n + 2]};
            const double* Qr = cyclic_rotation(m, f);
            if (Qr) { rot_vec<T>(Qr, gPn); rot_vec<T>(Qr, Hn); }
            T cg = g.corr[0] * (wl * gradP[3LL * o] + wn * gPn[0]) + g.corr[1] * (wl * gradP[3LL * o + 1] + wn * gPn[1])
                   + g.corr[2] * (wl * gradP[3LL * o + 2] + wn * gPn[2]);
            snGradP = g.nod * (pn - po) + cg;
            rho_f = wl * ro + wn * rn;
            if (!prm.transonic) {
                phiHbyA = g.Sf[0] * (wl * (ro * HbyA[3LL * o]) + wn * (rn * Hn[0]))
                          + g.Sf[1] * (wl * (ro * HbyA[3LL * o + 1]) + wn * (rn * Hn[1]))
                          + g.Sf[2] * (wl * (ro * HbyA[3LL * o + 2]) + wn * (rn * Hn[2])) - rho_f * rel;
                // SIMPLEC-consistent form (AtU = AU - H1): interpolate(rho/AtU - rho/AU) snGrad(p) |Sf| is added to phiHbyA
                // and to the laplacian(rho/AtU, p) flux alike, so it cancels identically in pRes and phiRes (linear
                // interpolation is linear); what remains is the rho/AU flux.  The oracle keeps both terms.
                flux = (wl * (ro * rAU[o]) + wn * (rn * rAU[n])) * g.magSf * snGradP;
            } else {
                T hs = g.Sf[0] * (wl * HbyA[3LL * o] + wn * Hn[0]) + g.Sf[1] * (wl * HbyA[3LL * o + 1] + wn * Hn[1])
                       + g.Sf[2] * (wl * HbyA[3LL * o + 2] + wn * Hn[2]);
                T phid = (wl * psio + wn * psin) * (hs - rel);
                phiHbyA = phid * (val(phid) >= 0.0 ? po : pn);
                flux = (wl * (ro * rAU[o]) + wn * (rn * rAU[n])) * g.magSf * snGradP;
            }
        } else {
            int c = m.owner[f];
            const PatchBC& bc = m.bc[m.bpatch[f - m.nIF]];
            T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
            T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c], Tc = W[prm.offT * N + c];
            BFace<T, G> b;
            eval_bface<T, RHO>(bc, g, m.cg[c], prm, Uc, pc, Tc, nc, nut[c], val(W[prm.offPhi * N + f]), b);
            T Hb[3] = {HbyA[3LL * c], HbyA[3LL * c + 1], HbyA[3LL * c + 2]};
            if (bc.U_code == DAS_BC_SYMMETRY) {
                T hn = b.nrm[0] * Hb[0] + b.nrm[1] * Hb[1] + b.nrm[2] * Hb[2];
#pragma unroll
                for (int k = 0; k < 3; k++) Hb[k] = Hb[k] - b.nrm[k] * hn;
            }
            if (prm.constrainHbyA && bc.U_code == DAS_BC_FIXED_VALUE) {
#pragma unroll
                for (int k = 0; k < 3; k++) Hb[k] = b.U.xb[k];
            }
            snGradP = b.p.gic * pc + b.p.gbc;
            T hs = g.Sf[0] * Hb[0] + g.Sf[1] * Hb[1] + g.Sf[2] * Hb[2];
            const bool incl = prm.mrf && bc.mrf_included;
            if (!prm.transonic) {
                phiHbyA = incl ? T(0.0) : b.rho_b * (hs - rel);
                flux = (b.rho_b * rAU[c]) * g.magSf * snGradP;
            } else {
                T psib = 1.0 / (prm.Rgas * b.Tt.xb);
                T phid = incl ? T(0.0) : psib * (hs - rel);
                phiHbyA = phid * b.p.xb;
                flux = (b.rho_b * rAU[c]) * g.magSf * snGradP;
            }
        }
        if (prm.transonic && prm.isPC && prm.transonicPC == 1) phiHbyA = T(0.0);  // the PC drops div(phid,p)
        q[f] = flux - phiHbyA;
        T pr = phiHbyA - flux - W[prm.offPhi * N + f];
        if (prm.transonic && prm.isPC && prm.transonicPC == 2) pr = W[prm.offPhi * N + f];  // phiRes = phi
        if (prm.normPhi) pr = pr * (1.0 / g.magSf);
        R[prm.offPhi * N + f] = pr;
        return;
    }
    if (f < m.nIF || m.cyc[f - m.nIF] >= 0) {  // internal or cyclic face
        int o = m.owner[f], n = f < m.nIF ? m.neigh[f] : m.owner[m.cyc[f - m.nIF]];
        const G wl = g.w, wn = 1.0 - g.w;
        T gPn[3] = {gradP[3LL * n], gradP[3LL * n + 1], gradP[3LL * n + 2]};
        T Hn[3] = {HbyA[3LL * n], HbyA[3LL * n + 1], HbyA[3LL * n + 2]};
        const double* Qr = cyclic_rotation(m, f);
        if (Qr) { rot_vec<T>(Qr, gPn); rot_vec<T>(Qr, Hn); }
        phiHbyA = g.Sf[0] * (wl * HbyA[3LL * o] + wn * Hn[0]) + g.Sf[1] * (wl * HbyA[3LL * o + 1] + wn * Hn[1])
                  + g.Sf[2] * (wl * HbyA[3LL * o + 2] + wn * Hn[2]);
        T ro(1.0), rn(1.0);
        phiHbyA = phiHbyA - rel;  // MRF.makeRelative(phiHbyA) / makeRelative(interpolate(rho), phiHbyA)
        if (RHO) {
            ro = W[prm.offP * N + o] / (prm.Rgas * W[prm.offT * N + o]);
            rn = W[prm.offP * N + n] / (prm.Rgas * W[prm.offT * N + n]);
            phiHbyA = (wl * ro + wn * rn) * phiHbyA;
        }
        T gp = (wl * (ro * rAU[o]) + wn * (rn * rAU[n])) * g.magSf;
        T cg = g.corr[0] * (wl * gradP[3LL * o] + wn * gPn[0]) + g.corr[1] * (wl * gradP[3LL * o + 1] + wn * gPn[1])
               + g.corr[2] * (wl * gradP[3LL * o + 2] + wn * gPn[2]);
        flux = gp * (g.nod * (W[prm.offP * N + n] - W[prm.offP * N + o]) + cg);
    } else {
        int c = m.owner[f];
        const PatchBC& bc = m.bc[m.bpatch[f - m.nIF]];
        T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
        T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c];
        T Tc = RHO ? W[prm.offT * N + c] : T(0.0);
        BFace<T, G> b;
        eval_bface<T, RHO>(bc, g, m.cg[c], prm, Uc, pc, Tc, nc, nut[c], val(W[prm.offPhi * N + f]), b);
        T Hb[3] = {HbyA[3LL * c], HbyA[3LL * c + 1], HbyA[3LL * c + 2]};
        if (bc.U_code == DAS_BC_SYMMETRY) {
            T hn = b.nrm[0] * Hb[0] + b.nrm[1] * Hb[1] + b.nrm[2] * Hb[2];
#pragma unroll
            for (int k = 0; k < 3; k++) Hb[k] = Hb[k] - b.nrm[k] * hn;
        }
        if (prm.constrainHbyA && bc.U_code == DAS_BC_FIXED_VALUE) {
#pragma unroll
            for (int k = 0; k < 3; k++) Hb[k] = b.U.xb[k];
        }
        phiHbyA = b.rho_b * (g.Sf[0] * Hb[0] + g.Sf[1] * Hb[1] + g.Sf[2] * Hb[2] - rel);
        if (prm.mrf && bc.mrf_included) phiHbyA = T(0.0);  // relative flux through a patch that rotates with the zone
        flux = (b.rho_b * rAU[c]) * g.magSf * (b.p.gic * pc + b.p.gbc);
    }
    q[f] = flux - phiHbyA;
    T pr = phiHbyA - flux - W[prm.offPhi * N + f];
    if (prm.normPhi) pr = pr * (1.0 / g.magSf);
    R[prm.offPhi * N + f] = pr;
}

// ================================================================================ k_pres
// incompressible: pRes = (laplacian(rAU,p) - div(phiHbyA))/V = sum(q)/V ; compressible: pEqn = div(phiHbyA) - laplacian -> -sum(q)/V
template <class T, bool RHO, class G>
DAS_HD void body_pres(int c, const DevMeshT<G>& m, const ResParams& prm, const T* q, T* R) {
    const long long N = m.nC;
    T s(0.0);
    for (int k = m.cf_ptr[c]; k < m.cf_ptr[c + 1]; k++) {
        int fe = m.cf_face[k];
        int f = fe & 0x7fffffff;
        if (fe < 0) s -= q[f];
        else s += q[f];
    }
    if (RHO) s = -s;
    if (prm.normP) s = s * (1.0 / m.cg[c].V);
    R[prm.offP * N + c] = s;
}

// ================================================================================ force objective
// DAFunctionForce::calcFunction (reference src/adjoint/DAFunction/DAFunctionForce.C:79-158) for one boundary face:
//   F_f = scale * ( S_f p_b + S_f . devRhoReff_b ) . dir,  devRhoReff = (-rho nuEff) dev(twoSymm(grad U))
//   (reference DATurbulenceModel.C:360-376), boundary field from the boundary values of nuEff, rho and grad(U).
template <class T, bool RHO, class G, class DV>
DAS_HD T body_force(int f, const DevMeshT<G>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, const DV* dir, double scale) {
    const long long N = m.nC;
    const FaceGeomT<G>& g = m.fg[f];
    const int c = m.owner[f];
    T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c];
    T Tc = RHO ? W[prm.offT * N + c] : T(0.0);
    BFace<T, G> b;
    eval_bface<T, RHO>(m.bc[m.bpatch[f - m.nIF]], g, m.cg[c], prm, Uc, pc, Tc, nc, nut[c], val(W[prm.offPhi * N + f]), b);
    T gUb[9], dsn[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T snG = b.U.gic[j] * Uc[j] + b.U.gbc[j];
        T ngU = b.nrm[0] * gradU[9LL * c + j] + b.nrm[1] * gradU[9LL * c + 3 + j] + b.nrm[2] * gradU[9LL * c + 6 + j];
        dsn[j] = snG - ngU;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) gUb[3 * i + j] = gradU[9LL * c + 3 * i + j] + b.nrm[i] * dsn[j];
    T muEff_b = b.rho_b * (b.nu_b + b.nut_b);
    T tr3 = (2.0 / 3.0) * (gUb[0] + gUb[4] + gUb[8]);  // tr(twoSymm)/3
    T acc(0.0);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T fT(0.0);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            T sij = gUb[3 * i + j] + gUb[3 * j + i];
            if (i == j) sij = sij - tr3;
            fT += g.Sf[i] * sij;
        }
        acc += dir[j] * (g.Sf[j] * b.p.xb - muEff_b * fT);
    }
    return scale * acc;
}

// ================================================================================ face-integral objectives
// One boundary face's contribution q_f of the reference's patch functions (the weights - scale, area fractions, the
// quotient rule of ratio functions - are applied by the caller):
//   kind 0  force / moment   (S_f p_b + S_f . devRhoReff_b) . d_f      DAFunctionForce.C:79-158, DAFunctionMoment.C:73-120
//                            (moment: d_f = axis x (C_f - center))
//   kind 1  massFlowRate     rho_b (U_b . S_f)                          DAFunctionMassFlowRate.C:52-80
//   kind 2  totalPressure    p_b + 0.5 rho_b |U_b|^2                    DAFunctionTotalPressure.C:60-90
//   kind 3  totalTemperature T_b (1 + 0.5 (gamma-1) Ma^2), Ma^2 = |U_b|^2 / (gamma R T_b), R = Cp - Cp/gamma
//                                                                       DAFunctionTotalTemperatureRatio.C:88-125
#define DAS_FN_FORCE 0
#define DAS_FN_MASSFLOW 1
#define DAS_FN_TOTALPRESSURE 2
#define DAS_FN_TOTALTEMPERATURE 3
template <class T, bool RHO, class G, class DV>
DAS_HD T body_facefn(int f, const DevMeshT<G>& m, const ResParams& prm, const T* W, const T* nut, const T* gradU, int kind, const DV* dir,
                     double gammaFn, double RFn) {
    if (kind == DAS_FN_FORCE) return body_force<T, RHO>(f, m, prm, W, nut, gradU, dir, 1.0);
    const long long N = m.nC;
    const FaceGeomT<G>& g = m.fg[f];
    const int c = m.owner[f];
    T Uc[3] = {W[3LL * c], W[3LL * c + 1], W[3LL * c + 2]};
    T pc = W[prm.offP * N + c], nc = W[prm.offN * N + c];
    T Tc = RHO ? W[prm.offT * N + c] : T(0.0);
    BFace<T, G> b;
    eval_bface<T, RHO>(m.bc[m.bpatch[f - m.nIF]], g, m.cg[c], prm, Uc, pc, Tc, nc, nut[c], val(W[prm.offPhi * N + f]), b);
    T U2 = b.U.xb[0] * b.U.xb[0] + b.U.xb[1] * b.U.xb[1] + b.U.xb[2] * b.U.xb[2];
    if (kind == DAS_FN_MASSFLOW) return b.rho_b * (b.U.xb[0] * g.Sf[0] + b.U.xb[1] * g.Sf[1] + b.U.xb[2] * g.Sf[2]);
    if (kind == DAS_FN_TOTALPRESSURE) return b.p.xb + 0.5 * b.rho_b * U2;
    // total temperature (compressible solvers only; the caller checks)
    return b.Tt.xb + (0.5 * (gammaFn - 1.0) / (gammaFn * RFn)) * U2;
}

// ================================================================================ DAScalarTransportFoam
template <class T, class G>
DAS_HD void body_gradT(int c, const DevMeshT<G>& m, const ResParams& prm, const T* W, const double* phiF, T* gradT) {
    const CellGeomT<G>& cgc = m.cg[c];
    T Tc = W[c];
    T gT[3] = {T(0.0), T(0.0), T(0.0)};
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        int fe = m.cf_face[s];
        int f = fe & 0x7fffffff;
        bool nb = fe < 0;
        const FaceGeomT<G>& g = m.fg[f];
        double sg = nb ? -1.0 : 1.0;
        T Tf;
        if (f < m.nIF) {
            G wc = nb ? G(1.0 - g.w) : g.w;
            Tf = wc * Tc + (1.0 - wc) * W[m.cf_other[s]];
        } else {
            const PatchBC& bc = m.bc[m.bpatch[f - m.nIF]];
            ScalarBC<T, G> b;
            bc_scalar<T>(bc.T_code, bc.T_val, bc.dT_val, g.nod, phiF[f], Tc, b);
            Tf = b.xb;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) gT[i] += (sg * g.Sf[i]) * Tf;
    }
    G rV = 1.0 / cgc.V;
#pragma unroll
    for (int i = 0; i < 3; i++) gradT[3LL * c + i] = gT[i] * rV;
}

// TRes = (ddt(T) + div(phi,T) - laplacian(DT,T)) & T   (Euler, Gauss upwind, Gauss linear corrected)
template <class T, class G>
DAS_HD void body_T(int c, const DevMeshT<G>& m, const ResParams& prm, const T* W, const double* phiF, const double* Told,
                   const T* gradT, T* R) {
    const CellGeomT<G>& cgc = m.cg[c];
    T Tc = W[c];
    T d(0.0), off(0.0), sS(0.0), bd(0.0), bs(0.0);
    for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
        int fe = m.cf_face[s];
        int f = fe & 0x7fffffff;
        bool nb = fe < 0;
        const FaceGeomT<G>& g = m.fg[f];
        double sg = nb ? -1.0 : 1.0;
        double phi = phiF[f];
        if (f < m.nIF) {
            int o = m.cf_other[s];
            double wu = phi >= 0.0 ? 1.0 : 0.0;
            double dcoef, offc;
            if (!nb) { dcoef = wu * phi; offc = (1.0 - wu) * phi; }
            else { dcoef = -(1.0 - wu) * phi; offc = -wu * phi; }
            G gam = prm.DT * g.magSf;
            G cd = gam * g.nod;
            d += dcoef + cd;
            off += (offc - cd) * W[o];
            G wc = nb ? G(1.0 - g.w) : g.w, wo = 1.0 - wc;
            T cv = g.corr[0] * (wc * gradT[3LL * c] + wo * gradT[3LL * o]) + g.corr[1] * (wc * gradT[3LL * c + 1] + wo * gradT[3LL * o + 1])
                   + g.corr[2] * (wc * gradT[3LL * c + 2] + wo * gradT[3LL * o + 2]);
            sS += (sg * gam) * cv;
        } else {
            const PatchBC& bc = m.bc[m.bpatch[f - m.nIF]];
            ScalarBC<T, G> b;
            bc_scalar<T>(bc.T_code, bc.T_val, bc.dT_val, g.nod, phi, Tc, b);
            G gam_b = prm.DT * g.magSf;
            bd += phi * b.vic - gam_b * b.gic;
            bs += gam_b * b.gbc - phi * b.vbc;
        }
    }
    G rdt = cgc.V / prm.deltaT;
    d += rdt;
    sS += rdt * Told[c];
    T res = ((d + bd) * Tc + off - sS - bs) * (1.0 / cgc.V);
    if (!prm.normT) res = res * cgc.V;
    R[c] = res;
}

// a face-integral objective as the kernels see it (k_fn_value / k_fn_grad / k_fn_face)
struct FaceFnView {
    const int* faces;
    const unsigned char* group;  // 0 / 1: denominator / numerator set of ratio functions (all 0 otherwise)
    const double* w;             // per-face weight (value pass: base weights; derivative passes: effective weights)
    const double* dir;           // 3 per face (force / moment), may be null for the other kinds
    int nf, kind;
    double gammaFn, RFn;
};

}  // namespace das
