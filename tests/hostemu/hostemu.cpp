// TEST HARNESS ONLY (never linked into libdafoam_amd.so, never used by the product path).
// Runs the *same* templated kernel bodies that the HIP kernels wrap (csrc/das_kernels.hpp) in plain host
// loops so that the CPU-only test tier can check them against the oracle (tests/test_kernel_bodies_cpu.py).
#include "../../dafoam_amd/csrc/das_case.hpp"

using namespace das;

static int g_split = 0;  // 1: DASimpleFoam cell pass through the face / cell split (emu_set_cell_split)
extern "C" void emu_set_cell_split(int on) { g_split = on; }
template <class T, class MV>
static void eval_on(const MV& m, const CaseParams& cp, const ResParams& prm, const std::vector<T>& W, std::vector<T>& R);
template <class T>
static void eval(const Mesh& mesh, const CaseParams& cp, const ResParams& prm, const std::vector<T>& W, std::vector<T>& R) {
    eval_on<T>(host_view(mesh), cp, prm, W, R);
}
template <class T, class MV>
static void eval_on(const MV& m, const CaseParams& cp, const ResParams& prm, const std::vector<T>& W, std::vector<T>& R) {
    const long long N = m.nC;
    if (cp.solver == DAS_SOLVER_SIMPLEFOAM) {
        std::vector<T> nut(N), gU(9 * N), gP(3 * N), gN(3 * N), gH(3 * N), rAU(N), HbyA(3 * N), q(m.nF);
        for (int c = 0; c < m.nC; c++) body_grad<T, false>(c, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data());
        bool split = false;
        if constexpr (std::is_same<MV, DevMeshT<double>>::value) {
            // the face / cell split of body_cell (body_fcoef + body_bcoef + body_cell2) where the product path launches it
            bool cyc = false;
            for (int b = 0; b < m.nF - m.nIF; b++) cyc = cyc || m.cyc[b] >= 0;
            if (g_split && !prm.hasT && !prm.mrf && !cyc) {
                split = true;
                std::vector<T> fc((size_t)DAS_FC_N * m.nIF), brec((size_t)DAS_BREC_N * (m.nF - m.nIF));
                for (int f = 0; f < m.nIF; f++) body_fcoef<T>(f, m, prm, W.data(), nut.data(), gU.data(), gN.data(), fc.data());
                for (int b = 0; b < m.nF - m.nIF; b++) body_bcoef<T>(b, m, prm, W.data(), nut.data(), gU.data(), brec.data());
                for (int c = 0; c < m.nC; c++)
                    body_cell2<T>(c, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), fc.data(), brec.data(), R.data(), rAU.data(), HbyA.data());
            }
        }
        for (int c = 0; c < m.nC && !split; c++)
            body_cell<T, false>(c, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data(), R.data(), rAU.data(), HbyA.data());
        for (int f = 0; f < m.nF; f++) body_face<T, false>(f, m, prm, W.data(), nut.data(), gP.data(), rAU.data(), HbyA.data(), q.data(), R.data());
        for (int c = 0; c < m.nC; c++) body_pres<T, false>(c, m, prm, q.data(), R.data());
    } else if (DAS_IS_COMPRESSIBLE(cp.solver)) {
        std::vector<T> nut(N), gU(9 * N), gP(3 * N), gN(3 * N), gH(3 * N), rAU(N), HbyA(3 * N), q(m.nF), TU(3 * N);
        for (int c = 0; c < m.nC; c++) body_grad<T, true>(c, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data(), TU.data());
        for (int c = 0; c < m.nC; c++)
            body_cell<T, true>(c, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data(), R.data(), rAU.data(), HbyA.data(),
                               TU.data());
        for (int f = 0; f < m.nF; f++) body_face<T, true>(f, m, prm, W.data(), nut.data(), gP.data(), rAU.data(), HbyA.data(), q.data(), R.data());
        for (int c = 0; c < m.nC; c++) body_pres<T, true>(c, m, prm, q.data(), R.data());
    } else {
        std::vector<T> gT(3 * N);
        for (int c = 0; c < m.nC; c++) body_gradT<T>(c, m, prm, W.data(), cp.phi_frozen.data(), gT.data());
        for (int c = 0; c < m.nC; c++) body_T<T>(c, m, prm, W.data(), cp.phi_frozen.data(), cp.T_old.data(), gT.data(), R.data());
    }
}

static double g_pcBlend = 0.0;  // amd.pcUpwindBlend of the emulated PC residual
extern "C" void emu_set_pc_blend(double b) { g_pcBlend = b; }
extern "C" int emu_residual(const das_case_t* c, const double* Win, long long n, int isPC, const double* dir, double* Rv, double* Rd) {
    try {
        Mesh mesh;
        mesh.build(c);
        CaseParams cp;
        cp.from_case(c);
        if (!cp.beta_fi.empty()) cp.betaFI_ptr = cp.beta_fi.data();
        Options opt;
        opt.d["amd.pcUpwindBlend"] = g_pcBlend;
        ResParams prm = make_params(cp, opt, isPC);
        if (!dir) {
            std::vector<double> W(Win, Win + n), R(n);
            eval<double>(mesh, cp, prm, W, R);
            for (long long i = 0; i < n; i++) Rv[i] = R[i];
        } else {
            std::vector<Dual<1>> W(n), R(n);
            for (long long i = 0; i < n; i++) { W[i].v = Win[i]; W[i].d[0] = dir[i]; }
            eval<Dual<1>>(mesh, cp, prm, W, R);
            for (long long i = 0; i < n; i++) { Rv[i] = R[i].v; Rd[i] = R[i].d[0]; }
        }
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_residual: %s\n", e.what());
        return -1;
    }
}

// dR/d(patch value).tangent through the dual-number BC seeds (the product path's das_calc_dbc_product pass)
extern "C" int emu_residual_bc(const das_case_t* c, const double* Win, long long n, int patch, int field, const double* tangent, double* Rd) {
    try {
        Mesh mesh;
        mesh.build(c);
        CaseParams cp;
        cp.from_case(c);
        Options opt;
        ResParams prm = make_params(cp, opt, 0);
        PatchBC& b = mesh.bc[patch];
        if (field == 0) for (int k = 0; k < 3; k++) b.dU_val[k] = tangent[k];
        else if (field == 1) b.dp_val = tangent[0];
        else if (field == 2) b.dnuTilda_val = tangent[0];
        else b.dT_val = tangent[0];
        std::vector<Dual<1>> W(n), R(n);
        for (long long i = 0; i < n; i++) W[i] = Dual<1>(Win[i]);
        eval<Dual<1>>(mesh, cp, prm, W, R);
        for (long long i = 0; i < n; i++) Rd[i] = R[i].d[0];
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_residual_bc: %s\n", e.what());
        return -1;
    }
}

// dR/d(betaFINuTilda).tangent: the `field` input seeds of das_calc_dfield_product (tangent[nCells])
extern "C" int emu_residual_field(const das_case_t* c, const double* Win, long long n, const double* tangent, double* Rd) {
    try {
        Mesh mesh;
        mesh.build(c);
        CaseParams cp;
        cp.from_case(c);
        if (cp.beta_fi.empty()) cp.beta_fi.assign(mesh.nC, 1.0);
        cp.betaFI_ptr = cp.beta_fi.data();
        cp.dBetaFI_ptr = tangent;
        Options opt;
        ResParams prm = make_params(cp, opt, 0);
        std::vector<Dual<1>> W(n), R(n);
        for (long long i = 0; i < n; i++) W[i] = Dual<1>(Win[i]);
        eval<Dual<1>>(mesh, cp, prm, W, R);
        for (long long i = 0; i < n; i++) Rd[i] = R[i].d[0];
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_residual_field: %s\n", e.what());
        return -1;
    }
}

// dR/dX . dX with the METRICS as dual numbers: points X + eps dX -> das_geom.hpp bodies in Dual<1> -> the same kernel bodies with
// T = G = Dual<1> (states without tangents).  The exact counterpart of the finite-difference mesh products.
#include "../../dafoam_amd/csrc/das_geom.hpp"
extern "C" int emu_residual_geom(const das_case_t* c, const double* Win, long long n, const double* dX, double* Rd) {
    try {
        Mesh mesh;
        mesh.build(c);
        CaseParams cp;
        cp.from_case(c);
        if (!cp.beta_fi.empty()) cp.betaFI_ptr = cp.beta_fi.data();
        Options opt;
        ResParams prm = make_params(cp, opt, 0);
        typedef Dual<1> D;
        std::vector<D> P(3 * (size_t)mesh.nP);
        for (size_t i = 0; i < P.size(); i++) { P[i].v = mesh.points[i]; P[i].d[0] = dX[i]; }
        std::vector<FaceGeomT<D>> fg(mesh.nF);
        std::vector<CellGeomT<D>> cg(mesh.nC);
        const GeomTopo t = mesh.geom_topo();
        for (int f = 0; f < mesh.nF; f++) geom_face<D>(f, t, P.data(), fg[f]);
        for (int k = 0; k < mesh.nC; k++) { geom_cell<D>(k, t, fg.data(), cg[k]); cg[k].y = D(mesh.cg[k].y); }
        for (int f = 0; f < mesh.nF; f++) geom_weights<D>(f, t, cg.data(), fg.data(), fg[f]);
        DevMeshT<D> m;
        m.nC = mesh.nC; m.nF = mesh.nF; m.nIF = mesh.nIF;
        m.fg = fg.data(); m.cg = cg.data();
        m.cf_ptr = mesh.cf_ptr.data(); m.cf_face = mesh.cf_face.data(); m.cf_other = mesh.cf_other.data();
        m.owner = mesh.owner.data(); m.neigh = mesh.neighbour.data();
        m.bpatch = mesh.bface_patch.data(); m.bc = mesh.bc.data(); m.cyc = mesh.cyc_face.data();
        std::vector<D> W(n), R(n);
        for (long long i = 0; i < n; i++) W[i] = D(Win[i]);
        eval_on<D>(m, cp, prm, W, R);
        for (long long i = 0; i < n; i++) Rd[i] = R[i].d[0];
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_residual_geom: %s\n", e.what());
        return -1;
    }
}

// ---- SIMPLE sweeps (csrc/das_simple.hpp) in host loops: the same per-entity bodies as the device driver, serial Krylov solvers --------------
#include "../../dafoam_amd/csrc/das_simple.hpp"
namespace {
struct Ldu { const DevMeshT<double>* m; const double* diag; const double* up; const double* lo; };
static void ldu_mv(const Ldu& A, const std::vector<double>& x, std::vector<double>& y) {
    for (int c = 0; c < A.m->nC; c++) y[c] = body_ldu_row(c, *A.m, A.diag, A.up, A.lo, x.data());
}
static double vdot(const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }
// Jacobi-preconditioned BiCGStab; returns iterations
static int bicgstab(const Ldu& A, const std::vector<double>& b, std::vector<double>& x, double tol, int maxit) {
    const int n = A.m->nC;
    std::vector<double> r(n), r0(n), p(n, 0.0), v(n, 0.0), s(n), t(n), ph(n), sh(n);
    ldu_mv(A, x, r);
    for (int i = 0; i < n; i++) r[i] = b[i] - r[i];
    r0 = r;
    const double bn = std::sqrt(vdot(b, b)) + 1e-300;
    double rho = 1, alpha = 1, om = 1;
    for (int it = 1; it <= maxit; it++) {
        if (std::sqrt(vdot(r, r)) <= tol * bn) return it - 1;
        const double rho1 = vdot(r0, r);
        if (rho1 == 0.0 || !std::isfinite(rho1)) { if (getenv("EMU_SIMPLE_DEBUG")) fprintf(stderr, "bicgstab: breakdown rho %g at it %d |b| %.3e |r| %.3e\n", rho1, it, bn, std::sqrt(vdot(r, r))); return -it; }
        const double beta = (rho1 / rho) * (alpha / om);
        for (int i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - om * v[i]);
        for (int i = 0; i < n; i++) ph[i] = p[i] / A.diag[i];
        ldu_mv(A, ph, v);
        alpha = rho1 / vdot(r0, v);
        for (int i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
        if (std::sqrt(vdot(s, s)) <= tol * bn) { for (int i = 0; i < n; i++) x[i] += alpha * ph[i]; return it; }  // converged at the half step
        for (int i = 0; i < n; i++) sh[i] = s[i] / A.diag[i];
        ldu_mv(A, sh, t);
        om = vdot(t, s) / vdot(t, t);
        for (int i = 0; i < n; i++) { x[i] += alpha * ph[i] + om * sh[i]; r[i] = s[i] - om * t[i]; }
        rho = rho1;
    }
    if (getenv("EMU_SIMPLE_DEBUG")) fprintf(stderr, "bicgstab: %d iterations, relative residual %.3e\n", maxit, std::sqrt(vdot(r, r)) / bn);
    return maxit;
}
// Jacobi-preconditioned conjugate gradients on sign * A (A symmetric, sign * A positive definite: the pressure equation has sign = -1)
static int pcg(const Ldu& A, double sign, const std::vector<double>& b, std::vector<double>& x, double tol, int maxit) {
    const int n = A.m->nC;
    std::vector<double> r(n), z(n), p(n), q(n);
    ldu_mv(A, x, r);
    for (int i = 0; i < n; i++) r[i] = sign * (b[i] - r[i]);
    const double bn = std::sqrt(vdot(b, b)) + 1e-300;
    double rz = 0;
    for (int it = 1; it <= maxit; it++) {
        if (std::sqrt(vdot(r, r)) <= tol * bn) return it - 1;
        for (int i = 0; i < n; i++) z[i] = r[i] / (sign * A.diag[i]);
        const double rz1 = vdot(r, z);
        if (it == 1) p = z; else { const double beta = rz1 / rz; for (int i = 0; i < n; i++) p[i] = z[i] + beta * p[i]; }
        rz = rz1;
        ldu_mv(A, p, q);
        for (int i = 0; i < n; i++) q[i] *= sign;
        const double alpha = rz / vdot(p, q);
        for (int i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; }
    }
    if (getenv("EMU_SIMPLE_DEBUG")) fprintf(stderr, "pcg: %d iterations, relative residual %.3e\n", maxit, std::sqrt(vdot(r, r)) / bn);
    return maxit;
}
}  // namespace

// nSweeps SIMPLE iterations from Win; alphaP = explicit pressure relaxation; linTol = relative tolerance of the inner solves
extern "C" int emu_simple_iteration(const das_case_t* c, const double* Win, long long n, int nSweeps, double alphaP, double linTol, double* Wout) {
    try {
        Mesh mesh;
        mesh.build(c);
        CaseParams cp;
        cp.from_case(c);
        DAS_CHECK(cp.solver == DAS_SOLVER_SIMPLEFOAM && !cp.hasT && !cp.mrf && !cp.hasCyclic, DAS_ERR_ARG, "SIMPLE sweeps: DASimpleFoam + SA without T / MRF / cyclic pairs");
        Options opt;
        ResParams prm = make_params(cp, opt, 0);
        const DevMeshT<double> m = host_view(mesh);
        const long long N = m.nC, F = m.nF, nIF = m.nIF, nBF = F - nIF;
        std::vector<double> W(Win, Win + n), nut(N), gU(9 * N), gP(3 * N), gN(3 * N), gH(3 * N), fc(DAS_FC_N * nIF), brec(DAS_BREC_N * nBF);
        std::vector<double> D(N), bd(3 * N), sb(3 * N), rhsU(3 * N), up(nIF), lo(nIF), rAU(N), HbyA(3 * N), phiH(F), gpf(F), cpv(nIF), pbc(4 * nBF), dp(N), rp(N);
        std::vector<double> x(N), b(N), dg(N), pn(N), phiN(F), gPn(3 * N);
        for (int sw = 0; sw < nSweeps; sw++) {
            // ---- momentum predictor
            for (int cc = 0; cc < N; cc++) body_grad<double, false>(cc, m, prm, W.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data());
            for (int f = 0; f < nIF; f++) body_fcoef<double>(f, m, prm, W.data(), nut.data(), gU.data(), gN.data(), fc.data());
            for (int bb = 0; bb < nBF; bb++) body_bcoef<double>(bb, m, prm, W.data(), nut.data(), gU.data(), brec.data());
            for (int cc = 0; cc < N; cc++) body_simple_ueqn(cc, m, prm, W.data(), gP.data(), fc.data(), brec.data(), D.data(), bd.data(), sb.data(), rhsU.data());
            for (int f = 0; f < nIF; f++) body_simple_offdiag(f, m, prm, W.data(), fc.data(), 0, up.data(), lo.data());
            std::vector<double> Wn(W);
            for (int k = 0; k < 3; k++) {
                for (int cc = 0; cc < N; cc++) { dg[cc] = D[cc] + bd[3LL * cc + k]; b[cc] = rhsU[3LL * cc + k]; x[cc] = W[3LL * cc + k]; }
                bicgstab(Ldu{&m, dg.data(), up.data(), lo.data()}, b, x, linTol, 2000);
                for (int cc = 0; cc < N; cc++) Wn[3LL * cc + k] = x[cc];
            }
            // ---- pressure corrector
            for (int cc = 0; cc < N; cc++) body_simple_hbya(cc, m, Wn.data(), D.data(), bd.data(), sb.data(), up.data(), lo.data(), rAU.data(), HbyA.data());
            for (int f = 0; f < F; f++) body_simple_pface(f, m, prm, Wn.data(), nut.data(), rAU.data(), HbyA.data(), phiH.data(), gpf.data(), cpv.data(), pbc.data());
            gPn = gP;
            for (int cc = 0; cc < N; cc++) pn[cc] = W[prm.offP * N + cc];
            for (int corr = 0; corr < 2; corr++) {  // nNonOrthogonalCorrectors 1
                for (int cc = 0; cc < N; cc++) body_simple_peqn(cc, m, phiH.data(), gpf.data(), cpv.data(), pbc.data(), gPn.data(), dp.data(), rp.data());
                pcg(Ldu{&m, dp.data(), cpv.data(), cpv.data()}, -1.0, rp, pn, linTol, 20000);
                for (int cc = 0; cc < N; cc++) body_simple_gradp(cc, m, pn.data(), pbc.data(), gPn.data());
            }
            for (int f = 0; f < F; f++) body_simple_flux(f, m, pn.data(), gPn.data(), phiH.data(), gpf.data(), cpv.data(), pbc.data(), phiN.data());
            for (int cc = 0; cc < N; cc++) pn[cc] = W[prm.offP * N + cc] + alphaP * (pn[cc] - W[prm.offP * N + cc]);
            for (int cc = 0; cc < N; cc++) body_simple_gradp(cc, m, pn.data(), pbc.data(), gPn.data());
            for (int cc = 0; cc < N; cc++) {
                for (int k = 0; k < 3; k++) Wn[3LL * cc + k] = HbyA[3LL * cc + k] - rAU[cc] * gPn[3LL * cc + k];
                Wn[prm.offP * N + cc] = pn[cc];
            }
            for (int f = 0; f < F; f++) Wn[prm.offPhi * N + f] = phiN[f];
            // ---- SA transport at the updated U / phi
            for (int cc = 0; cc < N; cc++) body_grad<double, false>(cc, m, prm, Wn.data(), nut.data(), gU.data(), gP.data(), gN.data(), gH.data());
            for (int f = 0; f < nIF; f++) body_fcoef<double>(f, m, prm, Wn.data(), nut.data(), gU.data(), gN.data(), fc.data());
            for (int bb = 0; bb < nBF; bb++) body_bcoef<double>(bb, m, prm, Wn.data(), nut.data(), gU.data(), brec.data());
            for (int cc = 0; cc < N; cc++) body_simple_saeqn(cc, m, prm, Wn.data(), gU.data(), gN.data(), fc.data(), brec.data(), dg.data(), b.data());
            for (int f = 0; f < nIF; f++) body_simple_offdiag(f, m, prm, Wn.data(), fc.data(), 1, up.data(), lo.data());
            for (int cc = 0; cc < N; cc++) x[cc] = Wn[prm.offN * N + cc];
            bicgstab(Ldu{&m, dg.data(), up.data(), lo.data()}, b, x, linTol, 2000);
            for (int cc = 0; cc < N; cc++) Wn[prm.offN * N + cc] = std::max(x[cc], 1e-16);  // DAUtility::boundVar
            W.swap(Wn);
        }
        for (long long i = 0; i < n; i++) Wout[i] = W[i];
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_simple_iteration: %s\n", e.what());
        return -1;
    }
}
