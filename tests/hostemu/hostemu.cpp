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
