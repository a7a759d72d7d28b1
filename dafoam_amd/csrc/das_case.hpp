// Case parameters shared by the device driver and the host-emulation test harness.
#pragma once
#include "das_kernels.hpp"

namespace das {

struct CaseParams {
    int solver = DAS_SOLVER_SIMPLEFOAM;
    double nu = 1.5e-5, relax_U = 0.7, relax_nuTilda = 0.7, relax_T = 1.0, DT = 0.01, deltaT = 1.0;
    double Cp = 1005.0, molWeight = 28.96, mu = 1.8e-5, Pr = 0.7, Prt = 1.0;
    int mrf = 0, transonic = 0, transonicPC = 1, hasT = 0, sutherland = 0;
    int hasCyclic = 0;  // the mesh has coupled (cyclic) patch pairs
    double As = 1.4792e-06, Ts = 116.0;
    double om[3] = {0, 0, 0}, org[3] = {0, 0, 0};
    std::vector<double> phi_frozen, T_old;
    // `field` inputs (reference DAInputField.C): betaFINuTilda per cell; the pointers are what the kernels read (device memory
    // in the library, the host vector in the test harness), null = field absent (= 1) / no tangent
    std::vector<double> beta_fi;
    const double* betaFI_ptr = nullptr;
    const double* dBetaFI_ptr = nullptr;
    void from_case(const das_case_t* c) {
        solver = c->solver;
        nu = c->nu;
        relax_U = c->relax_U;
        relax_nuTilda = c->relax_nuTilda;
        relax_T = c->relax_T;
        DT = c->DT;
        deltaT = c->deltaT;
        if (solver != DAS_SOLVER_SCALARTRANSPORTFOAM) {
            mrf = c->mrf_active != 0;
            for (int k = 0; k < 3; k++) { om[k] = c->mrf_omega[k]; org[k] = c->mrf_origin[k]; }
            DAS_CHECK(!mrf || c->patch_mrf_rotating, DAS_ERR_ARG, "MRF needs the per-patch rotating flags");
        }
        for (int p = 0; p < c->n_patches; p++) hasCyclic = hasCyclic || c->patch_type[p] == DAS_PATCH_CYCLIC;
        hasT = (solver == DAS_SOLVER_SIMPLEFOAM) && c->simple_has_T != 0;
        if (hasT) {
            Pr = c->Pr; Prt = c->Prt;
            DAS_CHECK(Pr > 0 && Prt > 0, DAS_ERR_ARG, "DASimpleFoam with a T field needs positive Pr, Prt");
            DAS_CHECK(c->bc_T_code != nullptr, DAS_ERR_ARG, "DASimpleFoam with a T field needs the T patch table");
        }
        if (DAS_IS_COMPRESSIBLE(solver)) {
            transonic = (solver == DAS_SOLVER_TURBOFOAM) && c->transonic != 0;
            transonicPC = c->transonic_pc_option;
            Cp = c->Cp; molWeight = c->molWeight; mu = c->mu; Pr = c->Pr; Prt = c->Prt;
            DAS_CHECK(Cp > 0 && molWeight > 0 && mu > 0 && Pr > 0 && Prt > 0, DAS_ERR_ARG, "DARhoSimpleFoam/DATurboFoam need positive Cp, molWeight, mu, Pr, Prt");
            sutherland = c->transport_sutherland != 0;
            if (sutherland) {
                As = c->sutherland_As; Ts = c->sutherland_Ts;
                DAS_CHECK(As > 0 && Ts >= 0, DAS_ERR_ARG, "sutherland transport needs As > 0, Ts >= 0");
            }
        }
        if (c->beta_fi_nuTilda) beta_fi.assign(c->beta_fi_nuTilda, c->beta_fi_nuTilda + c->n_cells);
        if (c->phi_frozen) phi_frozen.assign(c->phi_frozen, c->phi_frozen + c->n_faces);
        if (c->T_old) T_old.assign(c->T_old, c->T_old + c->n_cells);
        if (solver == DAS_SOLVER_SCALARTRANSPORTFOAM)
            DAS_CHECK(!phi_frozen.empty() && !T_old.empty(), DAS_ERR_ARG, "DAScalarTransportFoam needs phi_frozen and T_old");
    }
};

inline ResParams make_params(const CaseParams& cp, const Options& opt, int isPC) {
    ResParams p;
    p.nu = cp.nu;
    p.alphaU = cp.relax_U;
    p.alphaN = cp.relax_nuTilda;
    p.DT = cp.DT;
    p.deltaT = cp.deltaT;
    p.isPC = isPC;
    p.convBlend = isPC ? opt.getd("amd.pcUpwindBlend") : 1.0;
    p.constrainHbyA = (int)opt.geti("useConstrainHbyA");
    p.normU = opt.list_has("normalizeResiduals", "URes");
    p.normP = opt.list_has("normalizeResiduals", "pRes");
    p.normN = opt.list_has("normalizeResiduals", "nuTildaRes");
    p.normPhi = opt.list_has("normalizeResiduals", "phiRes");
    p.normT = opt.list_has("normalizeResiduals", "TRes");
    const bool rho = DAS_IS_COMPRESSIBLE(cp.solver) || cp.hasT;  // layouts with a T block
    p.offP = 3;
    p.offT = rho ? 4 : 0;
    p.offN = rho ? 5 : 4;
    p.offPhi = rho ? 6 : 5;
    p.hasT = cp.hasT;
    { auto it = opt.i.find("amd.gradFaceParallel"); p.gradFaceParallel = it == opt.i.end() ? 1 : (int)it->second; }
    // the face / cell split of the cell pass (body_fcoef + body_bcoef + body_cell2) serves DASimpleFoam + SA without T field, MRF and cyclic pairs
    { auto it = opt.i.find("amd.cellFaceSplit"); p.cellFaceSplit = (it == opt.i.end() ? 0 : (int)it->second) && cp.solver == DAS_SOLVER_SIMPLEFOAM && !cp.hasT && !cp.mrf && !cp.hasCyclic; }
    p.Cp = cp.Cp;
    p.Rgas = 8314.47 / cp.molWeight;  // Foam::constant::thermodynamic::RR / molWeight
    p.mu = cp.mu;
    p.Pr = cp.Pr;
    p.Prt = cp.Prt;
    p.sutherland = cp.sutherland;
    p.As = cp.As;
    p.Ts = cp.Ts;
    {
        const double Cv = p.Cp - p.Rgas;  // modified Eucken: alpha = mu Cv (1.32 + 1.77 R / Cv) / Cp
        p.alphaFac = cp.sutherland ? Cv * (1.32 + 1.77 * p.Rgas / Cv) / p.Cp : 1.0 / cp.Pr;
    }
    p.turbo = cp.solver == DAS_SOLVER_TURBOFOAM;
    p.transonic = cp.transonic;
    p.transonicPC = cp.transonicPC;
    p.mrf = cp.mrf;
    p.wTU = nullptr;
    p.betaFI = cp.betaFI_ptr;
    p.dBetaFI = cp.dBetaFI_ptr;
    for (int k = 0; k < 3; k++) { p.om[k] = cp.om[k]; p.org[k] = cp.org[k]; }
    return p;
}

inline DevMesh host_view(const Mesh& m) {
    DevMesh d;
    d.nC = m.nC; d.nF = m.nF; d.nIF = m.nIF;
    d.fg = m.fg.data(); d.cg = m.cg.data();
    d.cf_ptr = m.cf_ptr.data(); d.cf_face = m.cf_face.data(); d.cf_other = m.cf_other.data();
    d.owner = m.owner.data(); d.neigh = m.neighbour.data();
    d.bpatch = m.bface_patch.data(); d.bc = m.bc.data();
    d.cyc = m.cyc_face.data();
    return d;
}

}  // namespace das
