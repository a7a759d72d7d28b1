// Host mesh: polyMesh arrays -> finite-volume metrics and addressing.
//
// Replaces what the reference obtains from OpenFOAM's fvMesh (un-vendored dependency,
// SURVEY.md section 0.2): mesh_.Sf(), magSf(), C(), V(), surfaceInterpolation::weights(),
// nonOrthDeltaCoeffs(), nonOrthCorrectionVectors(), cells()/cellCells() used by every
// reference residual (e.g. src/adjoint/DAResidual/DAResidualSimpleFoam.C:106-237) and by the
// connectivity builder (src/adjoint/DAJacCon/DAJacCon.C:304-667).
#include <algorithm>
#include <cmath>

#include "das_common.hpp"

namespace das {

double wall_seconds() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// ---- default options: the DAOPTION surface of the hot path (reference pyDAFoam.py:116-121,334,390-392,
// 396,417,426,433,506-510,521,526-548,551-563,568-589,597-608,628)
Options::Options() {
    s["solverName"] = "DASimpleFoam";
    s["adjStateOrdering"] = "state";
    s["normalizeResiduals"] = "URes,pRes,p_rghRes,nuTildaRes,phiRes,TRes,DRes,kRes,omegaRes,epsilonRes,alpha.waterRes";
    d["adjPartDerivFDStep.State"] = 1.0e-6;
    d["normalizeStates.U"] = 1.0;
    d["normalizeStates.p"] = 1.0;
    d["normalizeStates.nuTilda"] = 1.0;
    d["normalizeStates.phi"] = 1.0;
    d["normalizeStates.T"] = 1.0;
    i["useConstrainHbyA"] = 1;
    i["adjUseColoring"] = 1;
    i["maxCorrectBCCalls"] = 2;
    i["adjPCLag"] = 10000;
    i["printInterval"] = 100;
    i["debug"] = 0;
    i["adjEqnOption.globalPCIters"] = 0;
    i["adjEqnOption.asmOverlap"] = 1;
    i["adjEqnOption.localPCIters"] = 1;
    s["adjEqnOption.jacMatReOrdering"] = "rcm";  // of the cell graph (natural | rcm are implemented)
    i["adjEqnOption.pcFillLevel"] = 1;
    i["adjEqnOption.gmresMaxIters"] = 1000;
    i["adjEqnOption.gmresRestart"] = 1000;
    d["adjEqnOption.gmresRelTol"] = 1.0e-6;
    d["adjEqnOption.gmresAbsTol"] = 1.0e-14;
    d["adjEqnOption.gmresTolDiff"] = 1.0e2;
    i["adjEqnOption.useNonZeroInitGuess"] = 0;
    i["adjEqnOption.useMGSO"] = 0;
    i["adjEqnOption.printInfo"] = 1;
    i["adjEqnOption.dynAdjustTol"] = 0;
    i["adjEqnOption.readPCMat"] = 0;
    i["maxResConLv4JacPCMat.pRes"] = 2;
    i["maxResConLv4JacPCMat.phiRes"] = 1;
    i["maxResConLv4JacPCMat.URes"] = 2;
    i["maxResConLv4JacPCMat.TRes"] = 2;
    i["maxResConLv4JacPCMat.nuTildaRes"] = 2;
    d["jacLowerBounds.dRdW"] = 1.0e-30;
    d["jacLowerBounds.dRdWPC"] = 1.0e-30;
    // MI355X-specific knobs (not in the reference)
    s["amd.pcType"] = "bilu";       // "bilu": global node-block ILU(0), sync-free sweeps (das_bilu.hpp); "ras": RAS + ILU(k) blocks in LDS
    i["amd.pcCoarseAggregates"] = -1;  // two-level PC: piecewise-constant coarse space on the pressure (-1 auto, 0 off, n aggregates)
    s["amd.pcCoarseField"] = "p";
    s["amd.pcCoarseMode"] = "additive";  // additive | deflated (A-DEF1: one extra operator product per apply)
    i["amd.coloringOnDevice"] = 1;   // serial first-fit colouring as a data-flow kernel (das_color.hpp); 0: host variants
    d["amd.primalTau0"] = 1.0;          // Newton primal: initial pseudo-time factor (diagonal scaled by 1 + 1/tau), SER growth
    d["amd.primalSERExponent"] = 1.5;   // tau = tau0 (|R0| / |R|)^exponent (measured: 1.0 -> 52+ steps, 1.5 -> 20-29, 2.0 -> 17-21 on the bench channels)
    d["amd.primalLinearTol"] = 1.0e-3;  // relative tolerance of the inner GMRES solves
    i["amd.primalLinearIters"] = 300;
    i["amd.primalPCLag"] = 3;           // Newton steps per preconditioner rebuild
    i["amd.setupThreads"] = 32;     // host threads of the block-ILU setup (page-fault bound beyond that)
    i["amd.pcBlockCells"] = 1024;   // cells per additive-Schwarz block (one workgroup each)
    i["amd.jacMode"] = 1;           // operator assembly: 1 = dual numbers
    i["amd.pcJacMode"] = 0;         // PC assembly: 0 = finite differences (reference behaviour)
    i["amd.pcFactorFP32"] = 0;      // store the ILU factors of the preconditioner in fp32 (operator stays fp64)
    i["amd.keepAssemblyMaps"] = 0;  // 1: keep the coloured-assembly maps in HBM between assemblies (adjPCLag loops on small meshes)
    i["amd.cgsAlwaysRefine"] = 0;   // 1: CGS2 every iteration; 0: refine if needed (reference default)
    i["amd.blockBatchedPC"] = 1;    // block GMRES: all right-hand sides through one pair of preconditioner sweeps (0: column by column)
    i["amd.opPackVector"] = 1;      // Krylov operator: vector-state rows packed as group rows (das_opmat.hpp)
    s["amd.coloringAlgorithm"] = "firstfit";  // device colouring: "firstfit" (serial colours, data-flow over net bitmaps) | "speculative"
    i["amd.pcCoarseGlobal"] = 1;    // multi-GPU: one global pressure coarse space (das_ksp_set_global_coarse) instead of one per rank
    // "dcgs2": classical Gram-Schmidt with DELAYED re-orthogonalisation - the second projection of step j and the first of
    // step j+1 share one pass over the basis (2 instead of 4 basis reads per iteration, same iterates); "cgs": the
    // reference's KSP_GMRES_CGS_REFINE_IFNEEDED.  adjEqnOption.useMGSO = 1 selects modified Gram-Schmidt in both cases.
    s["amd.gmresOrthogonalization"] = "dcgs2";
}
double Options::getd(const std::string& k) const {
    auto it = d.find(k);
    if (it != d.end()) return it->second;
    auto it2 = i.find(k);
    if (it2 != i.end()) return (double)it2->second;
    throw Error(DAS_ERR_ARG, "option not found: " + k);
}
long long Options::geti(const std::string& k) const {
    auto it = i.find(k);
    if (it != i.end()) return it->second;
    auto it2 = d.find(k);
    if (it2 != d.end()) return (long long)it2->second;
    throw Error(DAS_ERR_ARG, "option not found: " + k);
}
const std::string& Options::gets(const std::string& k) const {
    auto it = s.find(k);
    if (it == s.end()) throw Error(DAS_ERR_ARG, "option not found: " + k);
    return it->second;
}
bool Options::list_has(const std::string& k, const std::string& item) const {
    const std::string& v = gets(k);
    size_t pos = 0;
    while (pos <= v.size()) {
        size_t e = v.find(',', pos);
        if (e == std::string::npos) e = v.size();
        if (v.compare(pos, e - pos, item) == 0) return true;
        pos = e + 1;
    }
    return false;
}

void Mesh::build(const das_case_t* c) {
    DAS_CHECK(c && c->points && c->face_ptr && c->face_pts && c->owner, DAS_ERR_ARG, "das_case: null mesh arrays");
    nP = c->n_points; nF = c->n_faces; nIF = c->n_internal_faces; nC = c->n_cells; nPatch = c->n_patches;
    DAS_CHECK(nP > 0 && nF > 0 && nC > 0 && nIF >= 0 && nIF <= nF, DAS_ERR_ARG, "das_case: bad sizes");
    points.assign(c->points, c->points + 3 * (size_t)nP);
    face_ptr.assign(c->face_ptr, c->face_ptr + nF + 1);
    face_pts.assign(c->face_pts, c->face_pts + face_ptr[nF]);
    owner.assign(c->owner, c->owner + nF);
    neighbour.assign(c->neighbour, c->neighbour + nIF);
    patch_start.assign(c->patch_start, c->patch_start + nPatch);
    patch_size.assign(c->patch_size, c->patch_size + nPatch);
    patch_type.assign(c->patch_type, c->patch_type + nPatch);
    for (int f = 0; f < nF; f++) DAS_CHECK(owner[f] >= 0 && owner[f] < nC, DAS_ERR_ARG, "owner out of range");
    for (int f = 0; f < nIF; f++)
        DAS_CHECK(neighbour[f] > owner[f] && neighbour[f] < nC, DAS_ERR_ARG, "neighbour must exceed owner (upper-triangular order)");
    int expect = nIF;
    bface_patch.assign(nF - nIF, -1);
    bc.resize(nPatch);
    for (int p = 0; p < nPatch; p++) {
        DAS_CHECK(patch_start[p] == expect, DAS_ERR_ARG, "patches must be contiguous after internal faces");
        for (int k = 0; k < patch_size[p]; k++) bface_patch[patch_start[p] - nIF + k] = p;
        expect += patch_size[p];
        PatchBC& b = bc[p];
        b.type = patch_type[p];
        b.U_code = c->bc_U_code ? c->bc_U_code[p] : DAS_BC_ZERO_GRADIENT;
        b.p_code = c->bc_p_code ? c->bc_p_code[p] : DAS_BC_ZERO_GRADIENT;
        b.nuTilda_code = c->bc_nuTilda_code ? c->bc_nuTilda_code[p] : DAS_BC_ZERO_GRADIENT;
        b.nut_code = c->bc_nut_code ? c->bc_nut_code[p] : DAS_NUT_CALCULATED;
        b.T_code = c->bc_T_code ? c->bc_T_code[p] : DAS_BC_ZERO_GRADIENT;
        for (int k = 0; k < 3; k++) b.U_val[k] = c->bc_U_val ? c->bc_U_val[3 * p + k] : 0.0;
        b.p_val = c->bc_p_val ? c->bc_p_val[p] : 0.0;
        b.nuTilda_val = c->bc_nuTilda_val ? c->bc_nuTilda_val[p] : 0.0;
        b.T_val = c->bc_T_val ? c->bc_T_val[p] : 0.0;
        b.dU_val[0] = b.dU_val[1] = b.dU_val[2] = 0.0;
        b.dp_val = b.dnuTilda_val = b.dT_val = 0.0;
        b.mrf_included = (c->mrf_active && c->patch_mrf_rotating) ? (c->patch_mrf_rotating[p] != 0) : 0;
        b.rot = 0;
        for (int k = 0; k < 9; k++) b.Q[k] = (k % 4 == 0) ? 1.0 : 0.0;
        if (patch_type[p] == DAS_PATCH_CYCLIC && c->patch_rotation) {
            double dev = 0;
            for (int k = 0; k < 9; k++) { b.Q[k] = c->patch_rotation[9 * p + k]; dev += std::fabs(b.Q[k] - ((k % 4 == 0) ? 1.0 : 0.0)); }
            b.rot = dev > 1e-14;
        }
    }
    DAS_CHECK(expect == nF, DAS_ERR_ARG, "patches do not cover all boundary faces");
    // cyclic pairs (cyclicPolyPatch: neighbPatch, same size, face k <-> face k)
    cyc_face.assign(nF - nIF, -1);
    for (int p = 0; p < nPatch; p++) {
        if (patch_type[p] != DAS_PATCH_CYCLIC) continue;
        DAS_CHECK(c->patch_neighbour, DAS_ERR_ARG, "cyclic patch without patch_neighbour table");
        const int q = c->patch_neighbour[p];
        DAS_CHECK(q >= 0 && q < nPatch && q != p && patch_type[q] == DAS_PATCH_CYCLIC && c->patch_neighbour[q] == p, DAS_ERR_ARG,
                  "cyclic patch: neighbour patch is not its cyclic partner");
        DAS_CHECK(patch_size[q] == patch_size[p], DAS_ERR_ARG, "cyclic patch pair with different sizes");
        for (int k = 0; k < patch_size[p]; k++) cyc_face[patch_start[p] - nIF + k] = patch_start[q] + k;
    }
    compute_geometry(c->y_wall);
    build_addressing();
}

static inline void cross(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// primitiveMesh::makeFaceCentresAndAreas / makeCellCentresAndVols and
// surfaceInterpolation::makeWeights / makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors.
void Mesh::compute_geometry(const double* y_wall) {
    fg.assign(nF, FaceGeom{});
    cg.assign(nC, CellGeom{});
    for (int f = 0; f < nF; f++) {
        int b = face_ptr[f], e = face_ptr[f + 1], nv = e - b;
        FaceGeom& g = fg[f];
        const double* P = points.data();
        if (nv == 3) {
            const double *p0 = P + 3 * face_pts[b], *p1 = P + 3 * face_pts[b + 1], *p2 = P + 3 * face_pts[b + 2];
            double a[3], c2[3], n[3];
            for (int k = 0; k < 3; k++) { a[k] = p1[k] - p0[k]; c2[k] = p2[k] - p0[k]; g.Cf[k] = (p0[k] + p1[k] + p2[k]) / 3.0; }
            cross(a, c2, n);
            for (int k = 0; k < 3; k++) g.Sf[k] = 0.5 * n[k];
        } else {
            double fc[3] = {0, 0, 0};
            for (int i = b; i < e; i++) for (int k = 0; k < 3; k++) fc[k] += P[3 * face_pts[i] + k];
            for (int k = 0; k < 3; k++) fc[k] /= nv;
            double sumN[3] = {0, 0, 0}, sumA = 0, sumAc[3] = {0, 0, 0};
            for (int i = 0; i < nv; i++) {
                const double* p = P + 3 * face_pts[b + i];
                const double* q = P + 3 * face_pts[b + (i + 1) % nv];
                double u[3], v[3], n[3];
                for (int k = 0; k < 3; k++) { u[k] = q[k] - p[k]; v[k] = fc[k] - p[k]; }
                cross(u, v, n);
                double a = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int k = 0; k < 3; k++) { sumN[k] += n[k]; sumAc[k] += a * (p[k] + q[k] + fc[k]); }
                sumA += a;
            }
            for (int k = 0; k < 3; k++) { g.Cf[k] = sumAc[k] / (3.0 * sumA); g.Sf[k] = 0.5 * sumN[k]; }
        }
        g.magSf = std::sqrt(g.Sf[0] * g.Sf[0] + g.Sf[1] * g.Sf[1] + g.Sf[2] * g.Sf[2]);
    }
    // cells
    std::vector<double> cEst(3 * (size_t)nC, 0.0);
    std::vector<int> cnt(nC, 0);
    for (int f = 0; f < nF; f++) {
        for (int k = 0; k < 3; k++) cEst[3 * (size_t)owner[f] + k] += fg[f].Cf[k];
        cnt[owner[f]]++;
        if (f < nIF) { for (int k = 0; k < 3; k++) cEst[3 * (size_t)neighbour[f] + k] += fg[f].Cf[k]; cnt[neighbour[f]]++; }
    }
    for (int c = 0; c < nC; c++) for (int k = 0; k < 3; k++) cEst[3 * (size_t)c + k] /= cnt[c];
    std::vector<double> V3(nC, 0.0), Cs(3 * (size_t)nC, 0.0);
    for (int f = 0; f < nF; f++) {
        const FaceGeom& g = fg[f];
        int o = owner[f];
        double pv = 0;
        for (int k = 0; k < 3; k++) pv += g.Sf[k] * (g.Cf[k] - cEst[3 * (size_t)o + k]);
        for (int k = 0; k < 3; k++) Cs[3 * (size_t)o + k] += pv * (0.75 * g.Cf[k] + 0.25 * cEst[3 * (size_t)o + k]);
        V3[o] += pv;
        if (f < nIF) {
            int n = neighbour[f];
            double pn = 0;
            for (int k = 0; k < 3; k++) pn += g.Sf[k] * (cEst[3 * (size_t)n + k] - g.Cf[k]);
            for (int k = 0; k < 3; k++) Cs[3 * (size_t)n + k] += pn * (0.75 * g.Cf[k] + 0.25 * cEst[3 * (size_t)n + k]);
            V3[n] += pn;
        }
    }
    for (int c = 0; c < nC; c++) {
        for (int k = 0; k < 3; k++) cg[c].C[k] = Cs[3 * (size_t)c + k] / V3[c];
        cg[c].V = V3[c] / 3.0;
        cg[c].y = y_wall ? y_wall[c] : 1.0;
        DAS_CHECK(cg[c].V > 0, DAS_ERR_ARG, "non-positive cell volume");
    }
    for (int f = 0; f < nF; f++) {
        FaceGeom& g = fg[f];
        const double* Co = cg[owner[f]].C;
        if (f < nIF) {
            const double* Cn = cg[neighbour[f]].C;
            double so = 0, sn = 0, d[3], md = 0, nd = 0;
            for (int k = 0; k < 3; k++) {
                so += g.Sf[k] * (g.Cf[k] - Co[k]);
                sn += g.Sf[k] * (Cn[k] - g.Cf[k]);
                d[k] = Cn[k] - Co[k];
                md += d[k] * d[k];
                nd += g.Sf[k] / g.magSf * d[k];
            }
            so = std::fabs(so); sn = std::fabs(sn); md = std::sqrt(md);
            g.w = sn / (so + sn);
            g.nod = 1.0 / std::max(nd, 0.05 * md);
            for (int k = 0; k < 3; k++) g.corr[k] = g.Sf[k] / g.magSf - d[k] * g.nod;
        } else if (cyc_face[f - nIF] >= 0) {
            // cyclicFvPatch::makeWeights / delta() for a translational pair: the neighbour cell is the owner of the
            // paired face, seen at  Cf - Q (Cf' - C')  (its image across the pair)
            const int f2 = cyc_face[f - nIF];
            const FaceGeom& g2 = fg[f2];
            const double* C2 = cg[owner[f2]].C;
            const double* Q = bc[bface_patch[f - nIF]].Q;  // forwardT: neighbour-side vectors -> this side
            double dOwn = 0, dNbr = 0, d[3], md = 0, nd = 0;
            const double r2[3] = {g2.Cf[0] - C2[0], g2.Cf[1] - C2[1], g2.Cf[2] - C2[2]};
            for (int k = 0; k < 3; k++) {
                dOwn += g.Sf[k] / g.magSf * (g.Cf[k] - Co[k]);
                dNbr += g2.Sf[k] / g2.magSf * r2[k];
                d[k] = (g.Cf[k] - Co[k]) - (Q[3 * k] * r2[0] + Q[3 * k + 1] * r2[1] + Q[3 * k + 2] * r2[2]);
                md += d[k] * d[k];
            }
            for (int k = 0; k < 3; k++) nd += g.Sf[k] / g.magSf * d[k];
            md = std::sqrt(md);
            g.w = dNbr / (dOwn + dNbr);
            g.nod = 1.0 / std::max(nd, 0.05 * md);
            for (int k = 0; k < 3; k++) g.corr[k] = g.Sf[k] / g.magSf - d[k] * g.nod;
        } else {
            // non-coupled patch: fvPatch::delta() is the PATCH-NORMAL part of Cf - Cn (OpenFOAM v1712+, fvPatch.C "Use patch-normal
            // delta for all non-coupled BCs"), so deltaCoeffs = nonOrthDeltaCoeffs = 1 / |nf . (Cf - Cn)|; identical to 1/|Cf - Cn|
            // on orthogonal wall cells, different on sheared ones (bump, airfoil)
            double nd = 0;
            for (int k = 0; k < 3; k++) nd += g.Sf[k] / g.magSf * (g.Cf[k] - Co[k]);
            g.w = 1.0;
            g.nod = 1.0 / std::fabs(nd);
            g.corr[0] = g.corr[1] = g.corr[2] = 0.0;
        }
    }
}

void Mesh::build_addressing() {
    cf_ptr.assign(nC + 1, 0);
    for (int f = 0; f < nF; f++) { cf_ptr[owner[f] + 1]++; if (f < nIF) cf_ptr[neighbour[f] + 1]++; }
    for (int c = 0; c < nC; c++) cf_ptr[c + 1] += cf_ptr[c];
    cf_face.assign(cf_ptr[nC], 0);
    cf_other.assign(cf_ptr[nC], -1);
    std::vector<int> pos(cf_ptr.begin(), cf_ptr.end() - 1);
    // face order inside a cell: ascending face id (matches OpenFOAM cells() construction order)
    for (int f = 0; f < nF; f++) {
        int o = owner[f];
        cf_face[pos[o]] = f;
        // a cyclic boundary face has a neighbour cell too: the owner of its paired face
        cf_other[pos[o]] = f < nIF ? neighbour[f] : (cyc_face[f - nIF] >= 0 ? owner[cyc_face[f - nIF]] : -1);
        pos[o]++;
        if (f < nIF) {
            int n = neighbour[f];
            cf_face[pos[n]] = f | (int)0x80000000;
            cf_other[pos[n]] = o;
            pos[n]++;
        }
    }
    cc_ptr.assign(nC + 1, 0);
    for (int c = 0; c < nC; c++) {
        int k = 0;
        for (int s = cf_ptr[c]; s < cf_ptr[c + 1]; s++) if (cf_other[s] >= 0) k++;
        cc_ptr[c + 1] = cc_ptr[c] + k;
    }
    cc.assign(cc_ptr[nC], 0);
    std::vector<int> tmp;
    int w = 0;
    for (int c = 0; c < nC; c++) {
        tmp.clear();
        for (int s = cf_ptr[c]; s < cf_ptr[c + 1]; s++) if (cf_other[s] >= 0 && cf_other[s] != c) tmp.push_back(cf_other[s]);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());  // (a thin periodic layer reaches the same cell twice)
        cc_ptr[c] = w;
        for (int x : tmp) cc[w++] = x;
    }
    cc_ptr[nC] = w;
    cc.resize(w);
}

}  // namespace das
