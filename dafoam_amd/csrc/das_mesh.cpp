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
#include "das_geom.hpp"

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
    // aggregates of the coarse space: "rcb" (recursive coordinate bisection of the cell centres, isotropic in space) | "strength"
    // (repeated pairwise matching along the strongest pressure-Laplacian coupling |Sf| / |d|: the aggregates follow the thin
    // direction of stretched cells - measured on the CPU: NACA0012 adjoint 763 -> 360 iterations where space-filling blocks give 530)
    s["amd.pcCoarseAggregation"] = "rcb";
    // PC residual: weight of the explicit second-order (linearUpwindV) correction of div(phi,U).  0 = the reference's div(pc) = upwind.  Measured
    // (round 4, NACA0012, exact LU of the PC matrix as preconditioner): the first-order / second-order mismatch IS the plateau of the
    // adjoint solve (96 iterations at 3200 cells with an exact solve of the first-order matrix, 9 with the operator on the PC pattern); an
    // incomplete factorisation tolerates a partial correction only (ILU(0): 155 -> 132 at 0.35, unstable from 0.5)
    // Round 5: the DEFAULT is 0.5 (round 4 measured it on every workload: NACA0012 wing 2 M cells - no convergence in 1000 iterations at 0,
    // 905 at 0.5; 200 x 63 section 416 -> 327; channel 191 -> 184: indifferent); 0 restores the reference's upwind div(pc).
    d["amd.pcUpwindBlend"] = 0.5;
    // additive | deflated (A-DEF1).  Round 5: "deflated" is the default - the form the wing needs to converge inside the reference's
    // 1000 / 1000 budget; its extra operator product per apply is replaced by the sparse A Z of the coarse space (coarse_az_ready:
    // ~1.2 entries per row) wherever the operator is the assembled single-rank matrix
    s["amd.pcCoarseMode"] = "deflated";
    // stability check of the incomplete factorisation (das_create_ml_rksp_matrix_free): estimate = max |(LU)^-1 P e - e| on two vectors; above
    // the limit the factorisation is rebuilt with another elimination order of the cells (0: no check)
    d["amd.pcStabilityLimit"] = 1.0e8;
    s["amd.pcOrderCandidates"] = "1245";  // elimination orders tried after the configured one (das_ksp_get_pc_stability lists them)
    d["amd.pcStabilityGood"] = 1.0e4;  // where the smallest estimate of all candidates is wanted (sub-domains): an estimate below this ends the search
    i["amd.pcSubdomains"] = -1;  // K > 1: restricted additive Schwarz inside the GPU (K node-block ILUs on RCB blocks + asmOverlap rings, own elimination orders, one merged level structure); -1 (default): 4 from 1 M cells on, else 1
    i["amd.pcCoarseSparseAZ"] = 1;   // deflated mode: A (Z u) through the precomputed sparse A Z (0: one full operator product per apply)
    i["amd.coloringOnDevice"] = 1;   // serial first-fit colouring as a data-flow kernel (das_color.hpp); 0: host variants
    // PYDAFOAM.solvePrimal: "newton" (default: pseudo-transient Newton-Krylov, das_solve_primal) | "simple" (the reference's own loop: SIMPLE sweeps on
    // the device, das_simple_iteration, in blocks of simpleSweepsPerCheck sweeps between residual checks; inner solvers to simpleLinearTol)
    s["amd.primalMethod"] = "newton";
    i["amd.simpleSweepsPerCheck"] = 10;
    d["amd.simpleAlphaP"] = 0.3;
    d["amd.simpleLinearTol"] = 1.0e-6;
    i["amd.simpleLinearIters"] = 2000;
    d["amd.primalTau0"] = 1.0;          // Newton primal: initial pseudo-time factor (diagonal scaled by 1 + 1/tau), SER growth
    d["amd.primalSERExponent"] = 1.5;   // tau = tau0 (|R0| / |R|)^exponent (measured: 1.0 -> 52+ steps, 1.5 -> 20-29, 2.0 -> 17-21 on the bench channels)
    // pseudo-time control: "ser" (tau = tau0 (|R0|/|R|)^p: starts close to the solution) | "ramp" (CFL ramp: tau grows by >= primalTauGrowth
    // per accepted full step, by the residual drop^p if larger (<= primalTauGrowthMax), shrinks with damped / rejected steps: cold starts)
    s["amd.primalPseudoTimeFields"] = "all";  // rows that get the pseudo-time term: "all" | "momentum" (U, T, nuTilda only: cold starts, see run_newton_primal)
    s["amd.primalTauMode"] = "ser";
    d["amd.primalTauGrowth"] = 1.5;
    d["amd.primalTauGrowthMax"] = 10.0;
    d["amd.primalTauMax"] = 1.0e12;
    d["amd.primalTauMin"] = 1.0e-3;
    d["amd.primalAcceptFactor"] = 1.5;      // a step may raise |R| by at most this factor
    s["amd.primalDampedSteps"] = "accept";  // ramp mode: "accept" a damped update and shrink tau with it | "reject" it, halve tau, recompute
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
    // device colouring: "firstfit" (serial colours, data-flow over net bitmaps) | "speculative" (order-independent rounds, ~20 % more colours) |
    // "auto" (default): first-fit unless the cell numbering has no wavefront parallelism (estimated dependency depth, ensure_coloring)
    s["amd.coloringAlgorithm"] = "auto";
    s["amd.volCoordMode"] = "dual";  // mesh-sensitivity product: "dual" (exact: Dual<1> points -> metrics -> residual) | "fd" (coloured central differences)
    i["amd.volCoordRings"] = 3;     // mesh-sensitivity product: face-neighbour rings a point's influence is followed over (the deepest stencil table)
    d["amd.volCoordRelStep"] = 1e-4;  // ... central-difference step of a point, relative to the smallest adjacent cell thickness
    i["amd.pcCoarseGlobal"] = 1;    // multi-GPU: one global pressure coarse space (das_ksp_set_global_coarse) instead of one per rank
    // "dcgs2": classical Gram-Schmidt with DELAYED re-orthogonalisation - the second projection of step j and the first of
    // step j+1 share one pass over the basis (2 instead of 4 basis reads per iteration, same iterates); "cgs": the
    // reference's KSP_GMRES_CGS_REFINE_IFNEEDED.  adjEqnOption.useMGSO = 1 selects modified Gram-Schmidt in both cases.
    s["amd.gmresOrthogonalization"] = "dcgs2";
    // > 0: GMRES with deflated restarting (GMRES-DR): gmresRestart basis vectors, this many harmonic Ritz vectors carried across restarts
    // (round 4: prototyped on the CPU, tools/gmres_dr_study.py; the device path has not been measured yet - opt-in, default off)
    i["amd.gmresDeflation"] = 0;
    // storage type of the Krylov basis: "fp64" | "split" (hi + lo floats: the inner-product pass of the delayed re-orthogonalisation reads
    // only the hi array - 12 instead of 16 bytes per basis entry and iteration - every vector-building pass reads hi + lo: Arnoldi relation to
    // 2^-48) | "fp32" (compressed basis, short well-conditioned solves only) | "auto" (default): split for bases >= 1 GB with dcgs2, else fp64
    s["amd.krylovBasisPrecision"] = "auto";
    // DASimpleFoam cell pass as per-face coefficient passes + a per-cell gather pass (k_fcoef, k_bcoef, k_cell2: every internal face evaluated once
    // instead of twice).  Measured at 2 M cells (profiles/r07n_*): dual numbers 489 + 21 + 938 us against 1100 us for the monolithic k_cell, fp64
    // 317 + 30 + 484 against ~700: the gather pass alone costs what the monolith costs - the time is in the per-cell gathers, not in the face
    // arithmetic the split removes.  Off by default (1: on).
    i["amd.cellFaceSplit"] = 0;
    i["amd.gradFaceParallel"] = 1;  // DASimpleFoam gradients by the face-parallel, LDS-staged kernel k_grad_fp: 1 = where it is the faster one (fp64 passes), 2 = dual passes too, 0 = never
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
    build_addressing();
    compute_geometry(c->y_wall);
}

// primitiveMesh::makeFaceCentresAndAreas / makeCellCentresAndVols and surfaceInterpolation::makeWeights /
// makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors: three passes over the per-entity bodies of das_geom.hpp (the device
// kernels of the mesh-sensitivity product run the same bodies).  Needs build_addressing() (cell -> faces lists).
GeomTopo Mesh::geom_topo() const {
    GeomTopo t;
    t.nC = nC; t.nF = nF; t.nIF = nIF;
    t.face_ptr = face_ptr.data(); t.face_pts = face_pts.data();
    t.owner = owner.data(); t.neigh = neighbour.data();
    t.cf_ptr = cf_ptr.data(); t.cf_face = cf_face.data();
    t.bpatch = bface_patch.data(); t.cyc = cyc_face.data(); t.bc = bc.data();
    return t;
}
void Mesh::compute_geometry(const double* y_wall) {
    DAS_CHECK((int)cf_ptr.size() == nC + 1, DAS_ERR_INTERNAL, "compute_geometry needs the cell -> faces addressing");
    fg.assign(nF, FaceGeom{});
    cg.assign(nC, CellGeom{});
    const GeomTopo t = geom_topo();
    const double* P = points.data();
    const int nthr = host_threads();
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int f = 0; f < nF; f++) geom_face(f, t, P, fg[f]);
    long long bad = 0;
#pragma omp parallel for schedule(static) num_threads(nthr) reduction(+ : bad)
    for (int c = 0; c < nC; c++) {
        const bool ok = geom_cell(c, t, fg.data(), cg[c]);
        cg[c].y = y_wall ? y_wall[c] : 1.0;
        if (!ok) bad++;
    }
    DAS_CHECK(bad == 0, DAS_ERR_ARG, "non-positive cell volume");
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int f = 0; f < nF; f++) geom_weights(f, t, cg.data(), fg.data(), fg[f]);
}

// fvMesh metrics of a bare polyhedral mesh (no case, no solver handle): what the synthetic-input generators need (face area vectors and
// centres, cell centres and volumes, interpolation weights) - the per-entity bodies of das_geom.hpp over all host threads
void mesh_metrics_only(int nP, const double* pts, int nF, int nIF, int nC, const int* fptr, const int* fpts, const int* own, const int* nei,
                       double* Sf, double* Cf, double* C, double* V, double* w) {
    Mesh m;
    m.nP = nP; m.nF = nF; m.nIF = nIF; m.nC = nC; m.nPatch = 0;
    m.points.assign(pts, pts + 3LL * nP);
    m.face_ptr.assign(fptr, fptr + nF + 1);
    m.face_pts.assign(fpts, fpts + fptr[nF]);
    m.owner.assign(own, own + nF);
    m.neighbour.assign(nei, nei + nIF);
    m.bface_patch.assign(nF - nIF, 0);
    m.cyc_face.assign(nF - nIF, -1);
    m.bc.assign(1, PatchBC{});
    m.build_addressing();
    m.compute_geometry(nullptr);
    const int nthr = host_threads();
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int f = 0; f < nF; f++) {
        for (int k = 0; k < 3; k++) { Sf[3LL * f + k] = m.fg[f].Sf[k]; Cf[3LL * f + k] = m.fg[f].Cf[k]; }
        if (f < nIF && w) w[f] = m.fg[f].w;
    }
#pragma omp parallel for schedule(static) num_threads(nthr)
    for (int c = 0; c < nC; c++) { for (int k = 0; k < 3; k++) C[3LL * c + k] = m.cg[c].C[k]; V[c] = m.cg[c].V; }
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

// ---- points: who feels a point, and which points may move together ------------------------------------------------------
// The residual rows of a cell (its cell-centred rows and the phi rows of the faces it owns) reach at most `rings` face-neighbour
// rings (pRes: 3, the deepest table of the reference, DAStateInfo*.C); a metric of a cell or face of that stencil depends on the
// points of the cell / of the two cells of the face.  Hence: the rows that can change when point p moves belong to the cells
// within `rings` rings of the cells touching p.  Two points whose sets are disjoint can be perturbed in the same residual pass:
// a greedy first-fit colouring over per-cell colour bitmaps (the structure of the Jacobian colouring, with points as columns
// and cells as nets) groups them.
// central-difference step of every point: relStep x the smallest thickness V / max |Sf| of the cells at the point
void point_steps(const Mesh& m, double relStep, std::vector<double>& h) {
    std::vector<double> thick(m.nC);
    for (int c = 0; c < m.nC; c++) {
        double amax = 0.0;
        for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) amax = std::max(amax, m.fg[m.cf_face[s] & 0x7fffffff].magSf);
        thick[c] = m.cg[c].V / amax;
    }
    h.assign(m.nP, 1e300);
    for (int f = 0; f < m.nF; f++) {
        const double l = f < m.nIF ? std::min(thick[m.owner[f]], thick[m.neighbour[f]]) : thick[m.owner[f]];
        for (int i = m.face_ptr[f]; i < m.face_ptr[f + 1]; i++) h[m.face_pts[i]] = std::min(h[m.face_pts[i]], relStep * l);
    }
    for (int p = 0; p < m.nP; p++) if (h[p] == 1e300) h[p] = 0.0;
}
void build_point_influence(const Mesh& m, int rings, int threads, PointInfluence& out) {
    const int nP = m.nP, nC = m.nC;
    out.rings = rings;
    // point -> touching cells
    std::vector<long long> pcPtr((size_t)nP + 1, 0);
    for (int f = 0; f < m.nF; f++)
        for (int i = m.face_ptr[f]; i < m.face_ptr[f + 1]; i++) pcPtr[m.face_pts[i] + 1] += f < m.nIF ? 2 : 1;
    for (int p = 0; p < nP; p++) pcPtr[p + 1] += pcPtr[p];
    std::vector<int> pc(pcPtr[nP]);
    {
        std::vector<long long> pos(pcPtr.begin(), pcPtr.end() - 1);
        for (int f = 0; f < m.nF; f++)
            for (int i = m.face_ptr[f]; i < m.face_ptr[f + 1]; i++) {
                const int p = m.face_pts[i];
                pc[pos[p]++] = m.owner[f];
                if (f < m.nIF) pc[pos[p]++] = m.neighbour[f];
            }
    }
    threads = das::host_threads(threads);
    std::vector<std::vector<int>> lists(nP);
#pragma omp parallel num_threads(threads)
    {
        std::vector<int> stamp(nC, -1), cur, nxt, all;
#pragma omp for schedule(dynamic, 256)
        for (int p = 0; p < nP; p++) {
            cur.clear(); all.clear();
            for (long long q = pcPtr[p]; q < pcPtr[p + 1]; q++) {
                const int c = pc[q];
                if (stamp[c] == p) continue;
                stamp[c] = p; cur.push_back(c); all.push_back(c);
            }
            for (int r = 0; r < rings; r++) {
                nxt.clear();
                for (int c : cur)
                    for (int q = m.cc_ptr[c]; q < m.cc_ptr[c + 1]; q++) {
                        const int o = m.cc[q];
                        if (stamp[o] == p) continue;
                        stamp[o] = p; nxt.push_back(o); all.push_back(o);
                    }
                cur.swap(nxt);
            }
            std::sort(all.begin(), all.end());
            lists[p] = all;
        }
    }
    out.ptr.assign((size_t)nP + 1, 0);
    for (int p = 0; p < nP; p++) out.ptr[p + 1] = out.ptr[p] + (long long)lists[p].size();
    out.cells.resize(out.ptr[nP]);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int p = 0; p < nP; p++) {
        std::copy(lists[p].begin(), lists[p].end(), out.cells.begin() + out.ptr[p]);
        std::vector<int>().swap(lists[p]);
    }
    // serial first-fit: colour(p) = smallest colour no cell of its set has seen yet
    int W = 4;  // 64-bit words per cell bitmap, grown on demand
    std::vector<unsigned long long> bits((size_t)nC * W, 0ull);
    out.color.assign(nP, -1);
    out.nColors = 0;
    std::vector<unsigned long long> acc;
    for (int p = 0; p < nP; p++) {
        if (out.ptr[p + 1] == out.ptr[p]) continue;  // a point no face uses
        acc.assign(W, 0ull);
        for (long long q = out.ptr[p]; q < out.ptr[p + 1]; q++) {
            const unsigned long long* b = &bits[(size_t)out.cells[q] * W];
            for (int w = 0; w < W; w++) acc[w] |= b[w];
        }
        int col = -1;
        for (int w = 0; w < W && col < 0; w++)
            if (~acc[w]) col = 64 * w + __builtin_ctzll(~acc[w]);
        if (col < 0) {  // all 64 W colours taken around this point: widen the bitmaps
            const int W2 = 2 * W;
            std::vector<unsigned long long> nb((size_t)nC * W2, 0ull);
            for (int c = 0; c < nC; c++) std::copy(&bits[(size_t)c * W], &bits[(size_t)c * W] + W, &nb[(size_t)c * W2]);
            bits.swap(nb);
            col = 64 * W;
            W = W2;
        }
        out.color[p] = col;
        out.nColors = std::max(out.nColors, col + 1);
        for (long long q = out.ptr[p]; q < out.ptr[p + 1]; q++) bits[(size_t)out.cells[q] * W + (col >> 6)] |= 1ull << (col & 63);
    }
    out.cptr.assign(out.nColors + 1, 0);
    for (int p = 0; p < nP; p++) if (out.color[p] >= 0) out.cptr[out.color[p] + 1]++;
    for (int c = 0; c < out.nColors; c++) out.cptr[c + 1] += out.cptr[c];
    out.cpoints.resize(out.cptr[out.nColors]);
    std::vector<int> pos(out.cptr.begin(), out.cptr.end() - 1);
    for (int p = 0; p < nP; p++) if (out.color[p] >= 0) out.cpoints[pos[out.color[p]]++] = p;
}

// ---- strength-of-connection aggregates for the pressure coarse space ---------------------------------------------------------
// The coupling of two face-neighbour cells in the pressure Laplacian is ~ |Sf| nonOrthDeltaCoeff (area over distance).  Every
// pass matches each aggregate with its strongest unmatched neighbour (sequential sweep in index order: deterministic) and sums
// the couplings of the merged pairs; passes continue until at most `maxAgg` aggregates are left or nothing can be merged.
int strength_aggregates(const Mesh& m, const std::vector<unsigned char>* ownedCell, int maxAgg, std::vector<int>& agg) {
    const int N = m.nC;
    agg.assign(N, -1);
    // owned cells -> compact ids
    int cnt = 0;
    for (int c = 0; c < N; c++) if (!ownedCell || (*ownedCell)[c]) agg[c] = cnt++;
    if (cnt == 0) return 0;
    struct Edge { int u, v; double w; };
    std::vector<Edge> edges;
    edges.reserve((size_t)m.nF);
    for (int f = 0; f < m.nF; f++) {
        int o = m.owner[f], nb = -1;
        if (f < m.nIF) nb = m.neighbour[f];
        else if (m.cyc_face[f - m.nIF] >= 0) { nb = m.owner[m.cyc_face[f - m.nIF]]; if (nb < o) continue; }  // one edge per pair
        if (nb < 0 || agg[o] < 0 || agg[nb] < 0 || o == nb) continue;
        edges.push_back({agg[o], agg[nb], m.fg[f].magSf * m.fg[f].nod});
    }
    std::vector<int> cur(cnt);  // aggregate id of every compact cell
    for (int i = 0; i < cnt; i++) cur[i] = i;
    int nA = cnt;
    std::vector<int> match, best;
    std::vector<double> bw;
    std::vector<long long> ptr;
    std::vector<int> adj;
    std::vector<double> adjw;
    while (nA > maxAgg && !edges.empty()) {
        // adjacency of the current aggregate graph
        ptr.assign((size_t)nA + 1, 0);
        for (const Edge& e : edges) { ptr[e.u + 1]++; ptr[e.v + 1]++; }
        for (int i = 0; i < nA; i++) ptr[i + 1] += ptr[i];
        adj.resize(ptr[nA]); adjw.resize(ptr[nA]);
        {
            std::vector<long long> pos(ptr.begin(), ptr.end() - 1);
            for (const Edge& e : edges) { adj[pos[e.u]] = e.v; adjw[pos[e.u]++] = e.w; adj[pos[e.v]] = e.u; adjw[pos[e.v]++] = e.w; }
        }
        match.assign(nA, -1);
        int nNew = 0;
        for (int i = 0; i < nA; i++) {
            if (match[i] >= 0) continue;
            int bj = -1;
            double bwv = 0.0;
            for (long long q = ptr[i]; q < ptr[i + 1]; q++) {
                const int j = adj[q];
                if (match[j] >= 0 || j == i) continue;
                if (adjw[q] > bwv || (adjw[q] == bwv && bj >= 0 && j < bj)) { bwv = adjw[q]; bj = j; }
            }
            match[i] = nNew;
            if (bj >= 0) match[bj] = nNew;
            nNew++;
        }
        if (nNew == nA) break;  // nothing merged
        for (int i = 0; i < cnt; i++) cur[i] = match[cur[i]];
        // coarse edges: relabel, drop self loops, merge duplicates
        for (Edge& e : edges) { int a = match[e.u], b = match[e.v]; if (a > b) std::swap(a, b); e.u = a; e.v = b; }
        edges.erase(std::remove_if(edges.begin(), edges.end(), [](const Edge& e) { return e.u == e.v; }), edges.end());
        std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) { return a.u != b.u ? a.u < b.u : a.v < b.v; });
        size_t w = 0;
        for (size_t r = 0; r < edges.size(); r++) {
            if (w > 0 && edges[w - 1].u == edges[r].u && edges[w - 1].v == edges[r].v) edges[w - 1].w += edges[r].w;
            else edges[w++] = edges[r];
        }
        edges.resize(w);
        nA = nNew;
    }
    for (int c = 0; c < N; c++) if (agg[c] >= 0) agg[c] = cur[agg[c]];
    return nA;
}

}  // namespace das
