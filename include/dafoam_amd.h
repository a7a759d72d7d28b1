/*
 * dafoam_amd.h - C-ABI of the MI355X-native discrete-adjoint hot path.
 *
 * Drop-in boundary: these entry points are what the reference's Cython layer
 * (reference src/pyDASolvers/pyDASolvers.pyx:45-114 extern block, :117-482 class
 * pyDASolvers, wrapping src/pyDASolvers/DASolvers.H) would bind for the adjoint
 * path.  Conventions follow the reference: caller-owned
 * 1-D contiguous float64 buffers borrowed for the call and written in place;
 * strings are NUL-terminated char*; soft failures are integer returns; hard
 * failures return a negative code and set das_last_error() (the reference
 * aborts the process via OpenFOAM FatalError, e.g. DASolver.C:992-993).
 * No exceptions, no C++/torch types cross this boundary.
 *
 * PETSc objects of the reference API (Mat dRdWT / Mat dRdWTPC / KSP, created by the
 * Python caller in reference dafoam/mphys/mphys_dafoam.py:468-475,519-529) are
 * replaced by opaque device-resident handles das_mat_t / das_ksp_t.
 *
 * All compute entry points run on the GPU; they FAIL (return DAS_ERR_NO_DEVICE)
 * when no HIP device is present - there is no CPU fallback.
 */
#ifndef DAFOAM_AMD_H
#define DAFOAM_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define DAS_OK 0
#define DAS_ERR_ARG -1
#define DAS_ERR_NO_DEVICE -2
#define DAS_ERR_HIP -3
#define DAS_ERR_STATE -4
#define DAS_ERR_INTERNAL -5

/* solver ids (reference run-time selection by "solverName", DASolver.C:122-152) */
#define DAS_SOLVER_SIMPLEFOAM 0          /* DASimpleFoam + SpalartAllmaras */
#define DAS_SOLVER_SCALARTRANSPORTFOAM 1 /* DAScalarTransportFoam */
#define DAS_SOLVER_RHOSIMPLEFOAM 2       /* DARhoSimpleFoam + SpalartAllmaras (perfect gas, hConst, const transport) */
#define DAS_SOLVER_TURBOFOAM 3          /* DATurboFoam + SpalartAllmaras: SIMPLEC-consistent / transonic pEqn, "h" energy, MRF */
#define DAS_IS_COMPRESSIBLE(s) ((s) == DAS_SOLVER_RHOSIMPLEFOAM || (s) == DAS_SOLVER_TURBOFOAM)

/* patch types */
#define DAS_PATCH_PATCH 0
#define DAS_PATCH_WALL 1
#define DAS_PATCH_SYMMETRY 2
#define DAS_PATCH_CYCLIC 3 /* coupled pair (translational or rotational): face k of the patch pairs with face k of patch_neighbour */

/* boundary-condition codes per patch and field */
#define DAS_BC_FIXED_VALUE 0
#define DAS_BC_ZERO_GRADIENT 1
#define DAS_BC_INLET_OUTLET 2
#define DAS_BC_SYMMETRY 3
#define DAS_BC_CYCLIC 4 /* placeholder code of coupled patches: no patch-field coefficients are evaluated */
/* nut patch treatment (reference DAField.C:1155-1218, DAMisc/nutUSpaldingWallFunctionDF) */
#define DAS_NUT_CALCULATED 0
#define DAS_NUT_LOWRE_WALL 1
#define DAS_NUT_SPALDING_WALL 2
#define DAS_NUT_SYMMETRY 3

typedef struct das_solver das_solver_t; /* replaces Foam::DASolvers (DASolvers.H) */
typedef struct das_mat das_mat_t;       /* replaces PETSc Mat dRdWT / dRdWTPC */
typedef struct das_ksp das_ksp_t;       /* replaces PETSc KSP */

/* What the reference reads from the OpenFOAM case directory (constant/polyMesh, 0/,
 * constant/transportProperties, system/fvSolution) for this path, as plain arrays. */
typedef struct das_case {
    int solver; /* DAS_SOLVER_* */
    int n_points, n_faces, n_internal_faces, n_cells, n_patches;
    const double* points;  /* 3*n_points */
    const int* face_ptr;   /* n_faces+1 (CSR into face_pts) */
    const int* face_pts;   /* point ids */
    const int* owner;      /* n_faces */
    const int* neighbour;  /* n_internal_faces */
    const int* patch_start; /* n_patches, absolute face index */
    const int* patch_size;
    const int* patch_type; /* DAS_PATCH_* */
    /* per-patch BC tables; *_val: 3 doubles per patch for U, 1 otherwise */
    const int* bc_U_code;
    const double* bc_U_val;
    const int* bc_p_code;
    const double* bc_p_val;
    const int* bc_nuTilda_code;
    const double* bc_nuTilda_val;
    const int* bc_nut_code;
    const int* bc_T_code;
    const double* bc_T_val;
    double nu;
    double relax_U, relax_nuTilda, relax_T;
    double DT, deltaT;        /* DAScalarTransportFoam */
    const double* y_wall;     /* n_cells, frozen wall distance (may be NULL for ScalarTransport) */
    const double* phi_frozen; /* n_faces, DAScalarTransportFoam only */
    const double* T_old;      /* n_cells, DAScalarTransportFoam only */
    /* DARhoSimpleFoam thermophysicalProperties (reference DAResidual.C:179-293): Cp [J/kg/K], molWeight [kg/kmol],
     * mu [Pa s], Pr, Prt */
    double Cp, molWeight, mu, Pr, Prt;
    /* constant/MRFProperties, one zone covering the mesh (OpenFOAM MRFZone; used by DARhoSimpleFoam and DATurboFoam,
     * reference DAResidualRhoSimpleFoam.C:123,186, DAResidualTurboFoam.C:107,130,161,200): angular velocity vector
     * [rad/s], origin, and per patch 1 = the patch rotates with the zone (not listed in nonRotatingPatches) */
    int mrf_active;
    double mrf_omega[3], mrf_origin[3];
    const int* patch_mrf_rotating; /* n_patches, may be NULL when mrf_active == 0 */
    /* DATurboFoam: system/fvSolution SIMPLE.transonic and the reference option transonicPCOption (-1/0: keep
     * div(phid,p) in the PC, 1: drop it, 2: additionally phiRes = phi) */
    int transonic, transonic_pc_option;
    /* DASimpleFoam with the optional passive T field (reference DAResidualSimpleFoam.C:50-76,215-235; states
     * [U | p | T | nuTilda | phi], DAStateInfoSimpleFoam.C:118-131); Pr / Prt above are then transportProperties' */
    int simple_has_T;
    /* cyclic (coupled) patches, OpenFOAM cyclicFvPatch semantics: per patch the index of the paired patch (-1 for
     * ordinary patches; may be NULL if no patch is cyclic) and, for rotational pairs, the tensor forwardT (row-major 3x3
     * per patch) that carries neighbour-side vectors into this side's frame (cyclicPolyPatch::forwardT; identity /
     * NULL for translational pairs; the partner patch holds the transpose) */
    const int* patch_neighbour;
    const double* patch_rotation;
    /* thermophysicalProperties transport "sutherland" (reference DAResidual::updateThermoVars, DAResidual.C:264-293):
     * mu = As sqrt(T) / (1 + Ts / T), alpha = mu Cv (1.32 + 1.77 R / Cv) / Cp; 0 = "const" transport (mu, Pr above) */
    int transport_sutherland;
    double sutherland_As, sutherland_Ts;
    /* betaFINuTilda (n_cells, may be NULL = 1): the field-inversion multiplier of the Spalart-Allmaras production term
     * (reference DASpalartAllmaras.C betaFINuTilda_), the volScalarField a `field` input assigns (DAInputField.C:88-110) */
    const double* beta_fi_nuTilda;
} das_case_t;

const char* das_last_error(void);
int das_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int das_device_count(void);

/* ---- construction / options ------------------------------------------------------------
 * das_create      <- pyDASolvers.__init__(argsAll, pyOptions)   pyDASolvers.pyx:134-153
 * das_init_solver <- pyDASolvers.initSolver()                    pyDASolvers.pyx:155 (DASolver::initSolver)
 * das_set_option_* <- pyDASolvers.updateDAOption(pyOptions)      pyDASolvers.pyx:355 (flattened "a.b" keys of DAOPTION,
 *                     reference dafoam/pyDAFoam.py:39-661; lists are comma-joined strings) */
das_solver_t* das_create(const das_case_t* c);
void das_destroy(das_solver_t* s);
int das_set_option_double(das_solver_t* s, const char* key, double v);
int das_set_option_int(das_solver_t* s, const char* key, long long v);
int das_set_option_str(das_solver_t* s, const char* key, const char* v);
int das_get_option_double(das_solver_t* s, const char* key, double* v);
int das_init_solver(das_solver_t* s, int device);

/* ---- sizes: getNLocalAdjointStates/getNLocalCells/getNGlobalCells/getNLocalPoints  pyDASolvers.pyx:305-317 */
long long das_get_n_local_adjoint_states(das_solver_t* s);
long long das_get_n_local_cells(das_solver_t* s);
long long das_get_n_global_cells(das_solver_t* s);
int das_set_n_global_cells(das_solver_t* s, long long nGlobal); /* sharded runs: set by the partitioner */
long long das_get_n_local_points(das_solver_t* s);
long long das_get_n_local_faces(das_solver_t* s);

/* das_update_of_mesh <- updateOFMesh(vol_coords)  pyDASolvers.pyx:297-300 (PYDAFOAM.setVolCoords, pyDAFoam.py:2111-2117):
 *   new point coordinates (3 * n_points); metrics are recomputed and re-uploaded, the wall distance stays frozen.
 * das_get_of_mesh_points <- getOFMeshPoints  pyDASolvers.pyx:278 */
int das_update_of_mesh(das_solver_t* s, const double* points);
int das_get_of_mesh_points(das_solver_t* s, double* points);
/* das_calc_dvolcoord_product <- calcJacTVecProduct(inputType "volCoord" -> outputType "residual" | "function")
 *   pyDASolvers.pyx:333-366 -> DASolver.C:1690-1839 with DAInput/DAInputVolCoord.C:33-70: the FULL product vector
 *   product[3 p + k] = sum_i seeds[i] dOutput_i/dX[p][k] over all mesh points, at the current states and points.  The reference
 *   gets it from one reverse sweep of its AD tape; here coloured forward-mode passes run entirely on the device: the points of a
 *   colour carry a unit tangent, metrics and residual follow as Dual<1> (exact; option amd.volCoordMode "fd": central
 *   differences; csrc/das_volcoord.hpp).  seeds: n states (residual) or 1 (function:
 *   any defined face function).  info4 (may be NULL) = {point colours, residual passes, seconds, build seconds}.
 * das_point_influence_build / _get: the host-side structure behind it (no GPU needed): point colours, the cells whose residual
 *   rows feel a point (CSR), the finite-difference step per point.
 * das_debug_device_geometry: the metrics the device passes produce for `points` (records of 12 / 5 doubles: FaceGeom, CellGeom of
 *   csrc/das_common.hpp), solver geometry untouched - test aid. */
int das_calc_dvolcoord_product(das_solver_t* s, const char* outputName, const char* outputType, const double* seeds, double* product /*3P*/,
                               double* info4);
int das_point_influence_build(das_solver_t* s, int* nColors, long long* nEntries);
int das_point_influence_get(das_solver_t* s, int* colors /*P*/, long long* ptr /*P+1*/, int* cells /*nEntries*/, double* steps /*P*/);
int das_debug_device_geometry(das_solver_t* s, const double* points /*3P*/, double* fg12 /*12F*/, double* cg5 /*5N*/);
/* das_debug_strength_aggregates: the aggregates of the pressure coarse space under option amd.pcCoarseAggregation "strength"
 *   (repeated pairwise matching along the strongest |Sf| / |d| coupling until at most maxAgg are left); host only - test aid. */
int das_debug_strength_aggregates(das_solver_t* s, int maxAgg, int* agg /*N*/, int* nAgg);
/* ---- host-side mesh geometry (fvMesh metrics; no GPU needed) - used by tests and input generators */
int das_get_geometry(das_solver_t* s, double* Sf /*3F*/, double* Cf /*3F*/, double* C /*3N*/, double* V /*N*/,
                     double* weights /*Fi*/, double* nonOrthDeltaCoeffs /*Fi*/, double* nonOrthCorr /*3Fi*/,
                     double* bDeltaCoeffs /*Fb*/);

/* fvMesh metrics of a bare polyhedral mesh (no case, no solver handle, no GPU): Sf / Cf per face, C / V per cell, linear interpolation
 * weights per internal face (may be NULL) - the synthetic-input generators of the bench use it at 2 M cells (numpy: 30 s) */
int das_mesh_metrics(int nPoints, const double* points, int nFaces, int nInternalFaces, int nCells, const int* facePtr, const int* facePts, const int* owner,
                     const int* neighbour, double* Sf /*3F*/, double* Cf /*3F*/, double* C /*3N*/, double* V /*N*/, double* weights /*Fi*/);

/* ---- state / residual access -------------------------------------------------------------
 * das_update_of_fields <- updateOFFields(states)   pyDASolvers.pyx:268   (DAField::stateVec2OFField)
 * das_get_of_fields    <- getOFFields(states)      pyDASolvers.pyx:273
 * das_get_residuals    <- getResiduals(residuals)  pyDASolvers.pyx:184   (DASolver.C:1157-1236; isPC=0)
 * das_calc_residuals   : same with explicit isPC (DAResidual::masterFunction, DAResidual.C:100-171) */
int das_update_of_fields(das_solver_t* s, const double* states);
int das_get_of_fields(das_solver_t* s, double* states);
int das_get_residuals(das_solver_t* s, double* residuals);
int das_calc_residuals(das_solver_t* s, int isPC, double* residuals);

/* ---- connectivity + colouring ------------------------------------------------------------
 * das_run_coloring <- runColoring()  pyDASolvers.pyx:161  (DASolver.C:708-743, DAJacCon, DAColoring)
 * Host graph work; needs no GPU.  The getters expose dRdWCon and the colour vector (the reference
 * writes them as dRdWCon.bin / dRdWColoring_<np>.bin, DAJacCon.C:1886-1975,2580-2586). */
/* das_solve_primal <- solvePrimal()  pyDASolvers.pyx (DASimpleFoam::solvePrimal, DASimpleFoam.C:123-185): converge the
 * residuals of the current states.  The reference iterates SIMPLE; its fixed point is R(W) = 0, which is solved here by a
 * pseudo-transient Newton-Krylov method built from the adjoint's own kernels (forward-mode operator, transposed node-block
 * ILU + coarse space; options amd.primalTau0 / primalLinearTol / primalLinearIters / primalPCLag; pseudo-time control
 * amd.primalTauMode "ser" | "ramp" (+ primalTauGrowth / GrowthMax / Max / Min, primalAcceptFactor, primalDampedSteps) and
 * amd.primalPseudoTimeFields "all" | "momentum" (cold starts: the term on the transport rows only)).  Returns 0 converged
 * (|R| <= max(relTol |R0|, absTol)) / 1 not converged; info4 = {Newton steps, GMRES iterations, |R0|, |R|}. */
int das_solve_primal(das_solver_t* s, int maxSteps, double relTol, double absTol, double* info4, double* hist, int histCap);
/* das_simple_iteration: nSweeps iterations of the reference's OWN primal loop - SIMPLE (DASimpleFoam::solvePrimal, DASimpleFoam.C:123-185:
 * UEqnSimple.H, pEqnSimple.H with nNonOrthogonalCorrectors 1, DASpalartAllmaras::correct) - on the device, from the current states:
 * relaxed momentum predictor, rAU / HbyA / constrainHbyA, pressure equation, phi = phiHbyA - flux, explicit p relaxation (alphaP =
 * fvSolution relaxationFactors.fields.p), U correction, SA transport + bound.  Inner solves: Jacobi-preconditioned BiCGStab (U, nuTilda) /
 * CG (p) to the relative tolerance linTol, at most maxLinIters iterations.  DASimpleFoam + SA without T field, MRF and cyclic pairs, one
 * rank.  info3 (optional) = inner iterations of the last sweep (U, p, nuTilda).  The Newton-Krylov das_solve_primal reaches the same
 * fixed point in far fewer steps; the sweeps reproduce the reference's iteration (and the oracle's, oracle/primal.py) sweep by sweep. */
int das_simple_iteration(das_solver_t* s, int nSweeps, double alphaP, double linTol, int maxLinIters, double* info3);
int das_run_coloring(das_solver_t* s);
/* das_set_coloring <- DAJacCon::readJacConColoring (DAJacCon.C:1980-2019): colours read back from a dRdWColoring_n.bin
 *                      cache; validated against the freshly built connectivity ("Conflicting Colors Found!" otherwise). */
int das_set_coloring(das_solver_t* s, const int* colors);
/* host-only profiling aid (no GPU needed): factorises one block-local CSR (sorted columns) exactly like a preconditioner
 * block - symbolic ILU(lfill), numeric, level schedules, entry streams - and returns the four phase times [s] */
int das_debug_factor_block(int nl, const long long* rowptr, const int* col, const double* val, int lfill, double* tim4, long long* nnzLU,
                           int* nLevelsL, int* nLevelsU);
int das_get_n_colors(das_solver_t* s, int isPC);
long long das_get_con_nnz(das_solver_t* s, int isPC);
int das_get_con(das_solver_t* s, int isPC, long long* rowptr /*n+1*/, int* colidx /*nnz*/);
int das_get_colors(das_solver_t* s, int isPC, int* colors /*n*/);

/* ---- Jacobians ---------------------------------------------------------------------------
 * das_calc_drdwt <- calcdRdWT(isPC, Mat dRdWT)  pyDASolvers.pyx:237 (DASolver.C:948-1089, DAPartDeriv.C:350-473).
 *   mode: 0 = coloured one-sided finite differences (reference behaviour, step adjPartDerivFDStep.State),
 *         1 = coloured dual-number (forward-mode AD) perturbations - exact derivatives on the same colouring.
 *   Result: transposed CSR on the device, entry (j,i) = s_j dR_i/dW_j. */
int das_calc_drdwt(das_solver_t* s, int isPC, int mode, das_mat_t** out);
long long das_mat_rows(das_mat_t* m);
long long das_mat_nnz(das_mat_t* m);
int das_mat_export(das_mat_t* m, long long* rowptr, int* colidx, double* vals);
/* y = A x on host buffers (testing) */
int das_mat_mult(das_mat_t* m, const double* x, double* y);
/* device CSR handle from host arrays (reference: PETSc.Mat().load of dRdWTPC.bin under adjEqnOption.readPCMat,
 * dafoam/mphys/mphys_dafoam.py:469-471) */
int das_mat_create_from_csr(long long n, const long long* rowptr, const int* colidx, const double* vals, das_mat_t** out);
void das_mat_destroy(das_mat_t* m);

/* das_initialize_drdwt_matrix_free <- initializedRdWTMatrixFree()  pyDASolvers.pyx:253 (DASolver.C:1321-1351):
 *   the reference creates a PETSc MatShell whose MULT replays a CoDiPack reverse tape; here the operator is
 *   assembled once per call with dual numbers (mode 1, isPC=0 schemes) and applied as a device SpMV.
 * das_destroy_drdwt_matrix_free    <- destroydRdWTMatrixFree()      pyDASolvers.pyx:256 */
int das_initialize_drdwt_matrix_free(das_solver_t* s);
int das_destroy_drdwt_matrix_free(das_solver_t* s);
/* nnz of the assembled matrix-free operator (after the jacLowerBounds filter), -1 if not initialised */
long long das_op_nnz(das_solver_t* s);
/* bytes of matrix data one dRdW^T.psi product streams in the operator's storage format: vector-state rows packed as group
 * rows (one int32 column list + three fp64 value planes per 16 entries, csrc/das_opmat.hpp), scalar rows as CSR (12 B/entry) */
long long das_op_format_bytes(das_solver_t* s);
/* the operator of initializedRdWTMatrixFree as CSR arrays on the host (rowptr[n+1], colidx[das_op_nnz], vals[das_op_nnz]) */
int das_op_export(das_solver_t* s, long long* rowptr, int* colidx, double* vals);

/* ---- unsteady adjoint terms (DAScalarTransportFoam, BASELINE configs[0]) ---------------------------------------------
 * das_calc_drdwold_t_psi <- calcdRdWOldTPsiAD(oldTimeLevel, psi, dRdWOldTPsi)  pyDASolvers.pyx:240 (DASolver.C:1910-1969):
 *     D_s (dR/dW_old)^T psi for oldTimeLevel 1 (W0) or 2 (W00).  Euler ddt: dR/dT0 = -1/deltaT (per-volume residual),
 *     level 2 and steady solvers give zero.
 * das_set_old_time_fields: frozen flux and old-time temperature of the current time step (the reference re-reads them
 *     from disk, readStateVars DASolver.C:3193). */
int das_calc_drdwold_t_psi(das_solver_t* s, int oldTimeLevel, const double* psi, double* out);
int das_set_old_time_fields(das_solver_t* s, const double* phi_frozen /* n_faces or NULL */, const double* T_old /* n_cells or NULL */);

/* ---- objective functions (adjoint right-hand side producers) -----------------------------------------------
 * das_define_force_function <- the "function" option entry {type: force, patches, directionMode: fixedDirection,
 *     direction, scale} consumed by DAFunctionForce (reference src/adjoint/DAFunction/DAFunctionForce.C:20-77)
 * das_calc_function         <- calcFunction(functionName)  pyDASolvers.pyx (DAFunctionForce::calcFunction :79-158) */
/* das_calc_jac_vec_product <- the forward-mode (ADF build) directional derivative of the residuals, DASolver.C:1364-1441
 *                             run forward: product_i = sum_j dR_i/dW_j s_j v_j at the current states.  One dual-number
 *                             residual pass; no colouring and no matrix are involved. */
int das_calc_jac_vec_product(das_solver_t* s, const double* v, double* product);
/* Boundary-value design inputs.
 * das_set_patch_value   <- DAInputPatchVelocity::run / DAInputPatchVar::run (src/adjoint/DAInput/DAInputPatchVelocity.C:33-135):
 *                          assigns the (ref)value of fixedValue / inletOutlet patches; any other patch type is an error.
 *                          field: "U" (value[3]) | "p" | "nuTilda" | "T" (value[1]).
 * das_calc_dbc_product  <- calcJacTVecProduct(inputType = patchVelocity | patchVar) pyDASolvers.pyx:208-235:
 *                          product[0] = seeds^T (dOutput/d(patch value) . tangent), outputType "residual" (seeds: n,
 *                          host) or "function" (seeds: 1).  One forward-mode (dual number) pass. */
int das_set_patch_value(das_solver_t* s, const int* patch_ids, int npatch, const char* field, const double* value);
int das_get_patch_value(das_solver_t* s, int patch_id, const char* field, double* value);
/* `field` inputs (reference DAInputField.C: a design variable that IS a volScalarField; inputInfo type "field"):
 * fieldName "betaFINuTilda" (values[nCells]).  das_calc_dfield_product: product[c] = sum_i seeds[i] dR_i/dfield_c (outputType
 * "residual"; ONE forward-mode residual pass with a unit tangent on every cell - a residual row depends on the field value of
 * its own cell only) or 0 for the patch-integral functions (outputType "function"), i.e. calcJacTVecProduct(field -> ...)
 * of DASolver.C:1690-1839. */
int das_set_field(das_solver_t* s, const char* fieldName, const double* values);
int das_get_field(das_solver_t* s, const char* fieldName, double* values);
int das_calc_dfield_product(das_solver_t* s, const char* fieldName, const char* outputName, const char* outputType, const double* seeds,
                            double* product);
int das_calc_dbc_product(das_solver_t* s, const int* patch_ids, int npatch, const char* field, const double* tangent, const char* outputName,
                         const char* outputType, const double* seeds, double* product);
int das_define_force_function(das_solver_t* s, const char* name, const int* patch_ids, int npatch, const double* direction, double scale);
/* das_define_face_function <- the other patch-integral entries of the "function" option dict (pyDAFoam.py:100-200):
 *   type "force"                 vecA = direction (unit)                                   DAFunctionForce.C:79-158
 *        "moment"                vecA = axis (unit), vecB = center                         DAFunctionMoment.C:73-120
 *        "massFlowRate"          sum rho_b (U_b . S_f) * scale                             DAFunctionMassFlowRate.C:52-80
 *        "totalPressure"         area average of p_b + 0.5 rho_b |U_b|^2, * scale          DAFunctionTotalPressure.C:60-90
 *        "totalTemperatureRatio" TT_out / TT_in (area averages); patch_group[k] = 0 inlet, 1 outlet; gammaFn = the
 *                                thermophysicalProperties gamma, R = Cp - Cp/gamma         DAFunctionTotalTemperatureRatio.C:60-130
 * patch_group may be NULL for the non-ratio types.  Values and derivatives go through das_calc_function /
 * das_calc_jac_t_vec_product / das_calc_dbc_product like the force. */
int das_define_face_function(das_solver_t* s, const char* name, const char* type, const int* patch_ids, const int* patch_group, int npatch,
                             const double* vecA, const double* vecB, double scale, double gammaFn);
int das_calc_function(das_solver_t* s, const char* name, double* value);

/* das_calc_jac_t_vec_product <- calcJacTVecProduct(inputName,inputType,inputs,outputName,outputType,seeds,product)
 *   pyDASolvers.pyx:208-235 (DASolver.C:1690-1839).  Supported pair on this path: inputType "stateVar",
 *   outputType "residual": product = D_s (dR/dW)^T seeds  (normalizeJacTVecProduct, DASolver.C:1443-1553);
 *   outputType "function" (outputName = function name, one seed): product = seed * D_s dF/dW  (mphys_dafoam.py:746-801). */
int das_get_input_size(das_solver_t* s, const char* inputName, const char* inputType);
int das_get_output_size(das_solver_t* s, const char* outputName, const char* outputType);
int das_calc_jac_t_vec_product(das_solver_t* s, const char* inputName, const char* inputType, const double* inputs,
                               const char* outputName, const char* outputType, const double* seeds, double* product);
/* same product on device buffers (no copies; used by bench.py); d_x,d_y: n doubles in HBM */
int das_drdwt_mult_device(das_solver_t* s, const double* d_x, double* d_y);

/* ---- Krylov solve ------------------------------------------------------------------------
 * das_create_ml_rksp_matrix_free <- createMLRKSPMatrixFree(Mat jacPCMat, KSP ksp) pyDASolvers.pyx:259
 *     (DASolver.C:1102-1119 -> DALinearEqn::createMLRKSP, DALinearEqn.C:28-339): right-preconditioned
 *     restarted GMRES, PC = block (additive-Schwarz-like) ILU(pcFillLevel) of jacPCMat.
 * das_solve_linear_eqn <- solveLinearEqn(KSP, Vec rhs, Vec sol) pyDASolvers.pyx:265 (DALinearEqn.C:341-437);
 *     returns 0 converged / 1 failed by the reference's gmresTolDiff rule (:422-434), <0 on error. */
int das_create_ml_rksp_matrix_free(das_solver_t* s, das_mat_t* pc, das_ksp_t** ksp);
int das_solve_linear_eqn(das_solver_t* s, das_ksp_t* ksp, const double* rhs, double* sol);
/* das_solve_linear_eqn_block: nrhs (1..8) adjoint systems with the same operator through ONE block GMRES (the reference
 *     loops solveLinearEqn over the objective functions, mphys_dafoam.py:478-481): dRdW^T is streamed once per iteration
 *     for all systems, the block orthogonalisation runs as tall-skinny fp64 MFMA GEMMs.  rhs / sol: column-major n x nrhs;
 *     res0 / res (optional, nrhs each): initial / final residual norms.  Returns 1 if any system fails the reference rule. */
int das_solve_linear_eqn_block(das_solver_t* s, das_ksp_t* ksp, int nrhs, const double* rhs, double* sol, double* res0, double* res);
/* preconditioner introspection (tests): y = M^{-1} x on host buffers; block layout of the additive-Schwarz PC:
 * perm[n] = global state index at each permuted position, block_off[nBlocks+1] offsets into perm */
int das_ksp_apply_pc(das_solver_t* s, das_ksp_t* ksp, const double* x, double* y);
int das_ksp_get_n_blocks(das_ksp_t* ksp);
/* nnz(L+U) summed over all (overlapping) blocks and total extended unknowns of the preconditioner */
long long das_ksp_get_factor_nnz(das_ksp_t* ksp);
long long das_ksp_get_n_ext(das_ksp_t* ksp);
int das_ksp_get_blocks(das_ksp_t* ksp, int* perm, long long* block_off);
/* node structure of the default preconditioner (amd.pcType "bilu": one node-block ILU(0) of jacPCMat per GPU, the
 * reference's ASM+ILU stack DALinearEqn.C:199-299 with one sub-domain per rank): nodes of 8 unknown slots in processing
 * (level) order, nodeUnk[8 nNodes] = state index or -1, block CSR bptr[nNodes+1]/bcol[nBlocks] over node positions,
 * lvlPtr[nLevels+1], natural[nNodes] = index of the node in the natural (cell-by-cell) order.  das_pc_structure_* build it on the host without a GPU (tests), das_ksp_get_pc_structure* return
 * the one a KSP was factorised on. */
int das_pc_structure_build(das_solver_t* s, int* nNodes, long long* nBlocks, int* nLevels, int* reach);
int das_pc_structure_get(das_solver_t* s, int* nodeUnk, long long* bptr, int* bcol, int* lvlPtr, int* natural);
int das_ksp_get_pc_structure_sizes(das_ksp_t* ksp, int* nNodes, long long* nBlocks, int* nLevels);
int das_ksp_get_pc_structure(das_ksp_t* ksp, int* nodeUnk, long long* bptr, int* bcol, int* lvlPtr, int* natural);
/* nodeOut[nNodes * 8]: the unknown every slot WRITES: nodeUnk, or -1 for the overlap copies of a multi-block factorisation (amd.pcSubdomains) */
int das_ksp_get_pc_node_out(das_ksp_t* ksp, int* nodeOut);
int das_ksp_get_info(das_ksp_t* ksp, int* iters, double* res0, double* res, double* seconds);
int das_ksp_get_history(das_ksp_t* ksp, double* hist, int cap);
/* Krylov basis of the last solve (amd.krylovBasisPrecision): *fp32 = bit 0: plain fp32 storage, bit 1: split storage (hi + lo floats, the
 * inner products read the hi array only); device bytes mapped so far; bytes per stored basis vector */
int das_ksp_get_basis_info(das_ksp_t* ksp, int* fp32, double* mappedBytes, double* bytesPerVector);
/* columns of every closed Arnoldi cycle of the last solveLinearEqn (reference: KSPGMRESSetRestart, DALinearEqn.C:155 - every cycle
 * but the last holds exactly gmresRestart columns); writes min(cap, count) entries, returns the count */
int das_ksp_get_cycle_lengths(das_ksp_t* ksp, int* lens, int cap);
/* number of Gram-Schmidt refinement passes of the last solve (KSP_GMRES_CGS_REFINE_IFNEEDED, DALinearEqn.C:160); with
 * amd.gmresOrthogonalization "dcgs2": the number of explicit projections (exhausted Krylov space / lost orthogonality) */
int das_ksp_get_n_refine(das_ksp_t* ksp);
/* amd.gmresDeflation = k > 0 (opt-in; round 4, not yet measured on the device): GMRES with deflated restarting (GMRES-DR; PETSc's
 * counterpart is KSPDGMRES, not the reference's default KSPGMRES of DALinearEqn.C:28-339): gmresRestart basis vectors, k harmonic
 * Ritz vectors carried across restarts.  The dense m x m eigenproblem of a restart goes through a process-wide callback
 * fn(m, A row-major, wr, wi, vr, vi) -> 0 (eigenvector e in vr / vi [e m, e m + m)); das_debug_gmres_dr_restart exposes the host
 * algebra of one restart to the CPU tier (returns the number of kept vectors, which never splits a complex pair). */
int das_set_dense_eig_callback(void* fn);
int das_debug_gmres_dr_restart(int m, int kwant, const double* Hbar, const double* rvec, double* P1, double* Hnew, double* cnew);
/* the deflated-restart iteration itself on HOST vectors (operator A and preconditioner M as callbacks fn(x, y, user)): the very loop the
 * device solver runs, for the CPU tier; info4 = {iterations, deflated restarts, plain restarts, breakdowns}, res2 = {|r0|, |r|} */
int das_debug_gmres_dr_host(long long n, void* A, void* M, void* user, const double* b, double* x, int m, int kdef, double rtol, double atol,
                            long long maxIts, double* hist, int histCap, double* info4, double* res2);
/* how the last solve ended: reason 0 = tolerance met (KSP_CONVERGED_RTOL/ATOL), 1 = gmresMaxIters reached (KSP_DIVERGED_ITS),
 * 2 = stopped on stagnation after a Krylov breakdown / at the attainable accuracy (PETSc: KSP_CONVERGED_HAPPY_BREAKDOWN /
 * KSP_DIVERGED_BREAKDOWN; the reference's failure flag is still the tolerance rule of DALinearEqn.C:422-434);
 * nBreakdown = happy breakdowns detected, nSweepGrid / sweepPerXcd = launch shape of the preconditioner sweeps */
int das_ksp_get_status(das_ksp_t* ksp, int* reason, int* nBreakdown, int* nSweepGrid, int* sweepPerXcd);
/* Stability estimate of the incomplete factorisation, max |(LU)^-1 P e - e| on two test vectors (amd.pcStabilityLimit; -1: not computed),
 * and the elimination order of the cells the check settled on (0 mesh numbering, 1 reverse Cuthill-McKee = jacMatReOrdering "rcm",
 * 2 Cuthill-McKee, 3 mesh numbering backwards, 4 / 5 along / against the mean flow; -1: no check).  A factorisation above the limit is
 * rebuilt with the next order, rank by rank.  The reference's PETSc ILU has no such check; its MatFactorInfo shift (DALinearEqn.C:270-272)
 * only replaces zero pivots. */
int das_ksp_get_pc_stability(das_ksp_t* ksp, double* estimate, int* orderUsed);
/* amd.pcSubdomains K (default -1: 4 from 1 M cells on): restricted additive Schwarz INSIDE the rank - K node-block ILUs on recursive-coordinate-
 * bisection blocks of the cells plus adjEqnOption.asmOverlap rings, each with its own elimination order, merged into one level structure so
 * that one pair of sweeps runs them together (the reference reaches several sub-domains per device only through more MPI ranks,
 * DALinearEqn.C:212-216).  Returns K (1: one factorisation, -1: null handle); orders[K] / estimates[K] are optional. */
int das_ksp_get_pc_subdomains(das_ksp_t* ksp, int* orders, double* estimates);
/* two-level preconditioner (amd.pcCoarseAggregates / pcCoarseField / pcCoarseMode): number of aggregates of the
 * piecewise-constant pressure coarse space (0 = none) and, optionally, the aggregate of every cell (-1 = not owned) */
int das_ksp_get_coarse(das_ksp_t* ksp, int* aggOfCell);
/* 1 if the deflated coarse correction of this KSP takes A (Z u) from the precomputed sparse A Z (amd.pcCoarseSparseAZ, round 5), 0 if it runs
 * a full operator product per apply - test aid */
int das_ksp_coarse_sparse_az_active(das_ksp_t* ksp);
/* multi-GPU: ONE coarse space over all ranks instead of one per rank (the reference's ASM level couples the sub-domains through
 * its overlap, DALinearEqn.C:199-216; a per-rank coarse space lets the iteration count grow with the number of GPUs).
 * Collective over the installed communication.  naggGlobal <= 2048: size of the global coarse operator; aggOffset: position of
 * this rank's aggregates; aggRowGlobal[nLocalCells]: global aggregate of every local cell, owned (= aggOffset + local id) and
 * ghost (the owner rank's numbering), -1 = none.  Returns 0, or 1 if the coarse operator is singular (no correction). */
int das_ksp_set_global_coarse(das_solver_t* s, das_ksp_t* ksp, int naggGlobal, int aggOffset, const int* aggRowGlobal);
/* run exactly `iters` GMRES iterations on device-resident rhs/sol (bench.py "step"); no convergence exit */
int das_ksp_run_fixed_device(das_solver_t* s, das_ksp_t* ksp, const double* d_rhs, double* d_sol, int iters);
/* the same solve advanced in pieces on device-resident rhs/sol (bench.py times a window of iterations deep inside an
 * Arnoldi cycle): begin -> advance(n) ... -> end.  advance returns 1 once the solve is over (never in `fixed` mode),
 * end closes the open cycle (x += M^-1 V y, true residual) and returns the solveLinearEqn fail code. */
int das_ksp_begin_device(das_solver_t* s, das_ksp_t* ksp, const double* d_rhs, double* d_sol, int fixed);
int das_ksp_advance(das_solver_t* s, das_ksp_t* ksp, int iters);
int das_ksp_end(das_solver_t* s, das_ksp_t* ksp);
void das_ksp_destroy(das_ksp_t* k);

/* ---- multi-GPU sharding (one process per GPU; reference: MPI domain decomposition, one OpenFOAM sub-domain per rank) ----
 * The solver instance holds the rank's EXTENDED sub-mesh (owned cells + ghost layers).  owned[j] != 0 marks the
 * states/residuals this rank owns (DAIndex ordering of the extended mesh).  With a mask set:
 *   - assembled matrices keep only columns (residuals) owned by this rank (rows = all extended states),
 *   - the preconditioner is built on owned cells only,
 *   - after every operator product the halo callback is invoked on the device vector so that ghost-row
 *     contributions can be sent to their owner ranks and added there (then zeroed locally),
 *   - every fused dot-product result (device buffer of n doubles) goes through the all-reduce callback.
 * Callbacks run on the stream given to das_set_stream (pass the communication library's current stream). */
typedef void (*das_halo_cb)(double* d_vec, void* user);
typedef void (*das_allreduce_cb)(double* d_buf, int n, void* user);
int das_set_owned_mask(das_solver_t* s, const unsigned char* owned /* n states */);
int das_set_comm(das_solver_t* s, das_halo_cb halo, das_allreduce_cb allreduce, void* user);
/* Native transport set-up, step 1 (local, not collective): dlopen the RCCL PyTorch already loaded and bind its symbols.
 * All ranks exchange the return code (torch.distributed MIN all-reduce) BEFORE any of them calls the collective
 * das_comm_init_rccl: a rank without RCCL must not leave its peers blocked in ncclCommInitRank. */
int das_comm_load_rccl(void);
/* Native transport (no host code in the iteration loop): RCCL point-to-point halo reduction overlapped with the owned-row
 * product + in-stream ncclAllReduce of the Gram-Schmidt dots.  Reference: PETSc VecScatter of the MPIAIJ off-diagonal
 * block in MatMult and MPI_Allreduce in VecMDot (KSPGMRES, DALinearEqn.C:341-437).
 *   das_comm_unique_id   rank 0: 128-byte RCCL id, distributed by the host side (bootstrap only)
 *   das_comm_init_rccl   every rank: communicator on the solver's device
 *   das_comm_set_halo    the halo plan: peers[npeers]; sendIdx[sendOff[i]..sendOff[i+1]) = extended rows held for peer i
 *                        (evaluated first, sent); recvIdx[...] = my owned rows peer i holds as ghosts, in its send order;
 *                        ghostIdx[nGhost] = all local ghost rows (zeroed after the reduction)
 *   das_set_exchange_cb  host-staged transport of the same plan (gloo tests on single-GPU boxes) */
typedef void (*das_exchange_cb)(double* d_send, double* d_recv, void* user);
int das_comm_unique_id(char* out128);
int das_comm_init_rccl(das_solver_t* s, int rank, int world, const char* id128);
int das_comm_set_halo(das_solver_t* s, int npeers, const int* peers, const long long* sendOff, const int* sendIdx, const long long* recvOff,
                      const int* recvIdx, long long nGhost, const int* ghostIdx);
int das_set_exchange_cb(das_solver_t* s, das_exchange_cb cb, void* user);
/* Additive-Schwarz overlap of the preconditioner across ranks - adjEqnOption.asmOverlap, reference DALinearEqn.C:212-216
 * (PCASMSetOverlap; PETSc's default restricted variant): pcMask[state] != 0 for the unknowns of this rank's sub-domain solve (owned +
 * overlap rings of ghost cells); per peer of das_comm_set_halo (same order) the owned states it needs (ovSend) and the overlap ghost
 * states it owns (ovRecv, in its send order).  Call before calcdRdWT(1) / createMLRKSPMatrixFree.  pcMask NULL: no overlap.
 * das_set_gather_cb: host-staged transport of the gather (gloo tests on single-GPU boxes). */
int das_set_pc_overlap(das_solver_t* s, const unsigned char* pcMask /* n states */, int npeers, const long long* ovSendOff, const int* ovSendIdx,
                       const long long* ovRecvOff, const int* ovRecvIdx);
int das_set_gather_cb(das_solver_t* s, das_exchange_cb cb, void* user);
int das_comm_is_native(das_solver_t* s);
/* drop the native communicator again (all ranks fall back to the callback transport together) */
int das_comm_reset(das_solver_t* s);
int das_set_stream(das_solver_t* s, void* hip_stream);

/* ---- timing (getElapsedClockTime/getElapsedCpuTime pyDASolvers.pyx:332-336) and kernel timers */
double das_get_elapsed_clock_time(das_solver_t* s);
double das_get_elapsed_cpu_time(das_solver_t* s);
/* average duration [ms] of the named kernel family measured with HIP events on the launch stream since
 * the last reset: "spmv", "pc", "residual" ; returns <0 if never launched */
double das_timer_avg_ms(das_solver_t* s, const char* name);
/* tuning hook (tools/orth_bench.py): average ms of the two kernels of the delayed re-orthogonalisation (inner products with
 * `rows` rows per thread; update with `unroll` basis vectors in flight and `rpt` rows per thread) on synthetic vectors of
 * length n against K basis vectors; -1 for a variant that is not compiled in */
int das_debug_orth_bench(long long n, int K, int reps, int rows, int unroll, int rpt, double* ms_dots, double* ms_update);
long long das_timer_count(das_solver_t* s, const char* name);
void das_timer_reset(das_solver_t* s);
void das_timer_enable(das_solver_t* s, int on);

#ifdef __cplusplus
}
#endif
#endif
