"""CPU tier: everything of the hot path that CAN be pinned against the reference's own sources in this container is
pinned here, by parsing /root/reference at test time (VERDICT round 2, row c).

The reference cannot be built or imported here (OpenFOAM-AD / PETSc / CoDiPack / mpi4py are absent) and
its meshes are not vendored, so no NUMBER of its tests can be reproduced.  What its sources do hold in machine-checkable
form - and what this file compares with BOTH the oracle (oracle/) and the product (C library / Python mirror) - is:

  * the stencil ("connectivity level") tables   src/adjoint/DAStateInfo/DAStateInfo{SimpleFoam,RhoSimpleFoam,TurboFoam,
                                                ScalarTransportFoam}.C, DASpalartAllmaras.C:364-383 (addModelResidualCon)
    -> oracle.jacobian.STENCIL, and - through an independent ring walk written in this file from the semantics of
       DAJacCon::addStateConnections (DAJacCon.C:304-667: level k = cells reached by k face-neighbour steps; phi at level k
       = all faces of those cells) - the FULL connectivity pattern das_get_con returns, row by row
       (incl. the boundary-face rule of DAJacCon::setupdRdWCon, DAJacCon.C:2432-2449);
  * the PC level reduction defaults               dafoam/pyDAFoam.py:568-582 (maxResConLv4JacPCMat);
  * the Spalart-Allmaras constants                DASpalartAllmaras.C:47-80 -> oracle.residual.SA and csrc/das_kernels.hpp;
  * the wall-function constants / Newton settings nutUSpaldingWallFunction...DF.C;
  * every DAOPTION default                        dafoam/pyDAFoam.py:59-661 (parsed with `ast`) -> dafoam_amd.pyDAFoam.DAOPTION;
  * the Krylov configuration                      DALinearEqn.C:28-437 (GMRES, CGS refine-if-needed, right PC, unpreconditioned
                                                norm, ASM/ILU/shift, failure rule) -> the claims of csrc/das_device.hip;
  * state ordering                                DAIndex.C (volVector, volScalar, model, surfaceScalar) -> state_layout;
  * the file names / viewer calls of the PETSc binary dumps (DAUtility.C:282-441, DAJacCon.C:1886-2014) -> petsc_io.py.

The tests skip when /root/reference is absent (the GPU box); they are in the CPU tier, which runs in this container.
"""
import ast
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from common import options
from dafoam_amd.meshgen import channel_case, rho_channel_case, scalar_transport_case, simple_T_channel_case, turbo_channel_case
from oracle import jacobian as J
from oracle import residual as OR
from oracle.foam_mesh import Geometry

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "adjoint")), reason="/root/reference is not present on this machine")


def ref_text(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def parse_level_tables(src, setter):
    """All `<setter>.set("NAME", { {..}, {..}, ... });` blocks of a reference source -> [(NAME, [[states lv0], [lv1], ...])].
    Identifiers (pName) are kept as bare words."""
    src = strip_comments(src)
    out = []
    for m in re.finditer(re.escape(setter) + r"\.set\(\s*\"(\w+)\"\s*,\s*\{(.*?)\}\s*\)\s*;", src, flags=re.S):
        levels = [[t.strip().strip('"') for t in grp.split(",") if t.strip()] for grp in re.findall(r"\{([^{}]*)\}", m.group(2))]
        out.append((m.group(1), levels))
    return out


def reference_stencil(solver, has_T=False):
    """{residual: levels} of a solver as the reference builds it: DAStateInfo<solver>.C + the SA model table, with the model
    state `nut` renamed to `nuTilda` (DASpalartAllmaras::correctModelStates / correctStateResidualModelCon)."""
    sa = ref_text("src/adjoint/DAModel/DATurbulenceModel/DASpalartAllmaras.C")
    assert re.search(r'stateName == "nut"\)\s*\{\s*modelStates\[idxI\] = "nuTilda";', sa)  # the rename this function applies
    tabs = parse_level_tables(ref_text(f"src/adjoint/DAStateInfo/DAStateInfo{solver[2:]}.C"), "stateResConInfo_")
    res = {}
    for name, lv in tabs:
        if name == "TRes" and solver == "DASimpleFoam" and not has_T:
            continue
        res[name] = [["nuTilda" if s == "nut" else s for s in level] for level in lv]
    if solver != "DAScalarTransportFoam":
        sa_tabs = parse_level_tables(sa, "allCon")
        assert [n for n, _ in sa_tabs] == ["nuTildaRes", "nuTildaRes"]  # incompressible first, compressible second (:364-383)
        assert re.search(r'turbModelType_ == "incompressible"\)\s*\{\s*allCon\.set', strip_comments(sa))
        lv = sa_tabs[0 if solver == "DASimpleFoam" else 1][1]
        res["nuTildaRes"] = [["p" if s == "pName" else s for s in level] for level in lv]
    return res


def reference_daoption():
    tree = ast.parse(ref_text("dafoam/pyDAFoam.py"))
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DAOPTION"][0]
    init = [f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == "__init__"][0]
    return {st.targets[0].attr: ast.literal_eval(st.value) for st in init.body
            if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Attribute)}


# ------------------------------------------------------------------------------------------------- stencil tables
@pytest.mark.parametrize("solver", ["DASimpleFoam", "DARhoSimpleFoam", "DATurboFoam", "DAScalarTransportFoam"])
def test_oracle_stencil_tables_equal_the_reference_sources(solver):
    ref = reference_stencil(solver)
    ours = {k: v for k, v in J.STENCIL[solver].items() if k != "states"}
    assert set(ours) == set(ref)
    for name in ref:
        assert [sorted(l) for l in ours[name]] == [sorted(l) for l in ref[name]], name


def test_simplefoam_with_T_table_is_the_reference_plus_one_documented_entry():
    ref = reference_stencil("DASimpleFoam", has_T=True)
    assert ref["TRes"] == [["T", "nuTilda", "phi"], ["T", "nuTilda"], ["T"]]
    ours = {k: v for k, v in J.STENCIL["DASimpleFoam+T"].items() if k != "states"}
    for name in ref:
        for lv, (a, b) in enumerate(zip(ours[name], ref[name])):
            extra = set(a) - set(b)
            assert set(b) <= set(a) and extra == ({"U"} if (name, lv) == ("TRes", 0) else set()), (name, lv)


def test_pc_level_reduction_and_state_order_equal_the_reference():
    opt = reference_daoption()
    mine = {k: v for k, v in opt["maxResConLv4JacPCMat"].items() if k in J.MAX_RES_CON_LV_PC}
    assert mine == J.MAX_RES_CON_LV_PC and opt["maxResConLv4JacPCMat"]["pRes"] == 2 and opt["maxResConLv4JacPCMat"]["phiRes"] == 1
    # DAIndex.C: adjStateNames are registered volVector, volScalar, model, surfaceScalar - the "state" ordering blocks
    idx = strip_comments(ref_text("src/adjoint/DAIndex/DAIndex.C"))
    order = [m.start() for m in (re.search(r'stateInfo_\["volVectorStates"\]', idx), re.search(r'stateInfo_\["volScalarStates"\]', idx),
                                 re.search(r'stateInfo_\["modelStates"\]', idx), re.search(r'stateInfo_\["surfaceScalarStates"\]', idx))]
    assert order == sorted(order)
    si = strip_comments(ref_text("src/adjoint/DAStateInfo/DAStateInfoSimpleFoam.C"))
    assert re.search(r'\["volVectorStates"\]\.append\("U"\)', si) and re.search(r'\["volScalarStates"\]\.append\("p"\)', si)
    assert re.search(r'\["modelStates"\]\.append\("nut"\)', si) and re.search(r'\["surfaceScalarStates"\]\.append\("phi"\)', si)
    assert [n for n, _ in J.STENCIL["DASimpleFoam"]["states"]] == ["U", "p", "nuTilda", "phi"]
    rho = strip_comments(ref_text("src/adjoint/DAStateInfo/DAStateInfoRhoSimpleFoam.C"))
    assert re.search(r'\["volScalarStates"\]\.append\("p"\).*\["volScalarStates"\]\.append\("T"\)', rho, flags=re.S)
    assert [n for n, _ in J.STENCIL["DARhoSimpleFoam"]["states"]] == ["U", "p", "T", "nuTilda", "phi"]


def _walk_pattern(case, g, table, states, pc_levels=None):
    """dRdWCon from a level table by the semantics of DAJacCon::addStateConnections / setupdRdWCon, written independently of
    oracle.jacobian.connectivity and csrc/das_jaccon.cpp: level k = cells reached by exactly k face-neighbour steps; a vector
    state contributes its 3 components, a surface state all faces of the level's cells; a face residual collects from BOTH
    adjacent cells (DAJacCon.C:2459-2478).  Cyclic-free meshes only."""
    m = case.mesh
    N, F, nIF = g.nC, g.nF, g.nIF
    nbrs = [[] for _ in range(N)]
    for f in range(nIF):
        o, n_ = int(m.owner[f]), int(m.neighbour[f])
        nbrs[o].append(n_)
        nbrs[n_].append(o)
    cfaces = [[] for _ in range(N)]
    for f in range(F):
        cfaces[int(m.owner[f])].append(f)
        if f < nIF:
            cfaces[int(m.neighbour[f])].append(f)
    off, size, n = J.state_layout(J.stencil_name(case), N, F)
    kind = dict(states)

    surf = [nm for nm, kd in states if kd == "face"]

    def cols_of_cell(c, res, boundary_face=False):
        levels = table[res] if pc_levels is None else table[res][: pc_levels[res] + 1]
        cols, ring = set(), {c}
        for k, names in enumerate(levels):
            if k > 0:
                ring = {y for x in ring for y in nbrs[x]}
            for nm in names:
                if kind[nm] == "vec":
                    cols.update(off[nm] + 3 * x + d for x in ring for d in range(3))
                elif kind[nm] == "scl":
                    cols.update(off[nm] + x for x in ring)
            # the faces of the level's cells: decided by the level's own list, except for the row of a BOUNDARY face at
            # level > 0, where the reference looks one level back ("levelCheck = idxJ - 1", DAJacCon.C:2432-2449: a boundary
            # face has only its owner side, so one more level of faces is added)
            check = levels[k - 1] if (boundary_face and k > 0) else names
            for nm in surf:
                if nm in check:
                    cols.update(off[nm] + f for x in ring for f in cfaces[x])
        return cols

    rows = {}
    for nm, kd in states:
        res = nm + "Res"
        if kd == "face":
            for f in range(F):
                cs = cols_of_cell(int(m.owner[f]), res, boundary_face=f >= nIF)
                if f < nIF:
                    cs |= cols_of_cell(int(m.neighbour[f]), res)
                rows[off[nm] + f] = cs
        else:
            per = 3 if kd == "vec" else 1
            for c in range(N):
                cs = cols_of_cell(c, res)
                for d in range(per):
                    rows[off[nm] + per * c + d] = cs
    ind = np.concatenate([np.fromiter(sorted(rows[r]), dtype=np.int64) for r in range(n)])
    ptr = np.concatenate([[0], np.cumsum([len(rows[r]) for r in range(n)])])
    return sp.csr_matrix((np.ones(ind.size, np.int8), ind, ptr), shape=(n, n))


CASES = {
    "DASimpleFoam": lambda: channel_case(6, 5, 5),
    "DARhoSimpleFoam": lambda: rho_channel_case(5, 5, 4),
    "DATurboFoam": lambda: turbo_channel_case(5, 4, 4),
    "DAScalarTransportFoam": lambda: scalar_transport_case(6, 5, 4),
}


@pytest.mark.parametrize("solver", list(CASES))
def test_library_and_oracle_connectivity_equal_the_reference_tables_walked_independently(solver):
    """das_get_con (the pattern every Jacobian of the product is assembled on) and oracle.jacobian.connectivity, both for
    dRdWT and for the PC-reduced dRdWTPC, against the pattern generated in THIS file from the tables parsed out of the
    reference sources.  Structural row lengths of SURVEY.md section 8a on an interior hex cell come out of the same walk."""
    from dafoam_amd.pyDASolvers import pyDASolvers

    case = CASES[solver]()
    if any(getattr(p, "type", "") == "cyclic" for p in case.mesh.patches):
        pytest.skip("cyclic pairs are outside this walk")
    g = Geometry(case.mesh)
    table = reference_stencil(solver)
    states = J.STENCIL[solver]["states"]
    pcl = {k: v for k, v in reference_daoption()["maxResConLv4JacPCMat"].items()}
    s = pyDASolvers((solver + " -python").encode(), options(case), case=case)
    s.runColoring()
    for isPC in (0, 1):
        want = _walk_pattern(case, g, table, states, pc_levels=pcl if isPC else None)
        assert (s.getConnectivity(isPC) != want).nnz == 0, ("library", isPC)
        assert (J.connectivity(case, g, isPC=bool(isPC)) != want).nnz == 0, ("oracle", isPC)
    if solver == "DASimpleFoam":
        big = channel_case(9, 9, 9)
        gb = Geometry(big.mesh)
        full = _walk_pattern(big, gb, table, states)
        pc = _walk_pattern(big, gb, table, states, pc_levels=pcl)
        N = gb.nC
        c = (4 * 9 + 4) * 9 + 4  # the centre cell: every ring is complete
        ctr = gb.C[c]
        assert np.allclose(ctr, gb.C.mean(axis=0), atol=0.2 * np.ptp(gb.C, axis=0).max())
        f_int = next(f for f in range(gb.nIF) if c in (int(big.mesh.owner[f]), int(big.mesh.neighbour[f])))
        rl = lambda A, r: int(A.indptr[r + 1] - A.indptr[r])  # noqa: E731
        assert (rl(full, 3 * c), rl(full, 3 * N + c), rl(full, 4 * N + c), rl(full, 5 * N + f_int)) == (95, 275, 52, 149)
        assert (rl(pc, 3 * N + c), rl(pc, 5 * N + f_int)) == (161, 71)


# ------------------------------------------------------------------------------------------------- model constants
def test_spalart_allmaras_constants_equal_the_reference():
    sa = strip_comments(ref_text("src/adjoint/DAModel/DATurbulenceModel/DASpalartAllmaras.C"))
    ref = {m.group(1): float(m.group(2)) for m in re.finditer(r'lookupOrAddToDict\(\s*"(\w+)"\s*,\s*this->coeffDict_\s*,\s*([0-9.eE+-]+)\)', sa)}
    assert set(ref) == {"sigmaNut", "kappa", "Cb1", "Cb2", "Cw2", "Cw3", "Cv1", "Cs"}
    assert re.search(r"Cw1_\(Cb1_ / sqr\(kappa_\) \+ \(1\.0 \+ Cb2_\) / sigmaNut_\)", sa)
    for k, v in ref.items():
        assert OR.SA[k] == v, k
    assert OR.SA["Cw1"] == ref["Cb1"] / ref["kappa"] ** 2 + (1.0 + ref["Cb2"]) / ref["sigmaNut"]
    # the product's kernel constants (csrc/das_kernels.hpp)
    hdr = open(os.path.join(ROOT, "dafoam_amd", "csrc", "das_kernels.hpp")).read()
    mac = {m.group(1): float(m.group(2)) for m in re.finditer(r"#define SA_(\w+) ([0-9.eE+-]+)\s*$", hdr, flags=re.M)}
    assert mac == {"SIGMA": ref["sigmaNut"], "KAPPA": ref["kappa"], "CB1": ref["Cb1"], "CB2": ref["Cb2"], "CW2": ref["Cw2"], "CW3": ref["Cw3"],
                   "CV1": ref["Cv1"], "CS": ref["Cs"]}
    assert re.search(r"#define SA_CW1 \(SA_CB1 / \(SA_KAPPA \* SA_KAPPA\) \+ \(1\.0 \+ SA_CB2\) / SA_SIGMA\)", hdr)
    # functions of the model as the reference writes them (DASpalartAllmaras.C:124-172): fv1, fv2, the r clip at 10, Cs clip
    assert re.search(r"chi3 / \(chi3 \+ pow3\(Cv1_\)\)", sa) and re.search(r"1\.0 - chi / \(1\.0 \+ chi \* fv1\)", sa)
    assert re.search(r"scalar\(10\.0\)", sa) and re.search(r"Cs_ \* Omega", sa)


def test_spalding_wall_function_settings_equal_the_reference():
    wf = strip_comments(ref_text("src/adjoint/DAMisc/nutUSpaldingWallFunctionDF/nutUSpaldingWallFunctionFvPatchScalarFieldDF.C"))
    assert re.search(r"min\(kappa_ \* magUp\[facei\] / ut, 50\)", wf)              # the exponent clip both restatements carry
    assert re.search(r"tolerance_ != 1\.e-14", wf) and OR.WF["tol"] == 1e-14
    # the DF variant iterates to the root: maxIter 1000 (the stock OpenFOAM field: 10) - found by this very test in round 3,
    # both restatements had carried the stock value
    assert len(re.findall(r'maxIter_\(1000\)|lookupOrDefault<label>\("maxIter", 1000\)', wf)) == 2 and OR.WF["maxIter"] == 1000
    assert len(re.findall(r'tolerance_\(1\.e-14\)|lookupOrDefault<scalar>\("tolerance", 1\.e-14\)', wf)) == 2
    hdr = open(os.path.join(ROOT, "dafoam_amd", "csrc", "das_kernels.hpp")).read()
    assert re.search(r"#define DAS_SPALDING_MAXITER 1000", hdr) and re.search(r"err <= 1e-14", hdr)
    assert re.search(r"kappa = 0\.41, E = 9\.8", hdr) and (OR.WF["kappa"], OR.WF["E"]) == (0.41, 9.8)
    assert re.search(r"sqr\(calcUTau\(magGradU\)\) / \(magGradU \+ ROOTVSMALL\) - nuw", wf)  # nut_w = max(0, ut^2/(|dU/dn| + ROOTVSMALL) - nu)


# ------------------------------------------------------------------------------------------------- option surface
def test_every_daoption_default_equals_the_reference():
    from dafoam_amd.pyDAFoam import DAOPTION

    ref = reference_daoption()
    mine = vars(DAOPTION())
    assert len(ref) >= 50
    for name, val in ref.items():
        assert name in mine, f"option {name} (dafoam/pyDAFoam.py) is missing from dafoam_amd.pyDAFoam.DAOPTION"
        assert mine[name] == val and type(mine[name]) is type(val), name
    extra = set(mine) - set(ref)
    assert all(k.startswith("amd") for k in extra), extra  # additions carry the amd prefix


def test_incompressible_pressure_bounds_adjustment_like_the_reference():
    src = ref_text("dafoam/pyDAFoam.py")
    assert re.search(r'self\.defaultOptions\["primalVarBounds"\]\[1\]\["pMin"\] = -50000\.0', src)
    from dafoam_amd.pyDAFoam import DAOPTION

    o = DAOPTION()
    assert (o.primalVarBounds["pMin"], o.primalVarBounds["pMax"]) == (20000.0, 500000.0)


# ------------------------------------------------------------------------------------------------- Krylov configuration
def test_krylov_configuration_claims_match_dalineareqn():
    le = strip_comments(ref_text("src/adjoint/DALinearEqn/DALinearEqn.C"))
    assert re.search(r"KSPType kspObjectType = KSPGMRES;", le) and re.search(r"KSPSetType\(ksp, kspObjectType\)", le)
    assert "KSP_GMRES_CGS_REFINE_IFNEEDED" in le            # amd.gmresOrthogonalization "cgs" restates this rule
    assert "KSPGMRESModifiedGramSchmidtOrthogonalization" in le and "useMGSO" in le
    assert re.search(r"KSPSetPCSide\(ksp, PC_RIGHT\)", le)
    assert "KSP_NORM_UNPRECONDITIONED" in le
    assert "PCASM" in le and "PCASMSetOverlap" in le and "PCILU" in le and "PCFactorSetLevels" in le
    assert "MAT_SHIFT_NONZERO" in le and "KSPRICHARDSON" in le
    # failure rule (DALinearEqn.C:422-434): fail only when BOTH ratios exceed gmresTolDiff - gmres_end() in csrc/das_device.hip
    m = re.search(r"if \(relResRatio > resDiff && absResRatio > resDiff\)\s*\{[^}]*return 1;", le, flags=re.S)
    assert m, "failure rule not found in the reference"
    assert re.search(r'relResRatio = finalResNorm / initResNorm / daOption_\.getSubDictOption<scalar>\("adjEqnOption", "gmresRelTol"\)', le)
    dev = open(os.path.join(ROOT, "dafoam_amd", "csrc", "das_device.hip")).read()
    assert re.search(r"return \(relRatio > diff && absRatio > diff\) \? 1 : 0;", dev)
    # defaults the solve runs with
    adj = reference_daoption()["adjEqnOption"]
    assert (adj["gmresRestart"], adj["gmresMaxIters"], adj["gmresRelTol"], adj["gmresAbsTol"], adj["gmresTolDiff"]) == (1000, 1000, 1e-6, 1e-14, 1e2)
    assert (adj["asmOverlap"], adj["pcFillLevel"], adj["jacMatReOrdering"], adj["globalPCIters"], adj["localPCIters"]) == (1, 1, "rcm", 0, 1)


def test_fd_step_scaling_and_transposed_insert_like_dapartderiv():
    pd = strip_comments(ref_text("src/adjoint/DAPartDeriv/DAPartDeriv.C"))
    assert re.search(r"MatSetValue\(jacMat, colI, rowI, val, INSERT_VALUES\)", pd)  # transposed insert (DAPartDeriv.C:194-197)
    assert re.search(r"jacLowerBound", pd)
    assert reference_daoption()["adjPartDerivFDStep"] == {"State": 1e-6}
    # the phi columns are scaled by the face area in the perturbation (DAPartDeriv.C:283-311) - oracle.jacobian.state_scales
    assert re.search(r"magSf", pd)


# ------------------------------------------------------------------------------------------------- dump files
def test_petsc_dump_names_and_viewers_like_dautility():
    """The binary layout itself is PETSc's (not defined under /root/reference): big-endian, Vec classid 1211214, Mat classid
    1211216 - dafoam_amd.petsc_io writes exactly that (tests/test_host_cpu.py checks the bytes).  What the reference does
    define is WHO writes WHAT under which name."""
    ut = strip_comments(ref_text("src/adjoint/DAUtility/DAUtility.C"))
    for fn in ("writeMatrixBinary", "readMatrixBinary", "writeVectorBinary", "readVectorBinary"):
        assert re.search(r"void DAUtility::" + fn, ut), fn
    assert len(re.findall(r'fileNameStream << prefix << "\.bin";', ut)) >= 4
    assert "PetscViewerBinaryOpen" in ut and "MatView" in ut and "VecView" in ut and "FILE_MODE_WRITE" in ut
    jc = strip_comments(ref_text("src/adjoint/DAJacCon/DAJacCon.C"))
    assert len(re.findall(r'word fileName = modelType_ \+ "Coloring" \+ postFix \+ "_" \+ Foam::name\(nProcs\);', jc)) == 3  # exists / calc / read: dRdWColoring_<nProcs>.bin
    from dafoam_amd import petsc_io

    assert petsc_io.VEC_CLASSID == 1211214 and petsc_io.MAT_CLASSID == 1211216
    sol = strip_comments(ref_text("src/adjoint/DASolver/DASolver.C"))
    assert re.search(r'matName = "dRdWT";', sol) and re.search(r'matName = "dRdWTPC";', sol)
    assert re.search(r'writeJacobians\.found\("dRdWT"\) \|\| writeJacobians\.found\("all"\)', sol)


# methods of the reference's Cython class that this mirror does not provide, with the reason (kept in step with the reference by
# test_pydasolvers_api_surface: a method that appears upstream must be implemented or listed here)
NOT_IMPLEMENTED = {
    "calcCouplingFaceCoords": "aerostructural coupling surfaces (DAOutputForceCoupling / thermal coupling) - outside the adjoint hot path",
    "calcPCMatWithFvMatrix": "preconditioner from fvMatrix coefficients for the fixed-point adjoint (SURVEY 8(f) rank 4, not built)",
    "runFPAdj": "fixed-point adjoint (DASimpleFoam.C:189-1400; SURVEY 8(f) rank 4, not built)",
    "solveAdjointFP": "fixed-point adjoint (SURVEY 8(f) rank 4, not built)",
    "getInitStateVals": "prints the initial field values of the OpenFOAM case (diagnostic of the OpenFOAM front end)",
    "setPrimalInitialConditions": "primalInitCondition option of the OpenFOAM front end: the states arrive through the FoamCase / updateOFFields here",
    "initTensorFlowFuncs": "TensorFlow regression models (DARegression) - outside the hot path",
    "getNRegressionParameters": "regression models (DARegression) - outside the hot path",
    "meanStatesToStates": "unsteady time-averaged states (DAInputFieldUnsteady / time-accurate adjoint) - outside the hot path",
    "updateInputFieldUnsteady": "unsteady field inputs - outside the hot path",
    "writeSensMapSurface": "sensitivity-map post-processing output",
    "writeSensMapField": "sensitivity-map post-processing output",
}


def test_pydasolvers_api_surface():
    """Drop-in surface: every method of the reference's pyDASolvers class (src/pyDASolvers/pyDASolvers.pyx) is either provided by
    dafoam_amd.pyDASolvers.pyDASolvers under the same name or listed in NOT_IMPLEMENTED with the reason."""
    from dafoam_amd.pyDASolvers import pyDASolvers

    src = open(os.path.join(REF, "src", "pyDASolvers", "pyDASolvers.pyx")).read()
    cls = src[src.index("cdef class pyDASolvers"):]
    ref = sorted(set(n for n in re.findall(r"^\s+def\s+(\w+)\s*\(", cls, flags=re.M) if not n.startswith("__")))
    assert len(ref) >= 60
    mine = set(dir(pyDASolvers))
    missing = [n for n in ref if n not in mine and n not in NOT_IMPLEMENTED]
    assert not missing, f"reference methods neither implemented nor listed: {missing}"
    stale = [n for n in NOT_IMPLEMENTED if n in mine or n not in ref]
    assert not stale, f"NOT_IMPLEMENTED entries that are implemented or no longer upstream: {stale}"
    assert len([n for n in ref if n in mine]) >= 55


PYDAFOAM_NOT_IMPLEMENTED = {
    "_solverRegistry": "module-level table of OpenFOAM solver names of the Cython build",
    "_initializeOptions": "option initialisation lives in __init__ / _initOption here",
    "_initializeComm": "MPI communicator set-up: one process per GPU is launched by torch.distributed (dafoam_amd/distributed.py)",
    "_readOFGrid": "polyMesh reader of the OpenFOAM case directory: dafoam_amd.foam_io.read_polymesh",
    "_writeDecomposeParDict": "decomposePar front end: the mesh is partitioned in memory (dafoam_amd/distributed.py)",
    "_writeOpenFoamHeader": "decomposePar front end",
    "runDecomposePar": "decomposePar front end: ShardedAdjointGeneral.scattered() partitions in memory",
    "deletePrevPrimalSolTime": "time-directory housekeeping of the OpenFOAM front end",
    "renameSolution": "time-directory housekeeping of the OpenFOAM front end",
    "deformDynamicMesh": "unsteady dynamic-mesh cases - outside the hot path",
    "readDynamicMeshPoints": "unsteady dynamic-mesh cases - outside the hot path",
    "calcFFD2XvSeeds": "forward-mode (ADF build) seeds through pyGeo / IDWarp - the ADF build is not mirrored",
    "setPrimalInitialConditions": "primalInitCondition option of the OpenFOAM front end: states arrive through the FoamCase / setStates",
    "getNRegressionParameters": "regression models (DARegression) - outside the hot path",
}


def test_pydafoam_api_surface():
    """Every method of the reference's PYDAFOAM class (dafoam/pyDAFoam.py) is provided by dafoam_amd.pyDAFoam.PYDAFOAM under the
    same name or listed in PYDAFOAM_NOT_IMPLEMENTED with the reason."""
    from dafoam_amd.pyDAFoam import PYDAFOAM

    tree = ast.parse(open(os.path.join(REF, "dafoam", "pyDAFoam.py")).read())
    ref = [f.name for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "PYDAFOAM" for f in c.body if isinstance(f, ast.FunctionDef)]
    assert len(ref) >= 45
    mine = set(dir(PYDAFOAM))
    missing = [n for n in ref if n not in mine and n not in PYDAFOAM_NOT_IMPLEMENTED]
    assert not missing, f"reference methods neither implemented nor listed: {missing}"
    stale = [n for n in PYDAFOAM_NOT_IMPLEMENTED if n in mine or n not in ref]
    assert not stale, stale
