"""GPU tier (-m gpu): the HIP path, called through the C-ABI / pyDASolvers mirror, against the oracle on the
same seeded inputs.  Tolerances: residuals 1e-12 (fp64, different summation order), dual-number Jacobian
entries 1e-10, adjoint vector psi <= 1e-6 relative (BASELINE.json north_star)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from common import NORM_STATES, blocks, norm_states, options, relerr, unrolled_maps
from dafoam_amd.meshgen import (bench_channel_case, channel_case, periodic_channel_case, renumber_case, rho_channel_case, scalar_transport_case, simple_T_channel_case,
                                turbo_channel_case)
from oracle import jacobian as J
from oracle import linear as OL
from oracle.foam_mesh import Geometry
from oracle.residual import residual

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make(case, **extra):
    from dafoam_amd.pyDAFoam import PYDAFOAM

    return PYDAFOAM(options=options(case, **extra), case=case)


_PRIMAL_CACHE = {}


def converged_case(dims, wall_function=False, **kw):
    """Channel case whose state is a CONVERGED primal (oracle SIMPLE solver): the adjoint is linearised about a
    converged flow in the reference (solve_linear runs after solve_nonlinear, mphys_dafoam.py:314-433)."""
    from oracle.primal import solve_primal

    key = (dims, wall_function, tuple(sorted(kw.items())))
    if key not in _PRIMAL_CACHE:
        case = channel_case(*dims, wall_function=wall_function, perturb=0.0, **kw)
        g = Geometry(case.mesh)
        W, hist = solve_primal(case, g, max_iters=800, tol=1e-11)
        assert np.all(hist[-1] < 1e-7 * hist[0] + 1e-12), hist[-1]
        case.states = W
        _PRIMAL_CACHE[key] = case
    return _PRIMAL_CACHE[key]


def oracle_mats(case, g, pc_mode="fd"):
    sc = J.state_scales(case, g, norm_states(case))
    con = J.connectivity(case, g)
    col, _ = J.greedy_coloring(con)
    A = J.jacobian_colored(case, g, case.states, con, col, sc, mode="cs", lower_bound=0)
    return sc, con, col, A


@pytest.mark.parametrize("wall_function", [False, True])
@pytest.mark.parametrize("isPC", [0, 1])
def test_residual_parity_simplefoam(wall_function, isPC):
    case = channel_case(9, 8, 7, wall_function=wall_function)
    g = Geometry(case.mesh)
    D = make(case)
    R = np.zeros(case.states.size)
    D.solver.calcResiduals(isPC, R)
    Ro = residual(case, g, case.states, isPC=bool(isPC))
    for nm, sl in blocks(case, g):
        assert relerr(R[sl], Ro[sl]) < 1e-12, nm


def test_cell_face_split_kernels_equal_the_monolithic_cell_kernel():
    """Round 6 (amd.cellFaceSplit, off by default): k_fcoef + k_bcoef + k_cell2 against k_cell on the device - residual values, a dual-number
    Jacobian-transpose product and the FD PC matrix - on the bump channel with a wall function and on the NACA0012 O-grid."""
    from dafoam_amd.meshgen import naca0012_case
    from dafoam_amd.pyDASolvers import Mat

    for case in (channel_case(9, 8, 7, wall_function=True), naca0012_case(32, 10, 4, span=0.4, first_cell=1e-3)):
        out = []
        for split in (0, 1):
            D = make(case, amd={"cellFaceSplit": split, "pcUpwindBlend": 0.5})
            R = np.zeros(case.states.size)
            D.solver.getResiduals(R)
            psi = np.random.default_rng(2).standard_normal(R.size)
            prod = np.zeros(R.size)
            D.solverAD.calcJacTVecProduct("states", "stateVar", case.states, "residuals", "residual", psi, prod)
            D.solver.runColoring()
            pc = Mat()
            D.solver.calcdRdWT(1, pc)
            out.append((R, prod, pc.to_scipy().tocsr()))
        assert relerr(out[1][0], out[0][0]) < 1e-13 and relerr(out[1][1], out[0][1]) < 1e-12
        assert abs(out[1][2] - out[0][2]).max() <= 1e-6 * abs(out[0][2]).max()  # finite differences of two summation orders


def test_residual_parity_scalar_transport_config0():
    # BASELINE.json configs[0]: 18x17x16 = 4896-cell box
    case = scalar_transport_case()
    g = Geometry(case.mesh)
    D = make(case)
    R = np.zeros(case.states.size)
    D.solver.getResiduals(R)
    assert relerr(R, residual(case, g, case.states)) < 1e-13


def test_unstructured_renumbering_gpu():
    """Randomly renumbered (genuinely unstructured) mesh: residual, dual Jacobian and adjoint against the oracle."""
    from dafoam_amd.pyDASolvers import Mat

    case = renumber_case(converged_case((8, 6, 5), wall_function=True, lengths=(1.0, 0.2, 0.2), grading_y=2.0), seed=5)
    g = Geometry(case.mesh)
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0}, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0}, amd={"pcBlockCells": 100})
    R = np.zeros(case.states.size)
    D.solver.getResiduals(R)
    assert np.abs(R).max() < 1e-6 and np.abs(R - residual(case, g, case.states)).max() < 1e-9  # both ~0 at the fixed point
    sc, con, col, A = oracle_mats(case, g)
    D.solver.runColoring()
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    assert np.abs((M.to_scipy() - A).tocsr().data).max() <= 1e-10 * np.abs(A.data).max()
    rhs = np.zeros(A.shape[0])
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, spla.spsolve(A.tocsc(), rhs)) <= 1e-6


@pytest.mark.parametrize("wall_function", [False, True])
def test_rhosimplefoam_residual_jacobian_adjoint(wall_function):
    """DARhoSimpleFoam + SA (compressible, BASELINE configs[3] solver): residual (PC and non-PC), dual-number dRdWT and
    the adjoint vector at a converged primal state against the oracle."""
    from dafoam_amd.pyDASolvers import Mat
    from oracle.primal import solve_primal

    case = rho_channel_case(8, 6, 5, wall_function=wall_function, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    # synthetic state first: residual parity away from R = 0
    syn = rho_channel_case(8, 6, 5, wall_function=wall_function, lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.02)
    Ds = make(syn)
    R = np.zeros(syn.states.size)
    for pc in (0, 1):
        Ds.solver.calcResiduals(pc, R)
        Ro = residual(syn, g, syn.states, isPC=bool(pc))
        for nm, sl in blocks(syn, g):
            assert relerr(R[sl], Ro[sl]) < 1e-11, (pc, nm)
    W, hist = solve_primal(case, g, max_iters=800, tol=1e-11)
    case.states = W
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0}, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    D.solver.getResiduals(R)
    assert np.abs(R - residual(case, g, W)).max() < 1e-6 * np.abs(residual(syn, g, syn.states)).max()  # both ~0 at the fixed point
    sc, con, col, A = oracle_mats(case, g)
    D.solver.runColoring()
    assert (D.solver.getConnectivity(0) != con).nnz == 0
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    assert np.abs((M.to_scipy() - A).tocsr().data).max() <= 1e-10 * np.abs(A.data).max()
    rhs = np.zeros(A.shape[0])
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, spla.spsolve(A.tocsc(), rhs)) <= 1e-6


@pytest.mark.parametrize("variant", ["simple_mrf", "simple_T", "rho_mrf", "turbo", "turbo_transonic"])
def test_turbofoam_and_mrf_residual_jacobian_adjoint(variant):
    """DATurboFoam (BASELINE configs[4] solver: SIMPLEC-consistent or transonic pEqn, "h" energy with viscous and MRF
    pressure work, MRF Coriolis / relative flux / rotating walls) and DARhoSimpleFoam with MRF: residual (PC and non-PC),
    connectivity, dual-number dRdWT and the adjoint vector against the oracle (no converged turbo primal exists in the
    oracle, so the linearisation point is the synthetic state)."""
    from dafoam_amd.pyDASolvers import Mat

    if variant == "simple_mrf":  # DASimpleFoam with MRF (DAResidualSimpleFoam.C:139,182,245)
        case = channel_case(8, 6, 5, wall_function=True, lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.02)
        case.mrf = {"omega": (20.0, 0.0, 0.0), "origin": (0.0, -0.3, 0.0), "nonRotatingPatches": ["inlet", "outlet", "top"]}
        case.simple_consistent = True
    elif variant == "simple_T":  # DASimpleFoam with the optional passive T field (DAResidualSimpleFoam.C:215-235)
        case = simple_T_channel_case(8, 6, 5, wall_function=True, lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.02)
    else:
        kw = {"rho_mrf": dict(solver_name="DARhoSimpleFoam"), "turbo": {}, "turbo_transonic": dict(transonic=True)}[variant]
        case = turbo_channel_case(8, 6, 5, wall_function=True, lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.02, **kw)
    g = Geometry(case.mesh)
    W = case.states
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0, "gmresMaxIters": 600, "gmresRestart": 300},
             jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    R = np.zeros(W.size)
    for pc in (0, 1):
        D.solver.calcResiduals(pc, R)
        Ro = residual(case, g, W, isPC=bool(pc))
        for nm, sl in blocks(case, g):
            assert relerr(R[sl], Ro[sl]) < 1e-11, (pc, nm)
    sc, con, col, A = oracle_mats(case, g)
    D.solver.runColoring()
    assert (D.solver.getConnectivity(0) != con).nnz == 0
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    assert np.abs((M.to_scipy() - A).tocsr().data).max() <= 1e-10 * np.abs(A.data).max()
    # exact forward/transposed identity on this solver too
    rng = np.random.default_rng(2)
    v, a = rng.standard_normal(W.size), rng.standard_normal(W.size)
    Jv, pa = np.zeros(W.size), np.zeros(W.size)
    D.solver.calcJacVecProduct(v, Jv)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", a, pa)
    assert abs(a @ Jv - pa @ v) <= 1e-11 * np.linalg.norm(a) * np.linalg.norm(Jv)
    rhs = np.zeros(A.shape[0])
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, spla.spsolve(A.tocsc(), rhs)) <= 1e-6


@pytest.mark.parametrize("solver", ["DASimpleFoam", "DARhoSimpleFoam"])
def test_force_function_and_dFdW(solver):
    """DAFunctionForce restated: value and the state-scaled gradient dFdW (adjoint right-hand side) from the coloured
    dual-number pass, against the oracle's complex-step gradient; then the adjoint solved with that RHS."""
    from oracle.functions import force, force_gradient

    case = channel_case(6, 5, 4, wall_function=True) if solver == "DASimpleFoam" else rho_channel_case(6, 5, 4, perturb=0.02)
    g = Geometry(case.mesh)
    fn = {"CD": {"type": "force", "source": "patchToFace", "patches": ["bottom", "top"], "directionMode": "fixedDirection",
                 "direction": [1.0, 0.0, 0.0], "scale": 2.0}}
    D = make(case, function=fn)
    Fo = force(case, g, case.states, ["bottom", "top"], [1, 0, 0], 2.0)
    assert abs(D.solver.calcFunction("CD") - Fo) <= 1e-12 * abs(Fo)
    sc = J.state_scales(case, g, norm_states(case))
    dFo = force_gradient(case, g, case.states, ["bottom", "top"], [1, 0, 0], 2.0, sc)
    dF = np.zeros(case.states.size)
    D.solverAD.calcJacTVecProduct("states", "stateVar", case.states, "CD", "function", np.array([1.0]), dF)
    assert relerr(dF, dFo) < 1e-10
    assert D.solver.getOutputSize("CD", "function") == 1
    with pytest.raises(Exception):
        D.solver.calcFunction("nope")


@pytest.mark.parametrize("solver", ["DASimpleFoam", "DATurboFoam"])
def test_patch_functions_values_and_gradients(solver):
    """moment, massFlowRate, totalPressure (and totalTemperatureRatio for the compressible solvers; the objectives of the
    reference's DATurboFoam / MRF tests): values and state-scaled dFdW against the oracle's complex-step gradients,
    including the quotient rule of the ratio function and the rotating-wall velocity of the MRF zone."""
    from oracle import functions as Fn

    turbo = solver == "DATurboFoam"
    case = turbo_channel_case(6, 5, 4, wall_function=True, perturb=0.02) if turbo else channel_case(6, 5, 4, wall_function=True, perturb=0.02)
    g = Geometry(case.mesh)
    W = case.states
    walls = ["bottom", "top"]
    fn = {
        "CMZ": {"type": "moment", "source": "patchToFace", "patches": walls, "axis": [0.0, 0.0, 1.0], "center": [0.5, 0.1, 0.0], "scale": 3.0},
        "MFR": {"type": "massFlowRate", "source": "patchToFace", "patches": ["inlet"], "scale": -1.0},
        "TP": {"type": "totalPressure", "source": "patchToFace", "patches": ["outlet"], "scale": 0.5},
    }
    ref = {
        "CMZ": lambda Wp: Fn.moment(case, g, Wp, walls, [0, 0, 1], [0.5, 0.1, 0.0], 3.0),
        "MFR": lambda Wp: Fn.mass_flow_rate(case, g, Wp, ["inlet"], -1.0),
        "TP": lambda Wp: Fn.total_pressure(case, g, Wp, ["outlet"], 0.5),
    }
    if turbo:
        fn["TTR"] = {"type": "totalTemperatureRatio", "source": "patchToFace", "patches": ["inlet", "outlet"], "inletPatches": ["inlet"],
                     "outletPatches": ["outlet"], "scale": 1.0}
        ref["TTR"] = lambda Wp: Fn.total_temperature_ratio(case, g, Wp, ["inlet"], ["outlet"], 1.4)
    D = make(case, function=fn)
    sc = J.state_scales(case, g, norm_states(case))
    for name, f in ref.items():
        Fo = f(W)
        assert abs(D.solver.calcFunction(name) - Fo) <= 1e-10 * abs(Fo), name  # sums of p ~ 1e5 terms cancel in the moment
        dFo = Fn.gradient(f, W, sc)
        dF = np.zeros(W.size)
        D.solverAD.calcJacTVecProduct("states", "stateVar", W, name, "function", np.array([1.0]), dF)
        assert np.abs(dF - dFo).max() <= 1e-10 * np.abs(dFo).max(), name
    if not turbo:
        with pytest.raises(Exception, match="compressible"):
            make(case, function={"TTR": {"type": "totalTemperatureRatio", "patches": ["inlet", "outlet"], "inletPatches": ["inlet"],
                                         "outletPatches": ["outlet"]}})


@pytest.mark.parametrize("variant", ["translational", "turbo_sector"])
def test_cyclic_pair_residual_jacobian_adjoint(variant):
    """Cyclic (coupled) patch pairs through the GPU path - a translational pair (DASimpleFoam) and the compressor-passage
    configuration (DATurboFoam, annular sector with a ROTATIONAL pair, MRF about the axis, rotating hub) - against the
    UNCHANGED oracle on the three-fold unrolled twin of the periodic block (an ordinary mesh whose middle copy sees its
    periodic images as real neighbours):
      * residuals (PC / non-PC);
      * the assembled dual-number dRdWT: for random periodic-consistent state perturbations v (phi_front = -phi_back, the
        only states a cyclic case can take) A^T v equals the oracle's complex-step directional derivative of the unrolled
        residual along the unrolled image of s o v, restricted to the middle copy;
      * the adjoint vector: psi^T (dR/dW s o v) = rhs^T v for the same oracle derivatives (A psi = rhs tested against the
        oracle Jacobian action, no GPU quantity on the reference side)."""
    from dafoam_amd.meshgen import unroll_periodic_vector
    from dafoam_amd.pyDASolvers import Mat

    kw = {} if variant == "translational" else dict(sector=(0.5, 0.12), solver_name="DATurboFoam", mrf_omega=60.0)
    c1 = periodic_channel_case(6, 5, 5, wall_function=True, **kw)
    c3 = periodic_channel_case(6, 5, 5, copies=3, wall_function=True, **kw)
    idx, sgn = unrolled_maps(c1, c3)
    g1, g3 = Geometry(c1.mesh), Geometry(c3.mesh)
    W = c1.states
    n = W.size
    D = make(c1, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0, "gmresMaxIters": 600, "gmresRestart": 300},
             jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    R = np.zeros(n)
    for pc in (0, 1):
        D.solver.calcResiduals(pc, R)
        ref = residual(c3, g3, c3.states, isPC=bool(pc))[idx] * sgn
        for nm, sl in blocks(c1, g1):
            assert relerr(R[sl], ref[sl]) < 1e-11, (pc, nm)
    sc = J.state_scales(c1, g1, norm_states(c1))
    D.solver.runColoring()
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    A = M.to_scipy()
    rhs = np.zeros(n)
    rhs[0 : 3 * g1.nC : 3] = g1.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0
    slp = {p.name: slice(p.start, p.start + p.size) for p in c1.mesh.patches}
    F1 = c1.mesh.n_faces
    dth = None if variant == "translational" else kw["sector"][1]
    rng = np.random.default_rng(5)
    eps = 1e-30
    for _ in range(12):
        v = rng.standard_normal(n)
        vphi = v[n - F1:]
        vphi[slp["front"]] = -vphi[slp["back"]] * sc[n - F1:][slp["back"]] / sc[n - F1:][slp["front"]]  # s o v periodic-consistent
        W3 = unroll_periodic_vector(c1.mesh, c3.mesh, W + 1j * eps * (sc * v), 3, dtheta=dth)
        dR = (residual(c3, g3, W3).imag / eps)[idx] * sgn   # dR_i/dW . (s o v) on the middle copy, oracle only
        Av = A.T @ v
        assert np.abs(Av - dR).max() <= 1e-9 * np.abs(dR).max()
        assert abs(psi @ dR - rhs @ v) <= 1e-6 * max(abs(rhs @ v), np.linalg.norm(psi) * np.linalg.norm(dR) * 1e-3)


def test_normalize_residuals_option():
    # DAMacroFunctions.H:28-51: residuals not listed are volume-integrated / not area-divided
    case = channel_case(5, 5, 4)
    g = Geometry(case.mesh)
    D = make(case, normalizeResiduals=["URes", "nuTildaRes"])
    R = np.zeros(case.states.size)
    D.solver.getResiduals(R)
    Ro = residual(case, g, case.states, normalize=("URes", "nuTildaRes"))
    assert relerr(R, Ro) < 1e-12
    # calcPrimalResidualStatistics (reference DASolver.C:745-946): per-state norm2 / mean / max of the same residuals
    st = D.solver.calcPrimalResidualStatistics("calc")
    b = dict(blocks(case, g))
    assert np.allclose(st["URes"]["norm2"], np.sqrt((Ro[b["U"]].reshape(-1, 3) ** 2).sum(0)), rtol=1e-11)
    assert abs(st["pRes"]["max"] - np.abs(Ro[b["p"]]).max()) <= 1e-11 * np.abs(Ro[b["p"]]).max()
    assert abs(st["phiRes"]["mean"] - np.abs(Ro[b["phi"]]).mean()) <= 1e-11 * np.abs(Ro[b["phi"]]).mean()
    assert abs(st["totalResNorm2"] - np.linalg.norm(Ro)) <= 1e-11 * np.linalg.norm(Ro)


@pytest.mark.parametrize("solver", ["DASimpleFoam", "DAScalarTransportFoam"])
def test_golden_fixture(solver):
    name, case = ("oracle_channel_443.npz", channel_case(4, 4, 3)) if solver == "DASimpleFoam" else ("oracle_scalar_543.npz", scalar_transport_case(5, 4, 3))
    z = np.load(os.path.join(GOLD, name))
    D = make(case, adjEqnOption={"gmresRelTol": 1e-12, "printInfo": 0})
    R = np.zeros(z["W"].size)
    D.solver.getResiduals(R)
    assert relerr(R, z["R"]) < 1e-12
    D.solver.calcResiduals(1, R)
    assert relerr(R, z["R_pc"]) < 1e-12
    prod = np.zeros(R.size)
    D.solverAD.calcJacTVecProduct("states", "stateVar", z["W"], "residuals", "residual", z["seed"], prod)
    assert relerr(prod, z["dRdWTPsi"]) < 1e-10
    psi, fail = D.solveAdjoint(z["rhs"])
    assert fail == 0 and relerr(psi, z["psi"]) < 1e-6


@pytest.mark.parametrize("wall_function", [False, True])
def test_drdwt_dual_matches_oracle_complex_step(wall_function):
    from dafoam_amd.pyDASolvers import Mat

    case = channel_case(6, 6, 5, wall_function=wall_function)
    g = Geometry(case.mesh)
    sc, con, col, A = oracle_mats(case, g)
    D = make(case, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    D.solver.runColoring()
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    Ag = M.to_scipy()
    assert Ag.shape == A.shape
    d = (Ag - A).tocsr()
    assert np.abs(d.data).max() <= 1e-10 * np.abs(A.data).max()
    # PC matrix (reduced stencil, upwind) by dual numbers vs oracle complex step on the PC pattern
    Mp = Mat()
    D.solver.calcdRdWT(1, Mp, mode=1)
    P = J.jacobian_colored(case, g, case.states, J.connectivity(case, g, isPC=True), col, sc, mode="cs", isPC=True, lower_bound=0)
    d = (Mp.to_scipy() - P).tocsr()
    assert np.abs(d.data).max() <= 1e-10 * np.abs(P.data).max()


def test_drdwt_pc_finite_difference_like_reference():
    """Reference behaviour for dRdWTPC: coloured one-sided FD with delta=1e-6 (DAPartDeriv.C:412-456) and the
    jacLowerBounds filter (:192).  GPU-FD vs oracle-FD differ only by round-off amplified by 1/delta."""
    from dafoam_amd.pyDASolvers import Mat

    case = channel_case(6, 6, 5)
    g = Geometry(case.mesh)
    sc, con, col_o, A = oracle_mats(case, g)
    D = make(case)
    D.solver.runColoring()
    col, _ = D.solver.getColoring()
    M = Mat()
    D.solver.calcdRdWT(1, M)  # default: FD
    Pg = M.to_scipy()
    Po = J.jacobian_colored(case, g, case.states, J.connectivity(case, g, isPC=True), col.astype(np.int64), sc, mode="fd", isPC=True)
    assert np.abs((Pg - Po).tocsr().data).max() <= 1e-6 * np.abs(Po.data).max()
    Pex = J.jacobian_colored(case, g, case.states, J.connectivity(case, g, isPC=True), col.astype(np.int64), sc, mode="cs", isPC=True, lower_bound=0)
    assert np.abs((Pg - Pex).tocsr().data).max() <= 1e-4 * np.abs(Pex.data).max()
    # default lower bound 1e-30 drops exact zeros but keeps the diagonal
    assert Pg.nnz <= J.connectivity(case, g, isPC=True).nnz and np.all(Pg.diagonal() != 0)


def test_jac_t_vec_product_and_dot_product_identity():
    case = channel_case(7, 6, 5)
    g = Geometry(case.mesh)
    sc, con, col, A = oracle_mats(case, g)
    D = make(case)
    rng = np.random.default_rng(5)
    psi, v = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    prod = np.zeros_like(psi)
    D.solverAD.calcJacTVecProduct("states", "stateVar", case.states, "residuals", "residual", psi, prod)
    assert relerr(prod, A @ psi) < 1e-11
    # <psi, J (s*v)> = <D_s J^T psi, v>, J v from the oracle's complex step (independent of any assembled matrix)
    Jv = residual(case, g, case.states + 1j * 1e-30 * (sc * v)).imag / 1e-30
    assert abs(psi @ Jv - prod @ v) <= 1e-10 * abs(psi @ Jv)
    with pytest.raises(Exception):
        D.solverAD.calcJacTVecProduct("x", "volCoord", case.states, "r", "residual", psi, prod)


@pytest.mark.parametrize("kind", ["simple", "rho", "renumbered", "compacted"])
def test_packed_vector_rows_operator_equals_csr(kind):
    """The Krylov operator stores the U rows as group rows (one column list, three value planes; csrc/das_opmat.hpp): dRdW^T psi
    through the packed kernels == the plain CSR kernel (amd.opPackVector 0) == the oracle's matrix.  'compacted': a
    jacLowerBounds filter that removes entries component-wise breaks the shared lists - the operator must fall back."""
    case = {"simple": lambda: channel_case(9, 7, 6, wall_function=True), "rho": lambda: rho_channel_case(7, 6, 5),
            "renumbered": lambda: renumber_case(channel_case(8, 7, 5), seed=3), "compacted": lambda: channel_case(7, 6, 5)}[kind]()
    g = Geometry(case.mesh)
    extra = {"jacLowerBounds": {"dRdW": 1e-3, "dRdWPC": 1e-30}} if kind == "compacted" else {}
    sc = J.state_scales(case, g, norm_states(case))
    con = J.connectivity(case, g)
    col, _ = J.greedy_coloring(con)
    A = J.jacobian_colored(case, g, case.states, con, col, sc, mode="cs", lower_bound=0)
    rng = np.random.default_rng(11)
    out = {}
    for pack in (1, 0):
        D = make(case, amd={"opPackVector": pack}, **extra)
        D.solver.runColoring()
        D.solverAD.initializedRdWTMatrixFree()
        prods = []
        for _ in range(3):
            psi = rng.standard_normal(A.shape[0])
            prod = np.zeros_like(psi)
            D.solverAD.calcJacTVecProduct("states", "stateVar", case.states, "residuals", "residual", psi, prod)
            prods.append((psi, prod))
        out[pack] = prods
        rng = np.random.default_rng(11)
    for (psi, p1), (_, p0) in zip(out[1], out[0]):
        assert relerr(p1, p0) < 1e-14
        if kind != "compacted":
            # (compressible: the entries span ten orders of magnitude and agree with the oracle to 1e-10 of the LARGEST one -
            # test_rhosimplefoam_residual_jacobian_adjoint - which bounds a random product at ~1e-6 of its norm)
            assert relerr(p1, A @ psi) < (1e-5 if kind == "rho" else 1e-11)


@pytest.mark.parametrize("wall_function", [False, True])
def test_adjoint_vector_parity_simplefoam(wall_function):
    case = converged_case((10, 8, 6), wall_function=wall_function, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    # R(W*) = 0: the HIP residual vanishes at the oracle's converged SIMPLE fixed point
    D0 = make(case)
    R = np.zeros(case.states.size)
    D0.solver.getResiduals(R)
    assert np.abs(R).max() < 1e-6
    sc, con, col, A = oracle_mats(case, g)
    rhs = np.zeros(A.shape[0])
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi_o = spla.spsolve(A.tocsc(), rhs)
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0, "gmresMaxIters": 800})
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0
    assert relerr(psi, psi_o) <= 1e-6
    info = D.ksp.info()
    assert info["res"] <= 1e-9 * info["res0"] and info["iters"] > 0
    hist = D.ksp.history()
    assert hist.size == info["iters"] + 1
    # oracle GMRES on the oracle matrices reaches the same psi (PC independent fixed point)
    P = J.jacobian_colored(case, g, case.states, J.connectivity(case, g, isPC=True), col, sc, mode="fd", isPC=True)
    x, oinfo = OL.gmres(OL.CSR(A).matvec, rhs, OL.ILU(P, fill=0).solve, rel_tol=1e-10, restart=400)
    assert relerr(psi, x) <= 1e-6


def test_adjoint_scalar_transport_config0():
    case = scalar_transport_case()
    g = Geometry(case.mesh)
    sc, con, col, A = oracle_mats(case, g)
    rhs = g.V / g.V.sum()
    psi_o = spla.spsolve(A.tocsc(), rhs)
    D = make(case, adjEqnOption={"gmresRelTol": 1e-12, "printInfo": 0})
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, psi_o) <= 1e-8


@pytest.mark.parametrize("dims,block,overlap,fill,fp32", [((6, 6, 5), 4096, 0, 0, 0), ((8, 8, 6), 100, 1, 0, 0), ((8, 8, 6), 150, 1, 1, 0),
                                                          ((8, 8, 6), 100, 2, 1, 0), ((24, 20, 16), 1024, 1, 0, 0), ((8, 8, 6), 150, 1, 1, 1)])
def test_ras_ilu_apply_matches_oracle(dims, block, overlap, fill, fp32):
    """The GPU preconditioner apply (restricted additive Schwarz over RCB blocks + `overlap` cell rings, ILU(fill),
    level-scheduled in LDS) against the oracle's ILU(k) (oracle/csrc/oracle_linalg.c) on the same dRdWTPC matrix
    with the same blocks and the same cell-by-cell ordering."""
    from dafoam_amd.pyDASolvers import KSP, Mat

    case = channel_case(*dims, grading_y=2.0)
    g = Geometry(case.mesh)
    N, F = g.nC, g.nF
    D = make(case, amd={"pcType": "ras", "pcCoarseAggregates": 0, "pcBlockCells": block, "pcFactorFP32": fp32}, adjEqnOption={"asmOverlap": overlap, "pcFillLevel": fill, "printInfo": 0})
    D.solver.runColoring()
    pc = Mat()
    D.solver.calcdRdWT(1, pc)
    ksp = KSP()
    D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    perm, off = ksp.blocks()
    n = perm.size
    assert sorted(perm.tolist()) == list(range(n))
    P = pc.to_scipy().tocsr()
    owned_faces = [[] for _ in range(N)]
    for f in range(F):
        owned_faces[g.own[f]].append(f)
    x = np.random.default_rng(0).standard_normal(n)
    y_o = np.zeros(n)
    for b in range(off.size - 1):
        core_states = perm[off[b] : off[b + 1]]
        core = np.zeros(N, bool)
        core[core_states[core_states < 3 * N] // 3] = True
        ext = core.copy()
        for _ in range(overlap):
            ext = ext | (g.cellCells @ ext.astype(np.int8) > 0)
        idx, own_mask = [], []
        for c in np.nonzero(ext)[0]:
            st = [3 * c, 3 * c + 1, 3 * c + 2, 3 * N + c, 4 * N + c] + [5 * N + f for f in owned_faces[c]]
            idx += st
            own_mask += [core[c]] * len(st)
        idx, own_mask = np.array(idx), np.array(own_mask)
        z = OL.ILU(P[idx][:, idx], fill=fill).solve(x[idx])
        y_o[idx[own_mask]] = z[own_mask]
    y = ksp.applyPC(D.solver, x)
    assert relerr(y, y_o) < (1e-2 if fp32 else 1e-9)  # fp32 factor storage: preconditioner-only approximation


@pytest.mark.parametrize("dims,fp32,solver", [((6, 6, 5), 0, "simple"), ((9, 8, 7), 0, "simple"), ((24, 20, 16), 0, "simple"), ((9, 8, 7), 1, "simple"),
                                               ((8, 6, 5), 0, "rho"), ((18, 17, 16), 0, "scalar")])
def test_node_block_ilu_apply_matches_oracle(dims, fp32, solver):
    """The default preconditioner (amd.pcType "bilu": ONE node-block ILU(0) of dRdWTPC per GPU - the reference's ASM+ILU
    stack, DALinearEqn.C:199-299, with one sub-domain per rank - factorised on the device level by level and applied by
    sync-free sweeps) against the oracle: the SCALAR ILU(0) kernel (oracle/csrc/oracle_linalg.c) on the explicitly filled
    node pattern in the NATURAL cell order, and (small meshes) the dense-block numpy restatement in the processing order."""
    from dafoam_amd.pyDASolvers import KSP, Mat

    case = {"simple": lambda: channel_case(*dims, grading_y=2.0), "rho": lambda: rho_channel_case(*dims, perturb=0.02),
            "scalar": lambda: scalar_transport_case(*dims)}[solver]()
    D = make(case, amd={"pcFactorFP32": fp32, "pcCoarseAggregates": 0}, adjEqnOption={"printInfo": 0})
    D.solver.runColoring()
    pc = Mat()
    D.solver.calcdRdWT(1, pc)
    ksp = KSP()
    D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    S = ksp.pcStructure()
    nu = S["nodeUnk"]
    P = pc.to_scipy().tocsr()
    n = P.shape[0]
    assert np.array_equal(np.sort(nu[nu >= 0]), np.arange(n))          # every unknown in exactly one slot
    lev = np.repeat(np.arange(S["lvlPtr"].size - 1), np.diff(S["lvlPtr"]))
    for p in range(0, nu.shape[0], max(1, nu.shape[0] // 200)):        # level order keeps coupled nodes ordered
        cols = S["bcol"][S["bptr"][p]:S["bptr"][p + 1]]
        assert np.all(lev[cols[cols < p]] < lev[p]) and np.all(lev[cols[cols > p]] > lev[p])
    x = np.random.default_rng(0).standard_normal(n)
    y = ksp.applyPC(D.solver, x)
    B = OL.NodeBlockILU.__new__(OL.NodeBlockILU)
    B.n, B.nu, B.bptr, B.bcol = n, nu, S["bptr"].astype(np.int64), S["bcol"].astype(np.int64)
    y_twin = B.scalar_twin(P, node_order=np.argsort(S["natural"]))(x)
    tol = 1e-2 if fp32 else 1e-9
    assert relerr(y, y_twin) < tol
    if nu.shape[0] < 3000:
        Bo = OL.NodeBlockILU(P, nu, S["bptr"], S["bcol"])
        assert relerr(y, Bo.solve(x)) < tol
        # the only matrix entries outside the node pattern couple two "late" nodes (extra boundary faces of two cells)
        nPrimary = int(np.sum(np.any((nu >= 0) & (nu < n - case.mesh.n_faces), axis=1))) if solver != "scalar" else nu.shape[0]
        assert np.all(S["natural"][Bo.dropped_pairs.ravel()] >= nPrimary)
    # applying it twice gives the same answer (the sweeps re-arm their sentinels / tickets every call)
    assert np.array_equal(y, ksp.applyPC(D.solver, x))


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5, 6, 7])
def test_node_block_ilu_elimination_orders_match_oracle(order, monkeypatch):
    """Round 6: the elimination order of the cells is a choice of the factorisation (mesh numbering, reverse Cuthill-McKee = jacMatReOrdering
    "rcm", Cuthill-McKee, mesh numbering backwards, along / against the mean flow, Cuthill-McKee grown from the most upstream cell in both
    directions) - the stability check of das_create_ml_rksp_matrix_free switches between them.  Every order: a valid level structure, and
    the device apply equals the oracle's scalar ILU(0) on the filled node pattern eliminated in THAT order."""
    from dafoam_amd import _capi
    from dafoam_amd.pyDASolvers import KSP, Mat

    monkeypatch.setenv("DAS_BILU_ORDER", str(order))
    case = channel_case(9, 8, 7, grading_y=2.0)
    D = make(case, amd={"pcCoarseAggregates": 0, "pcStabilityLimit": 1e300}, adjEqnOption={"printInfo": 0})
    D.solver.runColoring()
    pc = Mat()
    D.solver.calcdRdWT(1, pc)
    ksp = KSP()
    D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    est, used = C.c_double(-1.0), C.c_int(-1)
    _capi.check(_capi.lib().das_ksp_get_pc_stability(ksp.handle, C.byref(est), C.byref(used)))
    assert used.value == order and 0.0 <= est.value < 1e12
    S = ksp.pcStructure()
    nu = S["nodeUnk"]
    P = pc.to_scipy().tocsr()
    n = P.shape[0]
    assert np.array_equal(np.sort(nu[nu >= 0]), np.arange(n))
    lev = np.repeat(np.arange(S["lvlPtr"].size - 1), np.diff(S["lvlPtr"]))
    for p in range(0, nu.shape[0], max(1, nu.shape[0] // 200)):
        cols = S["bcol"][S["bptr"][p]:S["bptr"][p + 1]]
        assert np.all(lev[cols[cols < p]] < lev[p]) and np.all(lev[cols[cols > p]] > lev[p])
    x = np.random.default_rng(order).standard_normal(n)
    y = ksp.applyPC(D.solver, x)
    B = OL.NodeBlockILU.__new__(OL.NodeBlockILU)
    B.n, B.nu, B.bptr, B.bcol = n, nu, S["bptr"].astype(np.int64), S["bcol"].astype(np.int64)
    assert relerr(y, B.scalar_twin(P, node_order=np.argsort(S["natural"]))(x)) < 1e-9


@pytest.mark.parametrize("variant", ["subdomains", "pick_min_of_all_orders"])
def test_adjoint_with_subdomain_ilus_and_order_fallback_reaches_the_same_psi(variant):
    """Round 6: (a) amd.pcSubdomains K - restricted additive Schwarz inside the GPU: K node-block ILUs on RCB blocks of the cells + asmOverlap
    rings, each with its own elimination order, sweeps on K streams; (b) a stability limit no factorisation passes - every candidate order is
    factorised and the one with the smallest estimate kept.  psi is preconditioner-independent: both reach the direct solve's."""
    from dafoam_amd import _capi

    case = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    sc, con, col, A = oracle_mats(case, g)
    rhs = np.zeros(A.shape[0])
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi_o = spla.spsolve(A.tocsc(), rhs)
    amd = {"pcSubdomains": 3} if variant == "subdomains" else {"pcStabilityLimit": 1e-30, "pcOrderCandidates": "124567"}
    D = make(case, amd=amd, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0, "gmresMaxIters": 800})
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, psi_o) <= 1e-6
    est, used = C.c_double(-1.0), C.c_int(-1)
    _capi.check(_capi.lib().das_ksp_get_pc_stability(D.ksp.handle, C.byref(est), C.byref(used)))
    assert est.value > 0.0 and 0 <= used.value <= 7
    # twice: the tickets re-arm
    psi2, fail2 = D.solveAdjoint(rhs)
    assert fail2 == 0 and relerr(psi2, psi_o) <= 1e-6
    if variant == "subdomains":
        # the merged multi-block factorisation against the oracle's host restatement of the same structure: an overlap unknown sits in one
        # node per block (all read the right-hand side), only the owner block's copy (nodeOut) writes
        K, ests = (C.c_int * 16)(), (C.c_double * 16)()
        assert _capi.lib().das_ksp_get_pc_subdomains(D.ksp.handle, K, ests) == 3
        S = D.ksp.pcStructure()
        nu, nout = S["nodeUnk"], S["nodeOut"]
        n = psi_o.size
        cnt = np.bincount(nu[nu >= 0], minlength=n)
        assert cnt.min() == 1 and cnt.max() >= 2                          # overlap rings: some unknowns in several nodes ...
        assert np.array_equal(np.sort(nout[nout >= 0]), np.arange(n))     # ... every unknown written exactly once
        assert np.all((nout == nu) | (nout == -1))
        Kh = OL.OmpKrylov(4)
        x = np.random.default_rng(1).standard_normal(n)
        D2 = make(case, amd=dict(amd, pcCoarseAggregates=0), adjEqnOption={"printInfo": 0})
        D2.solver.runColoring()
        from dafoam_amd.pyDASolvers import KSP, Mat
        pc2 = Mat()
        D2.solver.calcdRdWT(1, pc2)
        k2 = KSP()
        D2.solverAD.createMLRKSPMatrixFree(pc2, k2)
        S2 = k2.pcStructure()
        P2 = pc2.to_scipy().tocsr()
        Kh.set_pc_bilu((P2.indptr.astype(np.int64), P2.indices.astype(np.int32), P2.data), S2)
        assert relerr(k2.applyPC(D2.solver, x), Kh.pc_solve(x)) < 1e-9


@pytest.mark.parametrize("nrhs", [2, 3, 8])
def test_block_gmres_multi_rhs_matches_direct_solve(nrhs):
    """Block (multi right-hand-side) GMRES - SpMM + tall-skinny fp64 MFMA Gram-Schmidt (das_block.hpp): every column of the
    block solution against a sparse direct solve of the oracle Jacobian (<= 1e-6, the north-star tolerance), and against
    the single-system GMRES of the same library."""
    case = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    sc, con, col, A = oracle_mats(case, g)
    n = case.states.size
    rng = np.random.default_rng(3)
    rhs = np.zeros((n, nrhs))
    rhs[0 : 3 * g.nC : 3, 0] = g.V                      # drag-like functional
    rhs[1 : 3 * g.nC : 3, 1] = g.V                      # lift-like functional
    for r in range(2, nrhs):
        rhs[:, r] = rng.standard_normal(n) * (r + 1.0)
    rhs *= sc[:, None]
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "gmresAbsTol": 1e-16, "gmresMaxIters": 400, "gmresRestart": 400, "printInfo": 0})
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and psi.shape == (n, nrhs)
    lu = spla.splu(A.tocsc())
    for r in range(nrhs):
        assert relerr(psi[:, r], lu.solve(rhs[:, r])) <= 1e-6, r
    psi0, fail0 = D.solveAdjoint(np.ascontiguousarray(rhs[:, 0]))
    assert fail0 == 0 and relerr(psi[:, 0], psi0) <= 1e-6
    # the preconditioner of the block path takes all right-hand sides through ONE pair of sweeps (k_bilu_sweep_m, groups of 4 /
    # 2 / 1); column by column (amd.blockBatchedPC 0) is the same operator: same iteration count, same psi
    it_b = D.ksp.info()["iters"]
    D1 = make(case, adjEqnOption={"gmresRelTol": 1e-10, "gmresAbsTol": 1e-16, "gmresMaxIters": 400, "gmresRestart": 400, "printInfo": 0},
              amd={"blockBatchedPC": 0})
    psi1, fail1 = D1.solveAdjoint(rhs)
    assert fail1 == 0 and relerr(psi1, psi) <= 1e-8


def test_sutherland_transport_residual_and_jacobian():
    """thermophysicalProperties transport "sutherland" (DAResidual.C:264-293) through the GPU path: residual and
    dual-number dRdWT against the oracle."""
    from dafoam_amd.pyDASolvers import Mat

    case = rho_channel_case(7, 6, 5, perturb=0.02)
    case.thermo = dict(case.thermo, transport="sutherland", As=1.4792e-06, Ts=116.0)
    g = Geometry(case.mesh)
    D = make(case, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    R = np.zeros(case.states.size)
    D.solver.getResiduals(R)
    Ro = residual(case, g, case.states)
    for nm, sl in blocks(case, g):
        assert relerr(R[sl], Ro[sl]) < 1e-11, nm
    const = rho_channel_case(7, 6, 5, perturb=0.02)
    assert relerr(residual(const, g, const.states), Ro) > 1e-6  # the law is active
    sc, con, col, A = oracle_mats(case, g)
    D.solver.runColoring()
    M = Mat()
    D.solver.calcdRdWT(0, M, mode=1)
    assert np.abs((M.to_scipy() - A).tocsr().data).max() <= 1e-10 * np.abs(A.data).max()


def test_parity_tool_oracle_dump_vs_gpu_engine(tmp_path):
    """tests/parity_from_dafoam_dump.py end to end: dumps written by the ORACLE in the reference's on-disk formats (OpenFOAM
    ASCII case at a converged primal, PETSc-binary dRdWT / dRdWTPC / colouring, adjoint_* fields) compared with the GPU
    product path reading the same case directory - the recipe a real DAFoam dump goes through."""
    import parity_from_dafoam_dump as P

    case_dir = P.write_self_dump(str(tmp_path), engine="oracle", dims=(6, 5, 4))
    ok, rows = P.compare(case_dir, str(tmp_path), engine="gpu", tol=1e-6, verbose=False)
    assert ok, [r for r in rows if not r[2]]


@pytest.mark.parametrize("mode", ["additive", "deflated"])
def test_two_level_pc_apply_and_iteration_gain(mode):
    """Two-level preconditioner: node-block ILU(0) + piecewise-constant pressure coarse space (E = Z^T P Z, RCB aggregates).
    The apply against its numpy restatement on top of the oracle's incomplete factorisation, and - the reason it exists -
    fewer GMRES iterations than the one-level preconditioner on the same system, same psi."""
    from dafoam_amd.pyDASolvers import KSP, Mat
    import scipy.sparse as sp

    case = converged_case((14, 10, 8), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    N = g.nC
    opts = dict(adjEqnOption={"gmresRelTol": 1e-8, "gmresMaxIters": 500, "printInfo": 0})
    D = make(case, amd={"pcCoarseAggregates": 24, "pcCoarseMode": mode}, **opts)
    D.solver.runColoring()
    pc = Mat()
    D.solver.calcdRdWT(1, pc)
    ksp = KSP()
    D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    D.solverAD.initializedRdWTMatrixFree()
    nagg, agg = ksp.coarse(N)
    assert nagg == 24 and agg.min() == 0 and agg.max() == 23 and np.bincount(agg).min() >= N // 24 - 1
    P = pc.to_scipy().tocsr()
    n = P.shape[0]
    Z = sp.csr_matrix((np.ones(N), (3 * N + np.arange(N), agg)), shape=(n, nagg))
    Einv = np.linalg.inv((Z.T @ (P @ Z)).toarray())
    S = ksp.pcStructure()
    B = OL.NodeBlockILU(P, S["nodeUnk"], S["bptr"], S["bcol"])
    x = np.random.default_rng(1).standard_normal(n)
    cvec = Z @ (Einv @ (Z.T @ x))
    if mode == "additive":
        y_o = B.solve(x) + cvec
    else:
        Mop = Mat()
        D.solver.calcdRdWT(0, Mop, mode=1)
        y_o = B.solve(x - Mop.to_scipy() @ cvec) + cvec
    y = ksp.applyPC(D.solver, x)
    assert relerr(y, y_o) < 1e-9
    if mode == "deflated":
        # the A-DEF1 term A (Z u) came from the precomputed sparse A Z (k_az_build / k_az_apply) - the numpy line above used the whole
        # operator: this IS the direct comparison of the sparse A Z; the same apply with amd.pcCoarseSparseAZ 0 runs the full product
        from dafoam_amd import _capi
        assert _capi.lib().das_ksp_coarse_sparse_az_active(ksp.handle) == 1
        D0 = make(case, amd={"pcCoarseAggregates": 24, "pcCoarseMode": mode, "pcCoarseSparseAZ": 0}, **opts)
        D0.solver.runColoring()
        pc0 = Mat()
        D0.solver.calcdRdWT(1, pc0)
        k0 = KSP()
        D0.solverAD.createMLRKSPMatrixFree(pc0, k0)
        D0.solverAD.initializedRdWTMatrixFree()
        y0 = k0.applyPC(D0.solver, x)
        assert _capi.lib().das_ksp_coarse_sparse_az_active(k0.handle) == 0 and relerr(y, y0) < 1e-11
    sc = J.state_scales(case, g, norm_states(case))
    rhs = np.zeros(n)
    rhs[0 : 3 * N : 3] = g.V
    rhs *= sc
    psi2, fail2 = D.solveAdjoint(rhs)
    it2 = D.ksp.info()["iters"]
    D1 = make(case, amd={"pcCoarseAggregates": 0}, **opts)
    psi1, fail1 = D1.solveAdjoint(rhs)
    it1 = D1.ksp.info()["iters"]
    assert fail1 == 0 and fail2 == 0 and relerr(psi2, psi1) < 1e-6
    assert it2 < it1, (it1, it2)


def _with_inlet(case, Umag, aoa_deg):
    import copy

    c2 = copy.copy(case)
    c2.bcs = copy.deepcopy(case.bcs)
    a = aoa_deg * np.pi / 180.0
    c2.bcs["inlet"]["U"] = (case.bcs["inlet"]["U"][0], (Umag * np.cos(a), Umag * np.sin(a), 0.0))
    return c2


def test_total_derivative_patch_velocity_vs_primal_fd():
    """End to end, like the reference's regression tests (tests/runRegTests_DASimpleFoam*.py compare adjoint totals
    with forward-mode totals): dCD/d[UMag, AoA] = dF/dx - psi^T dR/dx from the GPU adjoint
    (calcJacTVecProduct patchVelocity -> function/residual, DAInputPatchVelocity.C:33-135) against central
    differences of the oracle's CONVERGED primal.  Also checks each partial against the oracle."""
    from oracle.functions import force
    from oracle.primal import solve_primal

    case = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    W = case.states
    walls = [p.name for p in case.mesh.patches if p.type == "wall"]
    D = make(case, adjEqnOption={"gmresRelTol": 1e-12, "gmresAbsTol": 1e-16, "gmresMaxIters": 400, "printInfo": 0},
             function={"CD": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "fixedDirection",
                              "direction": [1.0, 0.0, 0.0], "scale": 1.0}},
             inputInfo={"patchV": {"type": "patchVelocity", "patches": ["inlet"], "flowAxis": "x", "normalAxis": "y"}})
    n = W.size
    x0 = np.array([10.0, 0.0])
    assert D.solverAD.getInputSize("patchV", "patchVelocity") == 2
    dFdW = np.zeros(n)
    D.solverAD.calcJacTVecProduct("states", "stateVar", W, "CD", "function", np.ones(1), dFdW)
    psi, fail = D.solveAdjoint(dFdW)
    assert fail == 0
    dFdx, pRx = np.zeros(2), np.zeros(2)
    D.solverAD.calcJacTVecProduct("patchV", "patchVelocity", x0, "CD", "function", np.ones(1), dFdx)
    D.solverAD.calcJacTVecProduct("patchV", "patchVelocity", x0, "residual", "residual", psi, pRx)
    total = dFdx - pRx
    # partials against the oracle (central differences of the residual / objective w.r.t. the patch table)
    for k, h in ((0, 1e-4), (1, 1e-3)):
        e = np.eye(2)[k]
        cp_, cm_ = _with_inlet(case, *(x0 + h * e)), _with_inlet(case, *(x0 - h * e))
        ref_R = psi @ ((residual(cp_, g, W) - residual(cm_, g, W)) / (2 * h))
        ref_F = (force(cp_, g, W, walls, [1.0, 0, 0]) - force(cm_, g, W, walls, [1.0, 0, 0])) / (2 * h)
        assert abs(pRx[k] - ref_R) < 1e-6 * abs(ref_R), (k, pRx[k], ref_R)
        assert abs(dFdx[k] - ref_F) < 1e-6 * abs(ref_R), (k, dFdx[k], ref_F)
    # totals against finite differences of the converged primal
    for k, h, tol in ((0, 1e-2, 2e-5), (1, 0.05, 1e-2)):  # AoA: upwind-direction kinks make the FD itself noisy (0.3 %)
        e = np.eye(2)[k]
        Fs = []
        for sgn in (1, -1):
            c3 = _with_inlet(case, *(x0 + sgn * h * e))
            W3, _ = solve_primal(c3, g, W0=W, max_iters=2000, tol=1e-12)
            Fs.append(force(c3, g, W3, walls, [1.0, 0, 0]))
        fd = (Fs[0] - Fs[1]) / (2 * h)
        assert abs(total[k] - fd) < tol * abs(fd), (k, total[k], fd)
    # DAInput::run semantics: the patch value is assigned by the product call; unsupported patch types are errors
    from dafoam_amd._capi import DASError
    D2 = make(case, inputInfo={"w": {"type": "patchVar", "patches": ["bottom"], "varName": "p", "varType": "scalar"}})
    with pytest.raises(DASError, match="patch type not valid"):
        D2.solver.setSolverInput("w", "patchVar", 1, np.array([1.0]))


def test_shape_total_derivative_vs_primal_fd():
    """Shape sensitivity end to end (the purpose of the adjoint): design variable = height of the wall bump, i.e. a
    displacement field dX of the mesh points.  dF/db = dF/dX.dX - psi^T dR/dX.dX with psi, both directional mesh
    products (updateOFMesh + central difference of the metrics, calcVolCoordDirectionalProduct) and dFdW from the GPU,
    against central differences of the oracle's CONVERGED primal on the deformed meshes (frozen wall distance, like the
    reference's meshWaveFrozen).  The partial products are also checked against the oracle's geometry differences."""
    import copy

    from oracle.functions import force
    from oracle.primal import solve_primal

    dims, kw, b0 = (10, 8, 6), dict(lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.0), 0.1
    base = converged_case(dims, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(base.mesh)
    W = base.states
    walls, d = ["bottom", "top"], [1.0, 0.0, 0.0]

    def case_at(b):
        c = channel_case(*dims, bump=b, **kw)
        c.y_wall = base.y_wall
        return c

    h = 1e-4
    dX = ((case_at(b0 + h).mesh.points - case_at(b0 - h).mesh.points) / (2 * h)).ravel()
    D = make(base, adjEqnOption={"gmresRelTol": 1e-12, "gmresAbsTol": 1e-16, "gmresMaxIters": 400, "printInfo": 0},
             function={"CD": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "fixedDirection",
                              "direction": d, "scale": 1.0}})
    n = W.size
    dFdW = np.zeros(n)
    D.solverAD.calcJacTVecProduct("states", "stateVar", W, "CD", "function", np.ones(1), dFdW)
    psi, fail = D.solveAdjoint(dFdW)
    assert fail == 0
    pF = D.solverAD.calcVolCoordDirectionalProduct(dX, "CD", "function", np.ones(1), eps=1e-6)
    pR = D.solverAD.calcVolCoordDirectionalProduct(dX, "residual", "residual", psi, eps=1e-6)
    # the mesh is back where it was
    X = np.zeros(dX.size)
    D.solverAD.getOFMeshPoints(X)
    assert np.array_equal(X, base.mesh.points.ravel())
    # partials against the oracle on the two displaced meshes
    eps = 1e-6
    ca, cb = copy.copy(base), copy.copy(base)
    ca.mesh, cb.mesh = copy.copy(base.mesh), copy.copy(base.mesh)
    ca.mesh.points = base.mesh.points + eps * dX.reshape(-1, 3)
    cb.mesh.points = base.mesh.points - eps * dX.reshape(-1, 3)
    ga, gb = Geometry(ca.mesh), Geometry(cb.mesh)
    refR = psi @ ((residual(ca, ga, W) - residual(cb, gb, W)) / (2 * eps))
    refF = (force(ca, ga, W, walls, d) - force(cb, gb, W, walls, d)) / (2 * eps)
    assert abs(pR - refR) <= 1e-6 * abs(refR) and abs(pF - refF) <= 1e-6 * abs(refF)
    total = pF - pR
    hh = 3e-3
    Fs = []
    for sgn in (1, -1):
        c3 = case_at(b0 + sgn * hh)
        g3 = Geometry(c3.mesh)
        W3, _ = solve_primal(c3, g3, W0=W, max_iters=3000, tol=1e-12)
        Fs.append(force(c3, g3, W3, walls, d))
    fd = (Fs[0] - Fs[1]) / (2 * hh)
    assert abs(total - fd) <= 3e-4 * abs(fd), (total, fd)
    # the same total from the FULL product vectors over all points (calcJacTVecProduct(volCoord -> ...), what the reference
    # hands to its geometry parametrisation): dF/db = (dF/dX - [dR/dX]^T psi) . dX/db
    gF, gR = np.zeros(dX.size), np.zeros(dX.size)
    D.solverAD.calcJacTVecProduct("x", "volCoord", X, "CD", "function", np.ones(1), gF)
    D.solverAD.calcJacTVecProduct("x", "volCoord", X, "residual", "residual", psi, gR)
    assert abs((gF - gR) @ dX - fd) <= 3e-4 * abs(fd) and abs((gF - gR) @ dX - total) <= 1e-4 * abs(total)


def test_device_geometry_passes_equal_the_host_metrics():
    """The three metric kernels of the volCoord product (das_geom.hpp bodies: faces, cells, weights) give, for moved points, the
    metrics the host computes after updateOFMesh - bump + skew (non-orthogonal), and a rotational cyclic pair."""
    for case, moved in ((channel_case(9, 8, 7, wall_function=True), channel_case(9, 8, 7, wall_function=True, bump=0.13, skew=0.1)),
                        (periodic_channel_case(7, 6, 6, wall_function=True, sector=(0.5, 0.12), solver_name="DATurboFoam", mrf_omega=60.0), None)):
        D = make(case, normalizeStates=norm_states(case))
        X = (moved.mesh.points if moved is not None else case.mesh.points * np.array([1.0, 1.03, 0.98])).ravel()
        fg, cg = D.solver.deviceGeometry(X)
        # solver geometry untouched by the debug call; now move the host mesh and compare
        geo0 = D.solver.geometry()
        D.solver.updateOFMesh(X)
        geo = D.solver.geometry()
        assert not np.array_equal(geo0["V"], geo["V"])
        nIF = case.mesh.n_internal_faces
        assert relerr(fg[:, 0:3].ravel(), geo["Sf"]) < 1e-13 and relerr(fg[:, 9:12].ravel(), geo["Cf"]) < 1e-13
        assert relerr(cg[:, 0:3].ravel(), geo["C"]) < 1e-13 and relerr(cg[:, 3], geo["V"]) < 1e-13
        assert relerr(fg[:nIF, 4], geo["w"]) < 1e-12 and relerr(fg[:nIF, 5], geo["nonOrthDeltaCoeffs"]) < 1e-12
        assert np.abs(fg[:nIF, 6:9].ravel() - geo["nonOrthCorr"]).max() < 1e-12 and relerr(fg[nIF:, 5], geo["bDeltaCoeffs"]) < 1e-12
        assert np.array_equal(cg[:, 4], case.y_wall)  # frozen wall distance


@pytest.mark.parametrize("mode", ["dual", "fd"])
def test_volcoord_full_product_vector_on_the_device(mode):
    """calcJacTVecProduct(volCoord -> residual | function) (reference DASolver.C:1690-1839, DAInputVolCoord): the FULL product
    vector over all mesh points on the device.  amd.volCoordMode "dual" (default): Dual<1> points -> metrics -> residual, one
    pass per point colour and axis, exact; "fd": coloured central differences.  Checked (1) entry by entry against difference
    quotients of the ORACLE's residual / force on meshes with one moved point ("fd": the SAME step, 1e-8; "dual": a step small
    enough that no limiter switch lies inside it, 1e-5), (2) contracted with a displacement field against the directional product
    (one FD of the whole mesh), (3) for bit-reproducibility, (4) that states, points and metrics are what they were afterwards,
    (5) moment and area-averaged objectives."""
    import copy

    from oracle.functions import force

    base = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    case = channel_case(10, 8, 6, lengths=(1.0, 0.2, 0.2), grading_y=2.0, perturb=0.0, bump=0.1)
    case.y_wall, case.states = base.y_wall, base.states  # not converged on the bumped mesh: R != 0, all terms active
    W = case.states
    walls, d = ["bottom", "top"], [1.0, 0.0, 0.0]
    fns = {"CD": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "fixedDirection", "direction": d, "scale": 1.0},
           "CM": {"type": "moment", "source": "patchToFace", "patches": walls, "axis": [0.0, 0.0, 1.0], "center": [0.25, 0.05, 0.0], "scale": 1.0},
           "PT": {"type": "totalPressure", "source": "patchToFace", "patches": ["inlet"], "scale": 1.0}}
    D = make(case, function=fns, amd={"volCoordMode": mode})
    S = D.solverAD
    n, P3 = W.size, 3 * case.mesh.n_points
    rng = np.random.default_rng(3)
    seeds = rng.standard_normal(n)
    X0 = case.mesh.points.ravel().copy()
    R0 = np.zeros(n)
    S.getResiduals(R0)
    assert S.getInputSize("x", "volCoord") == P3
    pR = np.zeros(P3)
    S.calcJacTVecProduct("x", "volCoord", X0, "residual", "residual", seeds, pR)
    info = S._volCoordInfo
    assert 50 < info["colors"] < 600 and info["passes"] == (3 if mode == "dual" else 6) * info["colors"]
    pF = np.zeros(P3)
    S.calcJacTVecProduct("x", "volCoord", X0, "CD", "function", np.ones(1), pF)
    # (4)
    X = np.zeros(P3)
    S.getOFMeshPoints(X)
    R1 = np.zeros(n)
    S.getResiduals(R1)
    assert np.array_equal(X, X0) and np.array_equal(R0, R1)
    # (3)
    pR2 = np.zeros(P3)
    S.calcJacTVecProduct("x", "volCoord", X0, "residual", "residual", seeds, pR2)
    assert np.array_equal(pR, pR2)
    # (1) single entries against the oracle: interior, wall and corner points
    steps = S.pointInfluence()["steps"]
    mm = case.mesh
    wall_pts = np.unique(np.concatenate([mm.face_pts[mm.face_ptr[f]:mm.face_ptr[f + 1]] for pt in mm.patches if pt.name in walls
                                         for f in range(pt.start, pt.start + pt.size)]))
    pick = list(rng.choice(case.mesh.n_points, 6, replace=False)) + list(rng.choice(wall_pts, 6, replace=False))
    scaleR, scaleF = np.abs(pR).max(), np.abs(pF).max()

    def oracle_fd(p, ax, h):
        vals = []
        for sgn in (1.0, -1.0):
            c2 = copy.copy(case)
            c2.mesh = copy.copy(case.mesh)
            Xp = case.mesh.points.copy()
            Xp[p, ax] += sgn * h
            c2.mesh.points = Xp
            g2 = Geometry(c2.mesh)
            vals.append(np.array([seeds @ residual(c2, g2, W), force(c2, g2, W, walls, d)]))
        return (vals[0] - vals[1]) / (2 * h)

    for p in pick:
        for ax in range(3):
            if mode == "fd":
                # the same difference quotient from the oracle (the residual is only piecewise smooth in the point coordinates:
                # a reference with another step differs by 1e-6 .. 1e-5 at some wall points, by much more next to a limiter switch)
                refR, refF = oracle_fd(p, ax, steps[p])
                tolr, tols = 1e-8, 1e-9
            else:
                refR, refF = oracle_fd(p, ax, 0.02 * steps[p])
                tolr, tols = 1e-5, 1e-6
            assert abs(pR[3 * p + ax] - refR) <= tolr * abs(refR) + tols * scaleR, (p, ax, pR[3 * p + ax], refR)
            assert abs(pF[3 * p + ax] - refF) <= tolr * abs(refF) + tols * scaleF, (p, ax, pF[3 * p + ax], refF)
    assert 0 < np.count_nonzero(pF) < P3  # the force only feels the points near the walls (here: all but the mid-channel layers)
    # (2) contraction with a smooth displacement field (the directional product is ONE central difference of the whole mesh with a
    #     step that is not small against the wall cells: agreement to 1e-6 of the sum of the terms for "fd", a few 1e-5 for "dual")
    Xr = case.mesh.points
    dX = np.stack([0.3 * np.sin(3 * Xr[:, 1] / 0.2) * Xr[:, 0], 0.2 * Xr[:, 0] * (1 - Xr[:, 0]) + 0 * Xr[:, 1], 0.1 * np.cos(2 * Xr[:, 0])], axis=1).ravel()
    tol = 1e-6 if mode == "fd" else 1e-4
    dirR = S.calcVolCoordDirectionalProduct(dX, "residual", "residual", seeds, eps=1e-6)
    assert abs(pR @ dX - dirR) <= tol * np.abs(pR * dX).sum(), (pR @ dX, dirR)
    # (5)
    for nm in ("CD", "CM", "PT"):
        pM = np.zeros(P3)
        S.calcJacTVecProduct("x", "volCoord", X0, nm, "function", np.ones(1), pM)
        dirM = S.calcVolCoordDirectionalProduct(dX, nm, "function", np.ones(1), eps=1e-6)
        assert np.abs(pM).max() > 0 and abs(pM @ dX - dirM) <= (1e-7 if mode == "fd" else 1e-5) * np.abs(pM * dX).sum(), (nm, pM @ dX, dirM)
    v0 = S.calcFunction("CM")
    S.calcJacTVecProduct("x", "volCoord", X0, "CM", "function", np.ones(1), pM)
    assert S.calcFunction("CM") == v0  # the host copy of the moment arms is untouched / back, and the sums are deterministic


def test_volcoord_dual_and_difference_modes_agree_away_from_switches():
    """The exact ("dual") and the central-difference ("fd") mode of the mesh-sensitivity product.  With the default step (1e-4 of
    the smallest adjacent cell thickness) the two agree to 1e-8 of the largest entry on half of the entries and to 1e-4 on 95 % of
    them (measured: 8e-11 / 1.4e-5); the rest sit next to a switch - the V-limiter of linearUpwindV, an upwind selection, the |n_k|
    of a symmetry plane whose point leaves the plane - where a difference quotient ACROSS the switch is not the derivative
    (measured maximum: 0.6 %).  That explanation is asserted, not assumed: with a 100x smaller step the
    difference quotient converges to the dual value - fewer entries disagree and 95 % are within 1e-5 (large step: 1e-4; the measured
    numbers are printed) - so the dual value IS the derivative and the large-step disagreement is the quotient's.  CPU counterpart: test_dual_number_metrics_give_the_exact_mesh_derivative."""
    case = channel_case(9, 7, 6, wall_function=True, bump=0.1)
    n, P3 = case.states.size, 3 * case.mesh.n_points
    seeds = np.random.default_rng(8).standard_normal(n)
    X0 = case.mesh.points.ravel().copy()
    out = {}
    for key, amd in (("dual", {"volCoordMode": "dual"}), ("fd", {"volCoordMode": "fd"}), ("fd_small", {"volCoordMode": "fd", "volCoordRelStep": 1e-6})):
        D = make(case, amd=amd)
        out[key] = np.zeros(P3)
        D.solverAD.calcJacTVecProduct("x", "volCoord", X0, "residual", "residual", seeds, out[key])
    scale = np.abs(out["dual"]).max()
    err = np.abs(out["dual"] - out["fd"])
    err_s = np.abs(out["dual"] - out["fd_small"])
    print("dual vs fd(1e-4): p50/p95/max", np.percentile(err, [50, 95, 100]) / scale, " dual vs fd(1e-6): p50/p95/p99.9/max",
          np.percentile(err_s, [50, 95, 99.9, 100]) / scale, " entries > 1e-6:", int((err > 1e-6 * scale).sum()), "->", int((err_s > 1e-6 * scale).sum()))
    assert scale > 0 and np.percentile(err, 95) <= 1e-4 * scale and np.percentile(err, 50) <= 1e-8 * scale, (np.percentile(err, [50, 95, 100]), scale)
    assert err.max() <= 5e-2 * scale
    bad_large, bad_small = int((err > 1e-6 * scale).sum()), int((err_s > 1e-6 * scale).sum())
    assert bad_small <= bad_large, (bad_large, bad_small)
    assert np.percentile(err_s, 95) <= 1e-5 * scale, np.percentile(err_s, [50, 95, 99.9, 100]) / scale


def test_primal_bc_option_and_calc_output():
    """setPrimalBoundaryConditions (the primalBC option -> patch table, DASolver.C:3790-4030) and calcOutput (pyDASolvers.pyx:204:
    function value / residual vector) - small methods of the reference surface that need the device."""
    case = channel_case(6, 5, 4, wall_function=True)
    D = make(case, primalBC={"U0": {"variable": "U", "patches": ["inlet"], "value": [12.0, 0.5, 0.0]}},
             function={"CD": {"type": "force", "source": "patchToFace", "patches": ["bottom"], "directionMode": "fixedDirection", "direction": [1.0, 0.0, 0.0], "scale": 1.0}})
    S = D.solver
    n = case.states.size
    R0, R1, out1 = np.zeros(n), np.zeros(n), np.zeros(1)
    S.calcOutput("residual", "residual", R0)
    S.setPrimalBoundaryConditions(printInfo=0)
    S.calcOutput("residual", "residual", R1)
    S.calcOutput("CD", "function", out1)
    import copy

    c2 = copy.copy(case)
    c2.bcs = copy.deepcopy(case.bcs)
    c2.bcs["inlet"]["U"] = (c2.bcs["inlet"]["U"][0], np.array([12.0, 0.5, 0.0]))
    g = Geometry(case.mesh)
    assert relerr(R0, residual(case, g, case.states)) < 1e-12 and relerr(R1, residual(c2, g, case.states)) < 1e-12 and relerr(R1, R0) > 1e-3
    assert out1[0] == S.calcFunction("CD")


def test_volcoord_product_compressible_and_ratio_objective():
    """The volCoord product for DARhoSimpleFoam: residual seeds and the totalTemperatureRatio objective (a quotient of two area
    averages: differentiated through its linearisation at the base mesh) against the directional product - one central
    difference of the whole mesh through the HOST metrics."""
    case = rho_channel_case(9, 7, 6, wall_function=True)
    D = make(case, normalizeStates=norm_states(case),
             function={"TTR": {"type": "totalTemperatureRatio", "source": "patchToFace", "patches": ["inlet", "outlet"], "inletPatches": ["inlet"],
                               "outletPatches": ["outlet"], "scale": 1.0},
                       "MFR": {"type": "massFlowRate", "source": "patchToFace", "patches": ["outlet"], "scale": 1.0}})
    S = D.solverAD
    n, P3 = case.states.size, 3 * case.mesh.n_points
    X0 = case.mesh.points.ravel().copy()
    Xr = case.mesh.points
    # a displacement field that also deforms the inlet and outlet planes (the objectives live there)
    dX = np.stack([0.05 * np.sin(3 * Xr[:, 1] / 0.2) * Xr[:, 0], 0.03 * Xr[:, 1] * (1 + Xr[:, 0]), 0.02 * Xr[:, 2] * (1 + 0.5 * Xr[:, 0]) * (1 + Xr[:, 1])], axis=1).ravel()
    seeds = np.random.default_rng(5).standard_normal(n)
    for nm, ot, sd in (("residual", "residual", seeds), ("TTR", "function", np.ones(1)), ("MFR", "function", np.ones(1))):
        p = np.zeros(P3)
        S.calcJacTVecProduct("x", "volCoord", X0, nm, ot, sd, p)
        ref = S.calcVolCoordDirectionalProduct(dX, nm, ot, sd, eps=1e-6)
        assert np.abs(p).max() > 0 and abs(ref) > 1e-3 * np.abs(p * dX).sum(), nm  # a real sensitivity, not two zeros
        assert abs(p @ dX - ref) <= 1e-5 * np.abs(p * dX).sum(), (nm, p @ dX, ref)


@pytest.mark.parametrize("kind", ["simple", "rho", "scalar"])
def test_forward_mode_jac_vec_product(kind):
    """calcJacVecProduct (one dual-number residual pass) against the oracle's complex-step directional derivative, and
    the exact identity a.(J v) = (J^T a).v with the coloured transposed operator."""
    case = {"simple": lambda: channel_case(8, 7, 6, wall_function=True), "rho": lambda: rho_channel_case(7, 6, 5, perturb=0.02),  # no exact ties (max/abs kinks)
            "scalar": lambda: scalar_transport_case(8, 7, 6)}[kind]()
    g = Geometry(case.mesh)
    D = make(case)
    W = case.states
    n = W.size
    sc = J.state_scales(case, g, norm_states(case))
    rng = np.random.default_rng(5)
    v, a = rng.standard_normal(n), rng.standard_normal(n)
    Jv = np.zeros(n)
    D.solver.calcJacVecProduct(v, Jv)
    ref = residual(case, g, W + 1j * 1e-30 * sc * v).imag / 1e-30
    assert relerr(Jv, ref) < 1e-10
    pa = np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", a, pa)
    assert abs(a @ Jv - pa @ v) <= 1e-11 * np.linalg.norm(a) * np.linalg.norm(Jv)


def test_size_independent_properties_bench_size():
    """BASELINE.json's bench configuration (100x50x40 = 200k cells, 1.6 M states, 2.1e8 Jacobian non-zeros), where no
    oracle Jacobian is affordable: linearity, the dot-product identity against a central difference of the GPU
    residual is replaced by the exact forward-mode identity (every colour and every scatter slot takes part in a random
    product), and a CONVERGED adjoint solve (gmresRelTol 1e-6, the reference's defaults gmresMaxIters = gmresRestart = 1000)
    whose residual recurrence agrees with an independently recomputed true residual; the colouring is the serial first-fit
    (device kernel): no more than 450 colours."""
    case = bench_channel_case(100, 50, 40)
    D = make(case, adjEqnOption={"gmresRelTol": 1e-6, "gmresAbsTol": 1e-30, "printInfo": 0, "gmresMaxIters": 1000, "gmresRestart": 1000})
    n = case.states.size
    W = case.states
    rng = np.random.default_rng(11)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    pa, pb, pab = np.zeros(n), np.zeros(n), np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", a, pa)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", b, pb)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", 2.0 * a - 3.0 * b, pab)
    assert relerr(pab, 2.0 * pa - 3.0 * pb) < 1e-12
    g = Geometry(case.mesh)
    sc = J.state_scales(case, g, NORM_STATES)
    # exact dot-product identity: forward-mode J (s o v) from ONE dual pass (no colouring, no scatter maps) against the
    # coloured, scattered, transposed operator.  (Central differences of the residual are useless here: a random
    # perturbation flips the sign of thousands of near-zero cross-stream fluxes, i.e. upwind kinks.)
    for seed in (0, 1):
        v = np.random.default_rng(seed).standard_normal(n)
        Jv = np.zeros(n)
        D.solver.calcJacVecProduct(v, Jv)
        assert abs(a @ Jv - pa @ v) <= 1e-10 * np.linalg.norm(a) * np.linalg.norm(Jv)
    rhs = np.zeros(n)
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    info = D.ksp.info()
    assert D.solver.getColoring()[1] <= 450
    assert fail == 0 and info["res"] <= 1e-6 * info["res0"] and info["iters"] < 400, info  # two-level PC: ~190 iterations
    chk = np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", psi, chk)
    true_res = np.linalg.norm(chk - rhs)
    assert abs(true_res - info["res"]) <= 1e-6 * info["res0"], (true_res, info)


def test_config1_naca0012_200k_cells_drdwtpsi_against_the_oracle_at_size():
    """BASELINE configs[1] AS STATED: DASimpleFoam + SA on a NACA0012 O-grid of 800 x 250 x 1 = 200 k cells (1.8 M states),
    `dRdWTPsi vs CPU to 1e-6` - at size, against the ORACLE (not against the GPU itself): the residual, and for random a, v
        a . (dR/dW (s o v))_oracle  ==  ((dR/dW)^T a)_GPU . v
    with the left side from ONE complex-step evaluation of the oracle's vectorised numpy residual (no oracle Jacobian is
    affordable at this size) and the right side from the coloured, assembled, packed operator of the GPU path; tolerance 1e-6
    (north star), achieved ~1e-10.  The GPU's own forward-mode product is compared with the oracle's vector entry by entry."""
    from dafoam_amd.meshgen import naca0012_case

    case = naca0012_case(800, 250, 1)
    assert case.mesh.n_cells == 200000
    g = Geometry(case.mesh)
    W = case.states
    n = W.size
    D = make(case)
    R = np.zeros(n)
    D.solver.getResiduals(R)
    Ro = residual(case, g, W)
    for nm, sl in blocks(case, g):
        assert relerr(R[sl], Ro[sl]) < 1e-10, nm
    sc = J.state_scales(case, g, NORM_STATES)
    D.solver.runColoring()
    D.solverAD.initializedRdWTMatrixFree()
    for seed in (0, 1):
        rng = np.random.default_rng(seed)
        a, v = rng.standard_normal(n), rng.standard_normal(n)
        Jv_o = residual(case, g, W + 1j * 1e-30 * (sc * v)).imag / 1e-30
        pa = np.zeros(n)
        D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", a, pa)
        lhs, rhs_ = a @ Jv_o, pa @ v
        assert abs(lhs - rhs_) <= 1e-6 * np.linalg.norm(a) * np.linalg.norm(Jv_o) / np.sqrt(n), (lhs, rhs_)
        assert abs(lhs - rhs_) <= 1e-9 * abs(lhs), (lhs, rhs_)
        Jv = np.zeros(n)
        D.solver.calcJacVecProduct(v, Jv)
        assert relerr(Jv, Jv_o) < 1e-9


def test_naca0012_ogrid_residual_jacobian_adjoint():
    """BASELINE configs[1] mesh family (NACA0012 O-grid, stretched wall-normal cells of aspect ratio > 100, branch cut as
    internal faces): residual, dRdW^T.v and the adjoint vector of the drag-like functional against the oracle."""
    from dafoam_amd.meshgen import naca0012_case

    case = naca0012_case(56, 16, 1, first_cell=5e-4, radius=8.0, wall_function=True)
    g = Geometry(case.mesh)
    W = case.states
    n = W.size
    D = make(case, adjEqnOption={"gmresRelTol": 1e-10, "gmresMaxIters": 800, "gmresRestart": 800, "printInfo": 0},
             jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0})
    R = np.zeros(n)
    D.solver.getResiduals(R)
    Ro = residual(case, g, W)
    for nm, sl in blocks(case, g):
        assert relerr(R[sl], Ro[sl]) < 1e-10, nm
    sc, con, col, A = oracle_mats(case, g)
    v = np.random.default_rng(2).standard_normal(n)
    prod = np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", v, prod)
    assert relerr(prod, A @ v) < 1e-9
    rhs = np.zeros(n)
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    assert fail == 0 and relerr(psi, spla.spsolve(A.tocsc(), rhs)) <= 1e-6


def test_newton_krylov_primal_reaches_the_simple_fixed_point():
    """solvePrimal on the GPU (survey row f4): pseudo-transient Newton-Krylov on R(W) = 0 built from the adjoint's kernels
    (forward-mode operator, transposed node-block ILU + coarse space) against the ORACLE's SIMPLE loop (reference
    DASimpleFoam.C:123-185 restated in oracle/primal.py): same fixed point, residual dropped by ten orders; the adjoint
    linearised about the GPU primal equals the adjoint about the oracle primal."""
    from oracle.primal import solve_primal

    case = channel_case(10, 8, 6, perturb=0.0, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    Wo, hist = solve_primal(case, g, max_iters=800, tol=1e-11)
    sc = J.state_scales(case, g, norm_states(case))
    D = make(case, primalMinResTol=1e-10, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0})
    fail = D()
    info = D.primalInfo
    assert fail == 0 and info["res"] <= 1e-10 * info["res0"] and info["steps"] <= 60, info
    W = D.getStates()
    assert relerr(W / sc, Wo / sc) <= 1e-6
    R = np.zeros(W.size)
    D.solver.getResiduals(R)
    assert np.linalg.norm(residual(case, g, W)) <= 1e-8 * np.linalg.norm(residual(case, g, case.states))
    rhs = np.zeros(W.size)
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, afail = D.solveAdjoint(rhs)
    case.states = Wo
    sc2, con, col, A = oracle_mats(case, g)
    assert afail == 0 and relerr(psi, spla.spsolve(A.tocsc(), rhs)) <= 1e-5


def test_simple_sweeps_on_the_device_equal_the_oracle_and_reach_the_newton_fixed_point():
    """Round 6: das_simple_iteration - the reference's own primal loop (SIMPLE: DASimpleFoam.C:123-185, UEqnSimple.H,
    pEqnSimple.H, DASpalartAllmaras::correct) on the device.  (a) 1 and 3 sweeps from the synthetic state equal the oracle's sweeps
    (direct inner solves) to the inner solvers' tolerance, with and without the wall function and on the NACA0012 O-grid; (b) run to
    convergence, the sweeps reach the fixed point of the oracle's SIMPLE loop - the same one the Newton-Krylov primal reaches."""
    from dafoam_amd.meshgen import naca0012_case
    from oracle.primal import simple_iteration

    for case in (channel_case(7, 6, 5, perturb=0.0), channel_case(7, 6, 5, wall_function=True, perturb=0.0), naca0012_case(24, 8, 3, span=0.3, first_cell=1e-3, perturb=0.0)):
        g = Geometry(case.mesh)
        W0 = case.states.copy()
        Wo = [W0]
        for _ in range(3):
            Wo.append(simple_iteration(case, g, Wo[-1]))
        for ns in (1, 3):
            D = make(case)
            info = D.solver.simpleIteration(ns, alphaP=0.3, linTol=1e-13)
            assert info["U"] > 0 and info["p"] > 0
            W = np.zeros(W0.size)
            D.solver.getOFFields(W)
            for nm, sl in blocks(case, g):
                assert relerr(W[sl], Wo[ns][sl]) < 1e-9, (nm, ns)
    # (b) to convergence: the fixed point of the SIMPLE loop = the converged case of the adjoint tests (oracle SIMPLE, 1e-11)
    conv = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    start = channel_case(10, 8, 6, perturb=0.0, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(start.mesh)
    sc = J.state_scales(start, g, norm_states(start))
    D = make(start)
    R = np.zeros(start.states.size)
    D.solver.getResiduals(R)
    r0 = np.linalg.norm(R)
    D.solver.simpleIteration(400, alphaP=0.3, linTol=1e-10)
    D.solver.getResiduals(R)
    assert np.linalg.norm(R) < 1e-6 * r0
    W = np.zeros(R.size)
    D.solver.getOFFields(W)
    assert relerr(W / sc, conv.states / sc) < 1e-5


def test_newton_krylov_primal_compressible():
    """The same Newton-Krylov primal on DARhoSimpleFoam + SA: the fixed point of the oracle's compressible SIMPLE loop."""
    from oracle.primal import solve_primal

    case = rho_channel_case(8, 6, 5, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    Wo, hist = solve_primal(case, g, max_iters=800, tol=1e-11)
    sc = J.state_scales(case, g, norm_states(case))
    D = make(case, primalMinResTol=1e-9)
    fail = D.solvePrimal(maxSteps=80)
    info = D.primalInfo
    assert fail == 0 and info["res"] <= 1e-9 * info["res0"], info
    assert relerr(D.getStates() / sc, Wo / sc) <= 1e-6


def test_device_coloring_is_the_serial_first_fit():
    """The data-flow colouring kernel (das_color.hpp) reproduces the SERIAL first-fit colours exactly (below 20 k cells the
    host path is that serial sweep), validates like the reference demands (DAColoring::validateColoring, checked inside
    runColoring: "Conflicting Colors Found!"), and above 20 k cells needs fewer colours than the host's tile-parallel
    variant."""
    import scipy.sparse as sp

    case = channel_case(12, 10, 8, wall_function=True)
    Dd = make(case)
    Dd.solver.runColoring()
    cd, nd = Dd.solver.getColoring()
    Dh = make(case, amd={"coloringOnDevice": 0})
    Dh.solver.runColoring()
    ch, nh = Dh.solver.getColoring()
    assert nd == nh and np.array_equal(cd, ch)
    con = sp.csr_matrix(Dd.solver.getConnectivity(0))
    for i in range(0, con.shape[0], 7):  # independent validity check of a sample of rows
        c = cd[con.indices[con.indptr[i]:con.indptr[i + 1]]]
        assert c.min() >= 0 and np.unique(c).size == c.size
    big = channel_case(30, 28, 24)
    Db = make(big)
    Db.solver.runColoring()
    _, nbd = Db.solver.getColoring()
    Dbh = make(big, amd={"coloringOnDevice": 0})
    Dbh.solver.runColoring()
    _, nbh = Dbh.solver.getColoring()
    assert nbd <= nbh, (nbd, nbh)


@pytest.mark.parametrize("kind", ["simple", "rho", "cyclic", "renumbered"])
def test_speculative_device_coloring_is_valid_and_close_to_first_fit(kind):
    """The alternative device colouring (amd.coloringAlgorithm "speculative": rounds of speculative first-fit over per-net colour
    bitmaps, das_color.hpp): valid by the reference's contract - no row of dRdWCon holds two columns of one colour
    (DAColoring::validateColoring, DAColoring.C:931-1037; checked here by the ORACLE's validator on the oracle's / the library's
    connectivity) - deterministic (two runs, same colours), within 25 % of the serial first-fit's colour count, and the Jacobian
    assembled with it equals the oracle's (any valid colouring gives the same matrix)."""
    from dafoam_amd.pyDASolvers import Mat

    case = {"simple": lambda: channel_case(14, 12, 10, wall_function=True), "rho": lambda: rho_channel_case(10, 8, 7),
            "cyclic": lambda: periodic_channel_case(8, 6, 6, wall_function=True), "renumbered": lambda: renumber_case(channel_case(12, 9, 8), seed=2)}[kind]()
    D = make(case, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0}, amd={"coloringAlgorithm": "speculative"})
    D.solver.runColoring()
    cs, ns = D.solver.getColoring()
    assert cs.min() >= 0 and ns == cs.max() + 1
    con = D.solver.getConnectivity(0)
    assert J.validate_coloring(con, cs.astype(np.int64))
    D2 = make(case, amd={"coloringAlgorithm": "speculative"})
    D2.solver.runColoring()
    assert np.array_equal(D2.solver.getColoring()[0], cs)
    Df = make(case)  # the default: the serial first-fit over net bitmaps, equal to the host's serial sweep
    Df.solver.runColoring()
    cf, nf = Df.solver.getColoring()
    assert J.validate_coloring(con, cf.astype(np.int64))
    Dh = make(case, amd={"coloringOnDevice": 0})
    Dh.solver.runColoring()
    if case.mesh.n_cells < 20000:
        assert np.array_equal(Dh.solver.getColoring()[0], cf)
    assert ns <= 1.3 * nf + 8, (ns, nf)
    if kind in ("simple", "renumbered"):
        g = Geometry(case.mesh)
        sc, con_o, col, A = oracle_mats(case, g)
        M = Mat()
        D.solver.calcdRdWT(0, M, mode=1)
        assert np.abs((M.to_scipy() - A).tocsr().data).max() <= 1e-10 * np.abs(A.data).max()


def test_field_input_betaFINuTilda_product_and_total_derivative():
    """calcJacTVecProduct(field -> residual | function) for the `field` input betaFINuTilda (reference DAInputField.C,
    DASolver.C:1690-1839; the design variable of DAFoam's field inversion): the residual with a non-trivial field, the product
    psi^T dR/dbeta (ONE forward-mode pass, das_calc_dfield_product) against the oracle's complex step, dF/dbeta = 0 for a force,
    and the operator assembled after setSolverInput (the operator epoch: the cached dRdW^T must not be reused)."""
    import copy

    from dafoam_amd.pyDASolvers import Mat

    case = channel_case(8, 7, 6, wall_function=True)
    g = Geometry(case.mesh)
    N, n = g.nC, case.states.size
    rng = np.random.default_rng(4)
    beta = 1.0 + 0.2 * rng.standard_normal(N)
    D = make(case, inputInfo={"beta": {"type": "field", "fieldName": "betaFINuTilda", "fieldType": "scalar", "components": ["solver", "function"]}},
             function={"CD": {"type": "force", "source": "patchToFace", "patches": ["bottom"], "directionMode": "fixedDirection", "direction": [1.0, 0.0, 0.0], "scale": 1.0}})
    W = case.states
    D.solverAD.initializedRdWTMatrixFree()  # operator at beta = 1
    psi = rng.standard_normal(n)
    prod = np.zeros(N)
    D.solverAD.calcJacTVecProduct("beta", "field", beta, "residuals", "residual", psi, prod)
    assert np.array_equal(D.solver.getField("betaFINuTilda"), beta)
    cb = copy.copy(case)
    cb.beta_fi = beta
    R = np.zeros(n)
    D.solver.getResiduals(R)
    Ro = residual(cb, g, W)
    for nm, sl in blocks(case, g):
        assert relerr(R[sl], Ro[sl]) < 1e-12, nm
    # psi^T dR/dbeta_c: per cell, from one complex step (a row depends on its own cell's beta only)
    cc = copy.copy(case)
    cc.beta_fi = beta + 1e-30j * np.ones(N)
    dR = residual(cc, g, W.astype(complex)).imag / 1e-30
    ref = psi[4 * N : 5 * N] * dR[4 * N : 5 * N]
    assert np.abs(dR[: 4 * N]).max() == 0.0 and relerr(prod, ref) < 1e-12
    pf = np.ones(N)
    D.solverAD.calcJacTVecProduct("beta", "field", beta, "CD", "function", np.ones(1), pf)
    assert np.all(pf == 0.0)
    # the operator cached before setSolverInput(field) is stale: the next product re-assembles at the new field
    sc = J.state_scales(case, g, NORM_STATES)
    v = rng.standard_normal(n)
    pa = np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", psi, pa)
    Jv = residual(cb, g, W + 1j * 1e-30 * (sc * v)).imag / 1e-30
    assert abs(psi @ Jv - pa @ v) <= 1e-10 * abs(psi @ Jv)
    with pytest.raises(Exception, match="not implemented"):
        D2 = make(case, inputInfo={"a": {"type": "field", "fieldName": "alphaPorosity", "fieldType": "scalar"}})
        D2.solverAD.calcJacTVecProduct("a", "field", beta, "residuals", "residual", psi, prod)


def test_krylov_basis_is_mapped_on_demand_and_never_reallocated():
    """The basis address range is reserved once for the largest restart the memory budget allows and 2 GB chunks are mapped
    while the iteration advances (VmBuf, csrc/das_common.hpp): solves with different restarts on one KSP - first a short
    one, then the reference default - give the single-restart answers, and a restart that does not fit the budget is capped."""
    case = bench_channel_case(60, 30, 24)  # 43 k cells, n = 347 k: the default 1002-vector range (2.8 GB) goes through hipMalloc ...
    D = make(case, adjEqnOption={"gmresRelTol": 1e-8, "gmresAbsTol": 1e-30, "printInfo": 0, "gmresMaxIters": 1000, "gmresRestart": 30})
    n = case.states.size
    g = Geometry(case.mesh)
    rhs = np.zeros(n)
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= J.state_scales(case, g, NORM_STATES)
    psi30, fail30 = D.solveAdjoint(rhs)
    it30 = D.ksp.info()["iters"]
    D.solver.updateDAOption({"adjEqnOption": {"gmresRestart": 1000}})
    psi1k, fail1k = D.solveAdjoint(rhs)
    it1k = D.ksp.info()["iters"]
    assert fail30 == 0 and fail1k == 0 and it1k <= it30 and relerr(psi30, psi1k) < 1e-5
    # ... and a budget of 6 GB with n = 347 k states holds 2160 vectors: amd.maxKrylovBytes small forces the VM path's cap
    D2 = make(case, adjEqnOption={"gmresRelTol": 1e-8, "gmresAbsTol": 1e-30, "printInfo": 0, "gmresMaxIters": 1000, "gmresRestart": 1000},
              amd={"maxKrylovBytes": int(40 * 8 * n)})
    psic, failc = D2.solveAdjoint(rhs)
    assert failc == 0 and relerr(psic, psi1k) < 1e-5 and D2.ksp.info()["iters"] >= it1k


def test_cell_state_ordering():
    """adjStateOrdering "cell" (reference DAIndex.C:602-651): every state-length array crosses the boundary in the
    cell-by-cell ordering; results are the permuted "state"-ordering results."""
    case = converged_case((8, 6, 5), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    Ds = make(case, adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0})
    Dc = make(case, adjStateOrdering="cell", adjEqnOption={"gmresRelTol": 1e-10, "printInfo": 0})
    perm = Dc.solver._perm
    n = case.states.size
    assert np.array_equal(Dc.getStates(), case.states[perm])
    syn = channel_case(8, 6, 5, lengths=(1.0, 0.2, 0.2), grading_y=2.0).states
    Rs, Rc = np.zeros(n), np.zeros(n)
    Ds.setStates(syn)
    Dc.setStates(syn[perm])
    Ds.solver.getResiduals(Rs)
    Dc.solver.getResiduals(Rc)
    assert np.array_equal(Rc, Rs[perm])
    Ds.setStates(case.states)
    Dc.setStates(case.states[perm])
    psi = np.random.default_rng(0).standard_normal(n)
    ps, pc = np.zeros(n), np.zeros(n)
    Ds.solverAD.calcJacTVecProduct("s", "stateVar", case.states, "r", "residual", psi, ps)
    Dc.solverAD.calcJacTVecProduct("s", "stateVar", case.states[perm], "r", "residual", psi[perm], pc)
    assert relerr(pc, ps[perm]) < 1e-13
    g = Geometry(case.mesh)
    rhs = np.zeros(n)
    rhs[0 : 3 * g.nC : 3] = g.V * 10.0
    xs, fs = Ds.solveAdjoint(rhs)
    xc, fc = Dc.solveAdjoint(rhs[perm])
    assert fs == 0 and fc == 0 and relerr(xc, xs[perm]) < 1e-7


def test_unsteady_terms_scalar_transport():
    """calcdRdWOldTPsiAD (reference DASolver.C:1910-1969) for the Euler-ddt scalar transport residual and the old-time
    field setter, against complex-step derivatives of the oracle w.r.t. T_old."""
    import copy

    case = scalar_transport_case(7, 6, 5)
    g = Geometry(case.mesh)
    D = make(case)
    n = case.states.size
    psi = np.random.default_rng(1).standard_normal(n)
    out = np.zeros(n)
    D.solverAD.calcdRdWOldTPsiAD(1, psi, out)
    # oracle: dR/dT_old by complex step (diagonal), transposed product
    ref = np.zeros(n)
    c2 = copy.copy(case)
    for j in range(0, n, 7):  # sample of columns
        To = case.T_old.astype(np.complex128)
        To[j] += 1e-30j
        c2.T_old = To
        ref[j] = (residual(c2, g, case.states.astype(np.complex128)).imag / 1e-30) @ psi
    idx = np.arange(0, n, 7)
    assert relerr(out[idx], ref[idx]) < 1e-12
    D.solverAD.calcdRdWOldTPsiAD(2, psi, out)
    assert np.all(out == 0.0)
    # new time level: residual follows the updated old-time field
    T_old2 = case.T_old * 1.1 + 0.01
    D.solver.setOldTimeFields(T_old=T_old2)
    c2.T_old = T_old2
    R = np.zeros(n)
    D.solver.getResiduals(R)
    assert relerr(R, residual(c2, g, case.states)) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("restart", [1000, 25])
def test_delayed_reorthogonalisation_matches_reference_gram_schmidt(restart):
    """amd.gmresOrthogonalization "dcgs2" (default: second projection of a step fused with the first of the next one, two
    basis reads per iteration) against "cgs" (the reference's KSP_GMRES_CGS_REFINE_IFNEEDED, DALinearEqn.C:160) and against
    modified Gram-Schmidt: the same Krylov iterates - equal iteration counts, equal residual histories, equal psi."""
    case = converged_case((10, 8, 6), lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    g = Geometry(case.mesh)
    rhs = np.zeros(case.states.size)
    rhs[0 : 3 * g.nC : 3] = g.V
    out = {}
    for name, amd, adj in (("dcgs2", {"gmresOrthogonalization": "dcgs2"}, {}), ("cgs", {"gmresOrthogonalization": "cgs"}, {}),
                           ("mgs", {}, {"useMGSO": 1})):
        D = make(case, amd=dict(amd, pcCoarseAggregates=0), adjEqnOption=dict(adj, gmresRelTol=1e-10, gmresRestart=restart, gmresMaxIters=3000, printInfo=0))
        psi, fail = D.solveAdjoint(rhs)
        assert fail == 0
        out[name] = (psi, D.ksp.history(), D.ksp.info()["iters"])
        # every cycle but the last closes at exactly gmresRestart columns (KSPGMRESSetRestart, DALinearEqn.C:155): round 4 closed the
        # malloc-path cycles of "dcgs2" one column early (basis reservation restart + 2 against a request of restart + 3)
        lens = D.ksp.cycleLengths()
        assert lens.sum() == D.ksp.info()["iters"] and np.all(lens[:-1] == min(restart, 3000)) and 1 <= lens[-1] <= restart, (name, lens)
    for other in ("cgs", "mgs"):
        assert out["dcgs2"][2] == out[other][2]
        h1, h2 = out["dcgs2"][1], out[other][1]
        assert np.all(np.abs(h1 - h2) <= 1e-5 * h2 + 1e-12 * h2[0])  # (the last entries sit at the rounding level of the true residual)
        assert relerr(out["dcgs2"][0], out[other][0]) < 1e-8
    with pytest.raises(Exception, match="gmresOrthogonalization"):
        make(case, amd={"gmresOrthogonalization": "householder"}).solveAdjoint(rhs)


def test_size_independent_properties_larger_mesh():
    """At a size where the oracle Jacobian would take minutes: linearity of dRdW^T.psi, FD-vs-dual agreement of
    J^T psi via the dot-product identity with a GPU residual difference, and GMRES residual reduction."""
    case = bench_channel_case(32, 20, 16)
    D = make(case, adjEqnOption={"gmresRelTol": 1e-8, "printInfo": 0, "gmresMaxIters": 1500, "gmresRestart": 300},
             amd={"pcBlockCells": 1024})
    n = case.states.size
    rng = np.random.default_rng(7)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    pa, pb, pab = np.zeros(n), np.zeros(n), np.zeros(n)
    W = case.states
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", a, pa)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", b, pb)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", 2.0 * a - 3.0 * b, pab)
    assert relerr(pab, 2.0 * pa - 3.0 * pb) < 1e-12
    # dot-product identity with a central finite difference of the GPU residual
    g = Geometry(case.mesh)
    sc = J.state_scales(case, g, NORM_STATES)
    v = rng.standard_normal(n)
    eps = 3e-7  # small enough that (almost) no face flux changes sign (upwind kinks); FD round-off limits this check to ~1e-5
    Rp, Rm = np.zeros(n), np.zeros(n)
    D.solver.updateOFFields(W + eps * sc * v)
    D.solver.getResiduals(Rp)
    D.solver.updateOFFields(W - eps * sc * v)
    D.solver.getResiduals(Rm)
    D.solver.updateOFFields(W)
    Jv = (Rp - Rm) / (2 * eps)
    assert abs(a @ Jv - pa @ v) <= 1e-4 * abs(a @ Jv)
    rhs = np.zeros(n)
    rhs[0 : 3 * g.nC : 3] = g.V
    rhs *= sc
    psi, fail = D.solveAdjoint(rhs)
    info = D.ksp.info()
    assert fail == 0 and info["res"] <= 1e-7 * info["res0"]
    # independent check of the returned psi: ||A psi - rhs|| with a fresh product
    chk = np.zeros(n)
    D.solverAD.calcJacTVecProduct("s", "stateVar", W, "r", "residual", psi, chk)
    assert relerr(chk, rhs) <= 1e-6
