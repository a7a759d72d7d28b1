"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's patch-integral objectives (adjoint right-hand-side producers): force, moment,
massFlowRate, totalPressure, totalTemperatureRatio (file:line in the docstrings below).  Force:
  * DAFunctionForce::calcFunction          reference src/adjoint/DAFunction/DAFunctionForce.C:79-158
        F = scale * sum_{faces of the selected patches} ( S_f p_b + S_f . devRhoReff_b ) . dir
  * DATurbulenceModel::devRhoReff           reference src/adjoint/DAModel/DATurbulenceModel/DATurbulenceModel.C:360-376
        devRhoReff = (-rho nuEff) dev(twoSymm(grad(U)))   (boundary field from the boundary values)
dFdW is obtained by complex step (the reference: reverse-mode AD through calcJacTVecProduct(stateVar -> function),
DASolver.C:1690-1839, then normalizeJacTVecProduct :1443-1553).  PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np

from dafoam_amd.meshgen import NUT_LOWRE_WALL, NUT_SPALDING_WALL, NUT_SYMMETRY

from .residual import BCTable, Ops, bc_scalar, bc_vector, fv1_of, spalding_nut


def _boundary_state(case, g, W):
    """Boundary values the patch functions need: U_b, p_b, T_b, rho_b, the boundary field of devRhoReff."""
    N, F, nIF = g.nC, g.nF, g.nIF
    ops = Ops(g)
    bcell = ops.bc
    rho_solver = case.solver_name in ("DARhoSimpleFoam", "DATurboFoam")
    if rho_solver:
        from .residual_rho import RR, unpack_rho

        U, p, T, nuT, phi = unpack_rho(W, N, F)
    else:
        from .residual import unpack_simple

        U, p, nuT, phi = unpack_simple(W, N, F)
    phi_b = phi[nIF:]
    fields = ("U", "p", "T", "nuTilda", "nut") if rho_solver else ("U", "p", "nuTilda", "nut")
    bt = BCTable(case, g, fields)
    from .residual_rho import mrf_fields  # MRFZone::correctBoundaryVelocity on rotating fixedValue patches

    mrf = mrf_fields(case, g)
    if mrf is not None:
        rot = mrf["incl"] & (bt.code["U"] == 0)
        bt.val["U"] = np.where(rot[:, None], mrf["vFb"], bt.val["U"])
    delta = g.bDeltaCoeffs
    n_b = g.bnf
    Ub, _, _, UgIC, UgBC = bc_vector(bt.code["U"], bt.val["U"], U[bcell], delta, phi_b, n_b)
    pb = bc_scalar(bt.code["p"], bt.val["p"], p[bcell], delta, phi_b)[0]
    nb = bc_scalar(bt.code["nuTilda"], bt.val["nuTilda"], nuT[bcell], delta, phi_b)[0]
    Tb = None
    if rho_solver:
        Tb = bc_scalar(bt.code["T"], bt.val["T"], T[bcell], delta, phi_b)[0]
        R = RR / case.thermo["molWeight"]
        rho, rho_b = p / (R * T), pb / (R * Tb)
        nu, nu_b = case.thermo["mu"] / rho, case.thermo["mu"] / rho_b
    else:
        rho_b = np.ones(g.nBF)
        nu, nu_b = case.nu, case.nu * np.ones(g.nBF)
    nut = nuT * fv1_of(nuT / nu)
    nut_b = nb * fv1_of(nb / nu_b)
    cn = bt.code["nut"]
    nut_b = np.where(cn == NUT_LOWRE_WALL, 0.0 * nut_b, nut_b)
    nut_b = np.where(cn == NUT_SYMMETRY, nut[bcell], nut_b)
    wf = cn == NUT_SPALDING_WALL
    if wf.any():
        dU = U[bcell][wf] - Ub[wf]
        magUp = np.sqrt((dU * dU).sum(1) + 0.0)
        ywf = np.abs(((g.Cf[nIF:][wf] - g.C[bcell][wf]) * n_b[wf]).sum(1))
        tmp = nut_b.astype(W.dtype)
        tmp[wf] = spalding_nut(magUp, magUp * delta[wf], ywf, nu_b[wf] if rho_solver else nu)
        nut_b = tmp
    gradU = ops.grad_vector(U, Ub)
    gUc = gradU[bcell]
    snG = UgIC * U[bcell] + UgBC
    ngU = np.einsum("fk,fkj->fj", n_b, gUc)
    gUb = gUc + n_b[:, :, None] * (snG - ngU)[:, None, :]
    S = gUb + np.swapaxes(gUb, 1, 2)  # twoSymm
    tr = np.trace(S, axis1=1, axis2=2)
    dev = S.copy()
    for d in range(3):
        dev[:, d, d] = dev[:, d, d] - tr / 3.0
    devRhoReff_b = -(rho_b * (nu_b + nut_b))[:, None, None] * dev
    return dict(Ub=Ub, pb=pb, Tb=Tb, rho_b=rho_b, devRhoReff_b=devRhoReff_b)


def _select(g, case, patches):
    sel = np.zeros(g.nBF, bool)
    sl = g.patch_slices()
    for nm in patches:
        sel[sl[nm]] = True
    return sel


def _face_forces(g, b):
    fN = g.bSf * b["pb"][:, None]
    fT = np.einsum("fi,fij->fj", g.bSf, b["devRhoReff_b"])
    return fN + fT


def force(case, g, W, patches, direction, scale=1.0):
    """DAFunctionForce::calcFunction (reference DAFunctionForce.C:79-158), directionMode fixedDirection."""
    b = _boundary_state(case, g, W)
    sel = _select(g, case, patches)
    return scale * ((_face_forces(g, b) @ np.asarray(direction, dtype=float))[sel]).sum()


def moment(case, g, W, patches, axis, center, scale=1.0):
    """DAFunctionMoment::calcFunction (reference DAFunctionMoment.C:73-120): scale * ((Cf - center) x (fN + fT)) . axis."""
    b = _boundary_state(case, g, W)
    sel = _select(g, case, patches)
    r = g.Cf[g.nIF:] - np.asarray(center, dtype=float)
    return scale * ((np.cross(r, _face_forces(g, b)) @ np.asarray(axis, dtype=float))[sel]).sum()


def mass_flow_rate(case, g, W, patches, scale=1.0):
    """DAFunctionMassFlowRate::calcFunction (reference DAFunctionMassFlowRate.C:52-80): sum rho_b (U_b . S_f) * scale."""
    b = _boundary_state(case, g, W)
    sel = _select(g, case, patches)
    return scale * ((b["rho_b"] * (b["Ub"] * g.bSf).sum(1))[sel]).sum()


def total_pressure(case, g, W, patches, scale=1.0):
    """DAFunctionTotalPressure::calcFunction (reference DAFunctionTotalPressure.C:60-90): area average of p + rho |U|^2 / 2."""
    b = _boundary_state(case, g, W)
    sel = _select(g, case, patches)
    a = g.bMagSf[sel]
    val = b["pb"] + 0.5 * b["rho_b"] * (b["Ub"] * b["Ub"]).sum(1)
    return scale * (val[sel] * a).sum() / a.sum()


def total_temperature_ratio(case, g, W, inlet_patches, outlet_patches, gamma=1.4):
    """DAFunctionTotalTemperatureRatio::calcFunction (reference DAFunctionTotalTemperatureRatio.C:60-130): TT_out / TT_in,
    TT = T (1 + (gamma-1)/2 Ma^2), Ma^2 = |U|^2 / (gamma R T), R = Cp - Cp/gamma, area averages per patch set."""
    b = _boundary_state(case, g, W)
    Cp = case.thermo["Cp"]
    R = Cp - Cp / gamma
    U2 = (b["Ub"] * b["Ub"]).sum(1)
    TT = b["Tb"] * (1.0 + 0.5 * (gamma - 1.0) * U2 / (gamma * R * b["Tb"]))
    out = []
    for patches in (inlet_patches, outlet_patches):
        sel = _select(g, case, patches)
        a = g.bMagSf[sel]
        out.append((TT[sel] * a).sum() / a.sum())
    return out[1] / out[0]


def gradient(fun, W, state_scales):
    """s_j dF/dW_j of a callable F(W) for all states by complex step (small meshes only)."""
    n = W.size
    out = np.zeros(n)
    h = 1e-40
    for j in range(n):
        Wp = W.astype(np.complex128)
        Wp[j] += 1j * h * state_scales[j]
        out[j] = fun(Wp).imag / h
    return out


def force_gradient(case, g, W, patches, direction, scale, state_scales):
    """s_j dF/dW_j for all states by complex step (small meshes only)."""
    return gradient(lambda Wp: force(case, g, Wp, patches, direction, scale), W, state_scales)
