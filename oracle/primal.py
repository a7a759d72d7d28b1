"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's steady primal loop for DASimpleFoam + Spalart-Allmaras, used ONLY to produce
converged states W* about which the adjoint is linearised (the primal is upstream of the hot path, SURVEY.md
section 3.4/8f):

  * SIMPLE loop                      reference src/adjoint/DASolver/DASimpleFoam/DASimpleFoam.C:123-185,
                                     UEqnSimple.H, pEqnSimple.H
  * SA transport solve + correctNut  reference src/adjoint/DAModel/DATurbulenceModel/DASpalartAllmaras.C:386-405,407-488

It is built from the same operator restatements as oracle/residual.py, so its fixed point satisfies
R(W*) = 0 for the residual definitions of DAResidualSimpleFoam.C:106-237 (checked in tests).
PARITY UNPINNED (no reference run possible here).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from dafoam_amd.meshgen import BC_FIXED_VALUE, BC_SYMMETRY, NUT_LOWRE_WALL, NUT_SPALDING_WALL, NUT_SYMMETRY

from .residual import (
    SA, SMALL, VSMALL, BCTable, Ops, bc_scalar, bc_vector, dev2T, fv1_of, relax_diag, residual, sadd, spalding_nut, unpack_simple,
)


def _csr(N, oi, ni, diag, upper, lower):
    """ldu -> csr: row owner gets `upper` at column neighbour, row neighbour gets `lower` at column owner."""
    rows = np.concatenate([np.arange(N), oi, ni])
    cols = np.concatenate([np.arange(N), ni, oi])
    vals = np.concatenate([diag, upper, lower])
    return sp.csr_matrix((vals, (rows, cols)), shape=(N, N))


def simple_iteration(case, g, W, alpha_p=0.3, use_constrain_hbya=True):
    N, F, nIF = g.nC, g.nF, g.nIF
    ops = Ops(g)
    oi, ni, bcell = ops.oi, ops.ni, ops.bc
    U, p, nuT, phi = [x.copy() for x in unpack_simple(W, N, F)]
    nu = case.nu
    phi_i, phi_b = phi[:nIF], phi[nIF:]
    bt = BCTable(case, g, ("U", "p", "nuTilda", "nut"))
    delta = g.bDeltaCoeffs
    V = g.V
    n_b = g.bnf

    def bcs():
        Ub_ = bc_vector(bt.code["U"], bt.val["U"], U[bcell], delta, phi_b, n_b)
        pb_ = bc_scalar(bt.code["p"], bt.val["p"], p[bcell], delta, phi_b)
        nb_ = bc_scalar(bt.code["nuTilda"], bt.val["nuTilda"], nuT[bcell], delta, phi_b)
        return Ub_, pb_, nb_

    def nut_fields(Ub, nb):
        nut = nuT * fv1_of(nuT / nu)
        nut_b = nb * fv1_of(nb / nu)
        cn = bt.code["nut"]
        nut_b = np.where(cn == NUT_LOWRE_WALL, 0.0, nut_b)
        nut_b = np.where(cn == NUT_SYMMETRY, nut[bcell], nut_b)
        wf = cn == NUT_SPALDING_WALL
        if wf.any():
            dU = U[bcell][wf] - Ub[wf]
            magUp = np.sqrt((dU * dU).sum(1))
            ywf = np.abs(((g.Cf[nIF:][wf] - g.C[bcell][wf]) * n_b[wf]).sum(1))
            nut_b = nut_b.copy()
            nut_b[wf] = spalding_nut(magUp, magUp * delta[wf], ywf, nu)
        return nut, nut_b

    (Ub, UvIC, UvBC, UgIC, UgBC), (pb, pvIC, pvBC, pgIC, pgBC), (nb, nvIC, nvBC, ngIC, ngBC) = bcs()
    nut, nut_b = nut_fields(Ub, nb)
    nuEff, nuEff_b = nu + nut, nu + nut_b
    gradU = ops.grad_vector(U, Ub)
    gradP = ops.grad_scalar(p, pb)
    snGradU_b = UgIC * U[bcell] + UgBC
    gUc = gradU[bcell]
    ngU = np.einsum("fk,fkj->fj", n_b, gUc)
    gradU_b = gUc + n_b[:, :, None] * (snGradU_b - ngU)[:, None, :]

    # ---------------- UEqn (identical coefficients to oracle/residual.py)
    wu = (phi_i >= 0).astype(float)
    lower = -wu * phi_i
    upper = lower + phi_i
    diag = sadd(oi, -lower, N) + sadd(ni, -upper, N)
    sumPhi = ops.surface_sum(phi_i, phi_b)
    diag = diag - sumPhi
    iC = phi_b[:, None] * UvIC
    bC = -phi_b[:, None] * UvBC
    pos = phi_i > 0
    d_o, d_n = g.Cf[:nIF] - g.C[oi], g.Cf[:nIF] - g.C[ni]
    c_o = np.einsum("fi,fij->fj", d_o, gradU[oi])
    c_n = np.einsum("fi,fij->fj", d_n, gradU[ni])
    wl = g.w[:, None]
    corr = np.where(pos[:, None], c_o, c_n)
    mx = np.where(pos[:, None], (1.0 - wl) * (U[ni] - U[oi]), wl * (U[oi] - U[ni]))
    sfc, mxc = (corr * corr).sum(1), (corr * mx).sum(1)
    scale = np.where(sfc > 0, np.where(mxc < 0, 0.0, np.where(sfc > mxc, mxc / (sfc + VSMALL), 1.0)), 1.0)
    fcorr = phi_i[:, None] * corr * scale[:, None]
    src = -(sadd(oi, fcorr, N) - sadd(ni, fcorr, N))
    gam = ops.interp(nuEff) * g.magSf[:nIF]
    gam_b = nuEff_b * g.bMagSf
    cdiff = gam * g.nonOrthDeltaCoeffs
    upper, lower = upper - cdiff, lower - cdiff
    diag = diag + sadd(oi, cdiff, N) + sadd(ni, cdiff, N)
    fcorrL = gam[:, None] * np.einsum("fi,fij->fj", g.nonOrthCorr, ops.interp(gradU))
    src = src + (sadd(oi, fcorrL, N) - sadd(ni, fcorrL, N))
    iC = iC - gam_b[:, None] * UgIC
    bC = bC + gam_b[:, None] * UgBC
    tau = nuEff[:, None, None] * dev2T(gradU)
    tau_b = nuEff_b[:, None, None] * dev2T(gradU_b)
    src = src + ops.surface_sum(np.einsum("fi,fij->fj", g.Sf[:nIF], ops.interp(tau)), np.einsum("fi,fij->fj", g.bSf, tau_b))
    D0 = diag
    sumOff = sadd(oi, np.abs(upper), N) + sadd(ni, np.abs(lower), N)
    D = relax_diag(D0, sumOff, iC, bcell, case.relax["U"], N)
    src = src + (D - D0)[:, None] * U
    bdiag = sadd(bcell, iC, N)
    bsrc = sadd(bcell, bC, N)
    # solve(UEqn == -grad(p)) component-wise
    Unew = np.empty_like(U)
    for k in range(3):
        M = _csr(N, oi, ni, D + bdiag[:, k], upper, lower)
        Unew[:, k] = spla.spsolve(M.tocsc(), src[:, k] + bsrc[:, k] - V * gradP[:, k])
    U = Unew
    # ---------------- pEqn
    offU = sadd(oi, upper[:, None] * U[ni], N) + sadd(ni, lower[:, None] * U[oi], N)
    avgb = bdiag.sum(1) / 3.0
    A = (D + avgb) / V
    H = ((avgb[:, None] - bdiag) * U - offU + src + bsrc) / V[:, None]
    rAU = 1.0 / A
    HbyA = rAU[:, None] * H
    (Ub, UvIC, UvBC, UgIC, UgBC), _, _ = bcs()
    cU = bt.code["U"]
    HbyA_b = HbyA[bcell].copy()
    symU = cU == BC_SYMMETRY
    if symU.any():
        hn = (HbyA_b[symU] * n_b[symU]).sum(1)[:, None]
        HbyA_b[symU] = HbyA_b[symU] - n_b[symU] * hn
    if use_constrain_hbya:
        fx = cU == BC_FIXED_VALUE
        HbyA_b[fx] = Ub[fx]
    phiHbyA_i = (ops.interp(HbyA) * g.Sf[:nIF]).sum(1)
    phiHbyA_b = (HbyA_b * g.bSf).sum(1)
    gp = ops.interp(rAU) * g.magSf[:nIF]
    gp_b = rAU[bcell] * g.bMagSf
    cp = gp * g.nonOrthDeltaCoeffs
    # laplacian(rAU,p) - div(phiHbyA) = 0 ; explicit non-orthogonal correction with the current grad(p)
    for _ in range(2):  # nNonOrthogonalCorrectors 1
        corr_f = gp * (g.nonOrthCorr * ops.interp(gradP)).sum(1)
        dp = -(sadd(oi, cp, N) + sadd(ni, cp, N)) + sadd(bcell, gp_b * pgIC, N)
        Mp = _csr(N, oi, ni, dp, cp, cp)
        rhs = ops.surface_sum(phiHbyA_i, phiHbyA_b) - (sadd(oi, corr_f, N) - sadd(ni, corr_f, N)) - sadd(bcell, gp_b * pgBC, N)
        p_new = spla.spsolve(Mp.tocsc(), rhs)
        pb_new = pvIC * p_new[bcell] + pvBC
        gradP = ops.grad_scalar(p_new, pb_new)
    flux_i = cp * (p_new[ni] - p_new[oi]) + gp * (g.nonOrthCorr * ops.interp(gradP)).sum(1)
    flux_b = gp_b * (pgIC * p_new[bcell] + pgBC)
    phi = np.concatenate([phiHbyA_i - flux_i, phiHbyA_b - flux_b])
    p = p + alpha_p * (p_new - p)
    pb = pvIC * p[bcell] + pvBC
    U = HbyA - rAU[:, None] * ops.grad_scalar(p, pb)
    phi_i, phi_b = phi[:nIF], phi[nIF:]
    # ---------------- SA transport
    (Ub, *_), _, (nb, nvIC, nvBC, ngIC, ngBC) = bcs()
    gradU = ops.grad_vector(U, Ub)
    gradN = ops.grad_scalar(nuT, nb)
    y = case.y_wall
    k2y2 = (SA["kappa"] * y) ** 2
    chi = nuT / nu
    fv1 = fv1_of(chi)
    skew = 0.5 * (gradU - np.swapaxes(gradU, 1, 2))
    Omega = np.sqrt(2.0) * np.sqrt((skew * skew).sum((1, 2)))
    fv2 = 1.0 - chi / (1.0 + chi * fv1)
    Stilda = np.maximum(Omega + fv2 * nuT / k2y2, SA["Cs"] * Omega)
    r = np.minimum(nuT / (np.maximum(Stilda, SMALL) * k2y2), 10.0)
    gg = r + SA["Cw2"] * (r**6 - r)
    fw = gg * ((1.0 + SA["Cw3"] ** 6) / (gg**6 + SA["Cw3"] ** 6)) ** (1.0 / 6.0)
    wu = (phi_i >= 0).astype(float)
    lo = -wu * phi_i
    up = lo + phi_i
    dN = sadd(oi, -lo, N) + sadd(ni, -up, N) - ops.surface_sum(phi_i, phi_b)
    Dn = (nuT + nu) / SA["sigmaNut"]
    gn = ops.interp(Dn) * g.magSf[:nIF]
    gn_b = (nb + nu) / SA["sigmaNut"] * g.bMagSf
    cd = gn * g.nonOrthDeltaCoeffs
    up, lo = up - cd, lo - cd
    dN = dN + sadd(oi, cd, N) + sadd(ni, cd, N)
    fcn = gn * (g.nonOrthCorr * ops.interp(gradN)).sum(1)
    sN = sadd(oi, fcn, N) - sadd(ni, fcn, N)
    iCn = phi_b * nvIC - gn_b * ngIC
    bCn = -phi_b * nvBC + gn_b * ngBC
    dN = dN + V * SA["Cw1"] * fw * nuT / (y * y)  # fvm::Sp
    sN = sN + V * (SA["Cb2"] / SA["sigmaNut"] * (gradN * gradN).sum(1) + SA["Cb1"] * Stilda * nuT)
    D0n = dN
    sumOffN = sadd(oi, np.abs(up), N) + sadd(ni, np.abs(lo), N)
    Dn_rel = relax_diag(D0n, sumOffN, iCn, bcell, case.relax["nuTilda"], N)
    sN = sN + (Dn_rel - D0n) * nuT
    Mn = _csr(N, oi, ni, Dn_rel + sadd(bcell, iCn, N), up, lo)
    nuT = spla.spsolve(Mn.tocsc(), sN + sadd(bcell, bCn, N))
    nuT = np.maximum(nuT, 1e-16)  # DAUtility::boundVar
    return np.concatenate([U.ravel(), p, nuT, phi])


def solve_primal(case, g, W0=None, max_iters=2000, tol=1e-9, verbose=False):
    """SIMPLE iterations until the (normalised, per-block RMS) residual of oracle.residual drops below tol
    relative to the first iterate.  Returns (W, history)."""
    W = case.states.copy() if W0 is None else W0.copy()
    N = g.nC
    hist = []
    rho_solver = case.solver_name == "DARhoSimpleFoam"
    step = simple_iteration_rho if rho_solver else simple_iteration
    nsc = 3 if rho_solver else 2  # scalar cell blocks after U

    def norms(W):
        R = residual(case, g, W)
        out = [np.linalg.norm(R[: 3 * N]) / np.sqrt(3 * N)]
        for k in range(nsc):
            out.append(np.linalg.norm(R[(3 + k) * N : (4 + k) * N]) / np.sqrt(N))
        out.append(np.linalg.norm(R[(3 + nsc) * N :]) / np.sqrt(g.nF))
        return np.array(out)

    r0 = None
    for it in range(max_iters):
        W = step(case, g, W)
        if it % 10 == 0 or it == max_iters - 1:
            r = norms(W)
            if r0 is None:
                r0 = np.maximum(r, 1e-300)
            hist.append(r)
            if verbose:
                print(it, " ".join(f"{x:.3e}" for x in r), flush=True)
            if np.all(r / r0 < tol) or not np.all(np.isfinite(r)):
                break
    return W, np.array(hist)


# ============================================================================================ DARhoSimpleFoam
def simple_iteration_rho(case, g, W, alpha_p=0.3, use_constrain_hbya=True):
    """One compressible SIMPLE iteration (reference src/adjoint/DASolver/DARhoSimpleFoam/DARhoSimpleFoam.C solvePrimal:
    UEqn -> EEqn -> thermo update -> pEqn -> SA), built from the operators of oracle/residual_rho.py."""
    from .residual_rho import RR, TREF, unpack_rho

    N, F, nIF = g.nC, g.nF, g.nIF
    ops = Ops(g)
    oi, ni, bcell = ops.oi, ops.ni, ops.bc
    U, p, T, nuT, phi = [x.copy() for x in unpack_rho(W, N, F)]
    th = case.thermo
    Cp, mu, Pr, Prt = th["Cp"], th["mu"], th["Pr"], th["Prt"]
    R = RR / th["molWeight"]
    bt = BCTable(case, g, ("U", "p", "T", "nuTilda", "nut"))
    delta = g.bDeltaCoeffs
    V = g.V
    n_b = g.bnf
    phi_i, phi_b = phi[:nIF], phi[nIF:]

    def fields():
        Ubc = bc_vector(bt.code["U"], bt.val["U"], U[bcell], delta, phi_b, n_b)
        pbc = bc_scalar(bt.code["p"], bt.val["p"], p[bcell], delta, phi_b)
        Tbc = bc_scalar(bt.code["T"], bt.val["T"], T[bcell], delta, phi_b)
        nbc = bc_scalar(bt.code["nuTilda"], bt.val["nuTilda"], nuT[bcell], delta, phi_b)
        rho, rho_b = p / (R * T), pbc[0] / (R * Tbc[0])
        nu, nu_b = mu / rho, mu / rho_b
        nut = nuT * fv1_of(nuT / nu)
        nut_b = nbc[0] * fv1_of(nbc[0] / nu_b)
        cn = bt.code["nut"]
        nut_b = np.where(cn == NUT_LOWRE_WALL, 0.0, nut_b)
        nut_b = np.where(cn == NUT_SYMMETRY, nut[bcell], nut_b)
        wf = cn == NUT_SPALDING_WALL
        if wf.any():
            dU = U[bcell][wf] - Ubc[0][wf]
            magUp = np.sqrt((dU * dU).sum(1))
            ywf = np.abs(((g.Cf[nIF:][wf] - g.C[bcell][wf]) * n_b[wf]).sum(1))
            nut_b = nut_b.copy()
            nut_b[wf] = spalding_nut(magUp, magUp * delta[wf], ywf, nu_b[wf])
        return Ubc, pbc, Tbc, nbc, rho, rho_b, nu, nu_b, nut, nut_b

    (Ub, UvIC, UvBC, UgIC, UgBC), (pb, pvIC, pvBC, pgIC, pgBC), (Tb, *_), (nb, nvIC, nvBC, ngIC, ngBC), rho, rho_b, nu, nu_b, nut, nut_b = fields()
    muEff, muEff_b = rho * (nu + nut), rho_b * (nu_b + nut_b)
    gradU = ops.grad_vector(U, Ub)
    gradP = ops.grad_scalar(p, pb)
    snGradU_b = UgIC * U[bcell] + UgBC
    gUc = gradU[bcell]
    ngU = np.einsum("fk,fkj->fj", n_b, gUc)
    gradU_b = gUc + n_b[:, :, None] * (snGradU_b - ngU)[:, None, :]
    # ---------------- UEqn
    wu = (phi_i >= 0).astype(float)
    lower = -wu * phi_i
    upper = lower + phi_i
    sumPhi = ops.surface_sum(phi_i, phi_b)
    diag = sadd(oi, -lower, N) + sadd(ni, -upper, N) - sumPhi
    iC = phi_b[:, None] * UvIC
    bC = -phi_b[:, None] * UvBC
    pos = phi_i > 0
    c_o = np.einsum("fi,fij->fj", g.Cf[:nIF] - g.C[oi], gradU[oi])
    c_n = np.einsum("fi,fij->fj", g.Cf[:nIF] - g.C[ni], gradU[ni])
    wl = g.w[:, None]
    corr = np.where(pos[:, None], c_o, c_n)
    mx = np.where(pos[:, None], (1.0 - wl) * (U[ni] - U[oi]), wl * (U[oi] - U[ni]))
    sfc, mxc = (corr * corr).sum(1), (corr * mx).sum(1)
    scale = np.where(sfc > 0, np.where(mxc < 0, 0.0, np.where(sfc > mxc, mxc / (sfc + VSMALL), 1.0)), 1.0)
    fcorr = phi_i[:, None] * corr * scale[:, None]
    src = -(sadd(oi, fcorr, N) - sadd(ni, fcorr, N))
    gam = ops.interp(muEff) * g.magSf[:nIF]
    gam_b = muEff_b * g.bMagSf
    cdiff = gam * g.nonOrthDeltaCoeffs
    upper, lower = upper - cdiff, lower - cdiff
    diag = diag + sadd(oi, cdiff, N) + sadd(ni, cdiff, N)
    fcorrL = gam[:, None] * np.einsum("fi,fij->fj", g.nonOrthCorr, ops.interp(gradU))
    src = src + (sadd(oi, fcorrL, N) - sadd(ni, fcorrL, N))
    iC = iC - gam_b[:, None] * UgIC
    bC = bC + gam_b[:, None] * UgBC
    tau = muEff[:, None, None] * dev2T(gradU)
    tau_b = muEff_b[:, None, None] * dev2T(gradU_b)
    src = src + ops.surface_sum(np.einsum("fi,fij->fj", g.Sf[:nIF], ops.interp(tau)), np.einsum("fi,fij->fj", g.bSf, tau_b))
    D0 = diag
    sumOff = sadd(oi, np.abs(upper), N) + sadd(ni, np.abs(lower), N)
    D = relax_diag(D0, sumOff, iC, bcell, case.relax["U"], N)
    src = src + (D - D0)[:, None] * U
    bdiag = sadd(bcell, iC, N)
    bsrc = sadd(bcell, bC, N)
    Unew = np.empty_like(U)
    for k in range(3):
        Unew[:, k] = spla.spsolve(_csr(N, oi, ni, D + bdiag[:, k], upper, lower).tocsc(), src[:, k] + bsrc[:, k] - V * gradP[:, k])
    U = Unew
    # ---------------- EEqn (for he), then T and thermo update
    (Ub, *_), _, (Tb, *_), _, rho, rho_b, nu, nu_b, nut, nut_b = fields()
    he = Cp * (T - TREF)
    heb, hvIC, hvBC, hgIC, hgBC = bc_scalar(bt.code["T"], Cp * (bt.val["T"] - TREF), he[bcell], delta, phi_b)
    alphaEff, alphaEff_b = mu / Pr + rho * nut / Prt, mu / Pr + rho_b * nut_b / Prt
    gradHe = ops.grad_scalar(he, heb)
    loE = -wu * phi_i
    upE = loE + phi_i
    dE = sadd(oi, -loE, N) + sadd(ni, -upE, N) - sumPhi
    ga = ops.interp(alphaEff) * g.magSf[:nIF]
    ga_b = alphaEff_b * g.bMagSf
    cde = ga * g.nonOrthDeltaCoeffs
    upE, loE = upE - cde, loE - cde
    dE = dE + sadd(oi, cde, N) + sadd(ni, cde, N)
    fce = ga * (g.nonOrthCorr * ops.interp(gradHe)).sum(1)
    sE = sadd(oi, fce, N) - sadd(ni, fce, N)
    iCe = phi_b * hvIC - ga_b * hgIC
    bCe = -phi_b * hvBC + ga_b * hgBC
    K, Kb = 0.5 * (U * U).sum(1), 0.5 * (Ub * Ub).sum(1)
    Kf = np.where(phi_i >= 0, K[oi], K[ni])
    sE = sE - ops.surface_sum(phi_i * Kf, phi_b * Kb)
    sumOffE = sadd(oi, np.abs(upE), N) + sadd(ni, np.abs(loE), N)
    DE = relax_diag(dE, sumOffE, iCe, bcell, case.relax.get("T", 0.9), N)
    sE = sE + (DE - dE) * he
    he = spla.spsolve(_csr(N, oi, ni, DE + sadd(bcell, iCe, N), upE, loE).tocsc(), sE + sadd(bcell, bCe, N))
    T = he / Cp + TREF
    # ---------------- pEqn
    (Ub, *_), (pb, pvIC, pvBC, pgIC, pgBC), (Tb, *_), _, rho, rho_b, nu, nu_b, nut, nut_b = fields()
    offU = sadd(oi, upper[:, None] * U[ni], N) + sadd(ni, lower[:, None] * U[oi], N)
    avgb = bdiag.sum(1) / 3.0
    A = (D + avgb) / V
    H = ((avgb[:, None] - bdiag) * U - offU + src + bsrc) / V[:, None]
    rAU = 1.0 / A
    HbyA = rAU[:, None] * H
    cU = bt.code["U"]
    HbyA_b = HbyA[bcell].copy()
    symU = cU == BC_SYMMETRY
    if symU.any():
        hn = (HbyA_b[symU] * n_b[symU]).sum(1)[:, None]
        HbyA_b[symU] = HbyA_b[symU] - n_b[symU] * hn
    if use_constrain_hbya:
        fx = cU == BC_FIXED_VALUE
        HbyA_b[fx] = Ub[fx]
    phiHbyA_i = ops.interp(rho) * (ops.interp(HbyA) * g.Sf[:nIF]).sum(1)
    phiHbyA_b = rho_b * (HbyA_b * g.bSf).sum(1)
    gp = ops.interp(rho * rAU) * g.magSf[:nIF]
    gp_b = rho_b * rAU[bcell] * g.bMagSf
    cp = gp * g.nonOrthDeltaCoeffs
    for _ in range(2):
        corr_f = gp * (g.nonOrthCorr * ops.interp(gradP)).sum(1)
        dp = -(sadd(oi, cp, N) + sadd(ni, cp, N)) + sadd(bcell, gp_b * pgIC, N)
        rhs = ops.surface_sum(phiHbyA_i, phiHbyA_b) - (sadd(oi, corr_f, N) - sadd(ni, corr_f, N)) - sadd(bcell, gp_b * pgBC, N)
        p_new = spla.spsolve(_csr(N, oi, ni, dp, cp, cp).tocsc(), rhs)
        gradP = ops.grad_scalar(p_new, pvIC * p_new[bcell] + pvBC)
    flux_i = cp * (p_new[ni] - p_new[oi]) + gp * (g.nonOrthCorr * ops.interp(gradP)).sum(1)
    flux_b = gp_b * (pgIC * p_new[bcell] + pgBC)
    phi = np.concatenate([phiHbyA_i - flux_i, phiHbyA_b - flux_b])
    p = p + alpha_p * (p_new - p)
    pb = pvIC * p[bcell] + pvBC
    U = HbyA - rAU[:, None] * ops.grad_scalar(p, pb)
    phi_i, phi_b = phi[:nIF], phi[nIF:]
    # ---------------- SA (compressible)
    (Ub, *_), _, _, (nb, nvIC, nvBC, ngIC, ngBC), rho, rho_b, nu, nu_b, nut, nut_b = fields()
    gradU = ops.grad_vector(U, Ub)
    gradN = ops.grad_scalar(nuT, nb)
    y = case.y_wall
    k2y2 = (SA["kappa"] * y) ** 2
    chi = nuT / nu
    fv1 = fv1_of(chi)
    skew = 0.5 * (gradU - np.swapaxes(gradU, 1, 2))
    Omega = np.sqrt(2.0) * np.sqrt((skew * skew).sum((1, 2)))
    fv2 = 1.0 - chi / (1.0 + chi * fv1)
    Stilda = np.maximum(Omega + fv2 * nuT / k2y2, SA["Cs"] * Omega)
    r = np.minimum(nuT / (np.maximum(Stilda, SMALL) * k2y2), 10.0)
    gg = r + SA["Cw2"] * (r**6 - r)
    fw = gg * ((1.0 + SA["Cw3"] ** 6) / (gg**6 + SA["Cw3"] ** 6)) ** (1.0 / 6.0)
    wu = (phi_i >= 0).astype(float)
    lo = -wu * phi_i
    up = lo + phi_i
    dN = sadd(oi, -lo, N) + sadd(ni, -up, N) - ops.surface_sum(phi_i, phi_b)
    gn = ops.interp(rho * (nuT + nu) / SA["sigmaNut"]) * g.magSf[:nIF]
    gn_b = rho_b * (nb + nu_b) / SA["sigmaNut"] * g.bMagSf
    cd = gn * g.nonOrthDeltaCoeffs
    up, lo = up - cd, lo - cd
    dN = dN + sadd(oi, cd, N) + sadd(ni, cd, N)
    fcn = gn * (g.nonOrthCorr * ops.interp(gradN)).sum(1)
    sN = sadd(oi, fcn, N) - sadd(ni, fcn, N)
    iCn = phi_b * nvIC - gn_b * ngIC
    bCn = -phi_b * nvBC + gn_b * ngBC
    dN = dN + V * SA["Cw1"] * rho * fw * nuT / (y * y)
    sN = sN + V * rho * (SA["Cb2"] / SA["sigmaNut"] * (gradN * gradN).sum(1) + SA["Cb1"] * Stilda * nuT)
    sumOffN = sadd(oi, np.abs(up), N) + sadd(ni, np.abs(lo), N)
    Dn_rel = relax_diag(dN, sumOffN, iCn, bcell, case.relax["nuTilda"], N)
    sN = sN + (Dn_rel - dN) * nuT
    nuT = spla.spsolve(_csr(N, oi, ni, Dn_rel + sadd(bcell, iCn, N), up, lo).tocsc(), sN + sadd(bcell, bCn, N))
    nuT = np.maximum(nuT, 1e-16)
    return np.concatenate([U.ravel(), p, T, nuT, phi])
