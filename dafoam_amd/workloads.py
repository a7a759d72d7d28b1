"""Host-side drivers that prepare the BASELINE workloads on the GPU: what a DAFoam run script does between reading the case and
calling the adjoint - here: converge the DASimpleFoam + SA primal on the NACA0012 O-grid family (BASELINE configs[1] / [2]).

The reference always linearises about a CONVERGED primal (`DAFoamSolver.solve_nonlinear` before `solve_linear`,
dafoam/mphys/mphys_dafoam.py:314-433; its users start fine cases from `mapFields` of coarser ones).  The same sequence with this
library's primal (`PYDAFOAM.solvePrimal` -> das_solve_primal, pseudo-transient Newton-Krylov): solve on a coarse O-grid from a
smooth initial field, prolong, solve on the next finer one, ...; a spanwise extrusion of the finest 2-D solution is (up to the
spanwise part of the momentum diagonal in the Rhie-Chow term, ~1e-5 relative) the solution of the extruded case with symmetry
planes, and a few Newton steps on the extruded mesh remove that rest.  Everything here runs through the public API; no oracle."""
from __future__ import annotations

import time

import numpy as np

from .meshgen import extrude_naca_state, naca0012_case, naca_fluxes_from_velocity, prolong_naca_state

# COLD start (coarsest level, free stream + boundary-layer guess): CFL ramp, pseudo-time term on the transport rows only - with the term on
# the pressure rows the pseudo-time evolution itself is unstable there (round 4, CPU twin with exact Jacobians: blow-up beyond tau ~ 3)
NACA_PRIMAL_AMD = {"primalTauMode": "ramp", "primalTau0": 1.0, "primalTauGrowth": 1.5, "primalSERExponent": 1.0, "primalLinearIters": 1000,
                   "primalLinearTol": 1e-2, "primalPseudoTimeFields": "momentum"}
# PROLONGED start (every finer level): close to the solution - switched evolution relaxation on the initial residual with the term on all
# rows (the round-2 scheme of the channel workloads); measured on the 400 x 125 level: 28 steps / 15 s, where the ramp variants stall
NACA_PRIMAL_AMD_FINE = {"primalTauMode": "ser", "primalTau0": 1.0, "primalSERExponent": 1.5, "primalLinearIters": 1000, "primalLinearTol": 1e-3,
                        "primalPseudoTimeFields": "all"}


def naca_levels(n_around, n_normal, coarsest=100):
    """Grid-sequencing levels, coarse to fine: (n_around, n_normal) halved while at least `coarsest` cells stay around the airfoil."""
    lv = [(int(n_around), int(n_normal))]
    while lv[-1][0] // 2 >= coarsest and lv[-1][1] // 2 >= 8:
        lv.append((lv[-1][0] // 2, lv[-1][1] // 2))
    return lv[::-1]


def naca_converged_primal(n_around=800, n_normal=250, options=None, first_cell=2.0e-5, rel_tol=1e-8, max_steps=120, coarsest=100, verbose=False,
                          case_kwargs=None):
    """Converged DASimpleFoam + SA state on the one-layer NACA0012 O-grid of n_around x n_normal cells.  Returns (case, info):
    the FoamCase with `states` = the converged primal, info = list of per-level dicts (dims, steps, linearIterations, res0, res,
    seconds, fail).  `options`: PYDAFOAM options (amd.primal* entries are overridden by the cold-start settings)."""
    from .pyDAFoam import PYDAFOAM

    levels = naca_levels(n_around, n_normal, coarsest)
    fcs = [first_cell * 2 ** (len(levels) - 1 - i) for i in range(len(levels))]   # same growth ratio on every level
    ckw = dict(case_kwargs or {})
    ckw.setdefault("perturb", 0.0)
    W_prev, info, case = None, [], None
    for li, ((nx, ny), fc) in enumerate(zip(levels, fcs)):
        t0 = time.time()
        case = naca0012_case(nx, ny, 1, first_cell=fc, **ckw)
        if W_prev is not None:
            case.states = prolong_naca_state(levels[li - 1], W_prev, case, (nx, ny), first_cell=fc, coarse_first_cell=fcs[li - 1], fold_seam=bool(ckw.get("fold_seam", False)))
        opts = dict(options or {})
        opts["amd"] = dict(opts.get("amd", {}), **(NACA_PRIMAL_AMD if W_prev is None else NACA_PRIMAL_AMD_FINE))
        D = PYDAFOAM(options=opts, case=case)
        fail, inf = D.solver.solvePrimal(maxSteps=max_steps, relTol=rel_tol, absTol=0.0)
        W_prev = D.getStates().copy()
        case.states = W_prev
        rec = dict(dims=(nx, ny), first_cell=fc, steps=inf["steps"], linearIterations=inf["linearIterations"], res0=inf["res0"], res=inf["res"],
                   fail=int(fail), seconds=time.time() - t0)
        info.append(rec)
        if verbose:
            print(f"[naca primal] level {nx} x {ny}: {rec['steps']} Newton steps, {rec['linearIterations']} GMRES iterations, |R| {rec['res0']:.3e} -> {rec['res']:.3e}, "
                  f"{rec['seconds']:.1f} s, fail {rec['fail']}", flush=True)
        del D
    return case, info


def naca_extruded_case(case2d, dims2d, nz, dz=0.1, first_cell=2.0e-5, options=None, polish_steps=3, polish_tol=1e-3, verbose=False, case_kwargs=None):
    """The one-layer solution extruded to nz spanwise layers of thickness dz (symmetry planes front / back), polished by a few
    Newton steps on the extruded mesh.  Returns (case3d, info)."""
    from .pyDAFoam import PYDAFOAM

    nx, ny = dims2d
    ckw = dict(case_kwargs or {})
    ckw.setdefault("perturb", 0.0)
    t0 = time.time()
    case3 = naca0012_case(nx, ny, nz, span=dz * nz, first_cell=first_cell, y_wall_section=case2d.y_wall, **ckw)
    case3.states = extrude_naca_state(case2d, case2d.states, case3, (nx, ny, nz))
    if ckw.get("sweep_deg", 0.0) or ckw.get("taper", 0.0):
        # swept / tapered segment: the layers are no longer copies of the section - cell fields start from the section's, the face fluxes
        # from the interpolated velocities on the real faces; the Newton steps below then have real work to do
        N3 = case3.mesh.n_cells
        W3 = np.asarray(case3.states).copy()
        W3[5 * N3 :] = naca_fluxes_from_velocity(case3, W3[: 3 * N3].reshape(N3, 3))
        case3.states = W3
    info = dict(dims=(nx, ny, nz), steps=0, seconds=0.0)
    if polish_steps > 0:
        opts = dict(options or {})
        # a start next to the solution: large pseudo-time step at once
        # (measured at 2 M cells: ramp / transport-rows mode from tau 1e3: |R| 0.78 -> 1.4e-3 in 2 steps; the ser / all-rows mode stalls here)
        opts["amd"] = dict(opts.get("amd", {}), **dict(NACA_PRIMAL_AMD, primalTau0=1.0e3, primalTauGrowth=10.0))
        D = PYDAFOAM(options=opts, case=case3)
        fail, inf = D.solver.solvePrimal(maxSteps=polish_steps, relTol=polish_tol, absTol=0.0)
        case3.states = D.getStates().copy()
        info.update(steps=inf["steps"], linearIterations=inf["linearIterations"], res0=inf["res0"], res=inf["res"], fail=int(fail))
        del D
    info["seconds"] = time.time() - t0
    if verbose:
        print(f"[naca primal] extruded {nx} x {ny} x {nz}: {info}", flush=True)
    return case3, info


# compressible bump channel (BASELINE configs[3] family), round 6: cold start by a CFL ramp from a small pseudo-time step with the
# preconditioner rebuilt every step (switched evolution relaxation diverges from the smooth synthetic state, the NACA cold-start settings
# stall: profiles/r07s_*, r07t_*, r08d_*), finer levels from the prolonged coarse solution with switched evolution relaxation
RHO_PRIMAL_AMD_COLD = {"primalTauMode": "ramp", "primalTau0": 0.1, "primalTauGrowth": 1.3, "primalPCLag": 1, "primalPseudoTimeFields": "all"}
RHO_PRIMAL_AMD_FINE = {"primalTauMode": "ser", "primalTau0": 1.0, "primalSERExponent": 1.5, "primalPCLag": 1, "primalPseudoTimeFields": "all"}


def rho_channel_converged_primal(nx, ny, nz, options=None, levels=2, rel_tol=1e-8, max_steps=100, verbose=False, case_kwargs=None):
    """Converged DARhoSimpleFoam + SA state on the nx x ny x nz bump channel by grid sequencing: the coarsest level (every dimension halved
    `levels - 1` times) from the smooth synthetic state with the cold-start settings, every finer level from the prolonged solution.
    Returns (case, info) like naca_converged_primal."""
    from .meshgen import prolong_rho_channel_state, rho_channel_case
    from .pyDAFoam import PYDAFOAM

    ckw = dict(case_kwargs or {})
    dims = [(max(4, nx >> (levels - 1 - l)), max(4, ny >> (levels - 1 - l)), max(4, nz >> (levels - 1 - l))) for l in range(levels)]
    prev, info, case = None, [], None
    for li, d in enumerate(dims):
        t0 = time.time()
        case = rho_channel_case(*d, **ckw)
        if prev is not None:
            prolong_rho_channel_state(case, d, prev)
        opts = dict(options or {})
        opts["amd"] = dict(opts.get("amd", {}), **(RHO_PRIMAL_AMD_COLD if prev is None else RHO_PRIMAL_AMD_FINE))
        D = PYDAFOAM(options=opts, case=case)
        fail, inf = D.solver.solvePrimal(maxSteps=max_steps, relTol=rel_tol, absTol=0.0)
        W = D.getStates().copy()
        case.states = W
        prev = {"dims": d, "W": W}
        rec = dict(dims=d, steps=inf["steps"], linearIterations=inf["linearIterations"], res0=inf["res0"], res=inf["res"], fail=int(fail), seconds=time.time() - t0)
        info.append(rec)
        if verbose:
            print(f"[rho primal] level {d}: {rec['steps']} Newton steps, {rec['linearIterations']} GMRES iterations, |R| {rec['res0']:.3e} -> {rec['res']:.3e}, "
                  f"{rec['seconds']:.1f} s, fail {rec['fail']}", flush=True)
        del D
    return case, info
