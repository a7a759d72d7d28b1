"""ORACLE (test infrastructure only - never imported by the product path).

The CPU side of the psi parity legs (bench.py psi_parity_200k, tests/test_gpu_naca.py): an adjoint solve that is independent of the
GPU path in everything that determines psi - the matrices dRdW^T and dRdWTPC are ASSEMBLED ON THE HOST by oracle/adjoint_host.py
(own connectivity from the reference's stencil tables, own colouring, own residual evaluation with dual numbers / the reference's
finite differences) and the system is solved by the oracle's all-core GMRES (oracle/csrc/oracle_krylov_omp.c).  The preconditioner
only steers the convergence, psi does not depend on it: its node structure and the coarse aggregates are integer tables taken from
the library (KSP.pcStructure / KSP.coarse), its values are factorised on the host from the host-assembled dRdWTPC.
Reference roles: DASolver::calcdRdWT (DASolver.C:948-1089), DAPartDeriv::calcPartDerivMat (DAPartDeriv.C:350-473),
DALinearEqn::solveLinearEqn (DALinearEqn.C:341-437)."""
from __future__ import annotations

import time

import numpy as np


def host_adjoint_solve(case, norm_states, rhs, pc_structure, coarse, threads, rel_tol=1e-10, pc_blend=0.0, restart=1500, max_iters=3000,
                       max_seconds=240.0, stage=None):
    """Returns (psi, info): info has the timings of the host pipeline (geometry, connectivity + colouring, dRdWTPC by coloured finite
    differences, dRdW^T by coloured dual numbers, factorisation, GMRES) and the solver's iteration data.  `coarse` = (nagg, agg)."""
    from . import jacobian as J
    from . import linear as OL
    from .adjoint_host import HostAdjoint
    from .foam_mesh import Geometry

    def mark(what):
        if stage:
            stage(what)

    t = {}
    t0 = time.perf_counter()
    mark("host adjoint: geometry")
    g = Geometry(case.mesh)
    H = HostAdjoint(case, g, threads=threads)
    sc = J.state_scales(case, g, norm_states)
    t["geometry_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mark("host adjoint: connectivity + colouring")
    ncol = H.setup()
    t["connectivity_and_colouring_s"] = time.perf_counter() - t0
    W = np.asarray(case.states, dtype=np.float64)
    t0 = time.perf_counter()
    mark("host adjoint: dRdWTPC (coloured finite differences)")
    P = H.assemble(W, sc, True, pc_blend=pc_blend)
    t["dRdWTPC_fd_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mark("host adjoint: dRdWT (coloured dual numbers)")
    A = H.assemble(W, sc, False)
    t["dRdWT_dual_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mark("host adjoint: factorisation")
    K = OL.OmpKrylov(threads)
    K.set_operator(A)
    op_nnz, pc_nnz = int(A[1].size), int(P[1].size)
    del A
    K.set_pc_bilu(P, pc_structure)
    nagg, agg = coarse
    N = case.mesh.n_cells
    if nagg > 0 and agg.min() >= 0:
        K.set_coarse(P, 3 * N, N, agg)
    del P
    t["factorisation_s"] = time.perf_counter() - t0
    mark("host adjoint: GMRES")
    psi, inf = K.gmres(rhs, restart=restart, max_iters=max_iters, rel_tol=rel_tol, abs_tol=1e-300, max_seconds=max_seconds)
    info = dict(inf)
    info.update(t)
    info.update(threads=K.threads, colors=int(ncol), dRdWT_nnz=op_nnz, dRdWTPC_nnz=pc_nnz, levels=K.levels, matrices="host-assembled",
                jacobian_build_s=t["connectivity_and_colouring_s"] + t["dRdWTPC_fd_s"] + t["dRdWT_dual_s"])
    return psi, info
