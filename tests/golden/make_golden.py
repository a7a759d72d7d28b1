"""Generates tests/golden/*.npz from the ORACLE (the reference itself cannot be built or imported in this
container - SURVEY.md section 0.3/8c - so these fixtures pin the oracle, not the reference).
Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dafoam_amd.meshgen import channel_case, scalar_transport_case  # noqa: E402
from oracle import jacobian as J  # noqa: E402
from oracle.foam_mesh import Geometry  # noqa: E402
from oracle.residual import residual  # noqa: E402

NS = {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0, "T": 1.0}
here = os.path.dirname(os.path.abspath(__file__))


def make(case, name):
    g = Geometry(case.mesh)
    W = case.states
    sc = J.state_scales(case, g, NS)
    con = J.connectivity(case, g)
    col, _ = J.greedy_coloring(con)
    A = J.jacobian_colored(case, g, W, con, col, sc, mode="cs", lower_bound=0)
    rng = np.random.default_rng(0)
    seed = rng.standard_normal(W.size)
    rhs = np.zeros(W.size)
    rhs[: g.nC * (3 if case.solver_name == "DASimpleFoam" else 1) : (3 if case.solver_name == "DASimpleFoam" else 1)] = g.V
    rhs *= sc
    psi = spla.spsolve(A.tocsc(), rhs)
    np.savez_compressed(os.path.join(here, name), W=W, R=residual(case, g, W), R_pc=residual(case, g, W, isPC=True),
                        seed=seed, dRdWTPsi=A @ seed, rhs=rhs, psi=psi)


make(channel_case(4, 4, 3), "oracle_channel_443.npz")
make(scalar_transport_case(5, 4, 3), "oracle_scalar_543.npz")
print("golden fixtures written")
