"""GPU box: repeat the Krylov-exhaustion solves of tests/test_gpu_edge_cases.py (1-, 2- and 6-cell meshes, gmresRelTol 1e-12)
N times and log every run whose failure flag or psi error is off.  Usage: python tools/gpu/degenerate_loop.py [N] [out.log]"""
import sys

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from common import norm_states, options, relerr  # noqa: E402
from dafoam_amd.meshgen import channel_case  # noqa: E402
from dafoam_amd.pyDAFoam import PYDAFOAM  # noqa: E402
from oracle import jacobian as J  # noqa: E402
from oracle.foam_mesh import Geometry  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
extra_amd = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    extra_amd[k] = v if not v.lstrip("-").isdigit() else int(v)
bad = 0
for dims in [(1, 1, 1), (2, 1, 1), (3, 2, 1)]:
    case = channel_case(*dims, wall_function=True)
    g = Geometry(case.mesh)
    sc = J.state_scales(case, g, norm_states(case))
    con = J.connectivity(case, g)
    col, _ = J.greedy_coloring(con)
    A = J.jacobian_colored(case, g, case.states, con, col, sc, mode="cs", lower_bound=0)
    rhs = np.ones(case.states.size) * sc
    ref = spla.spsolve(A.tocsc(), rhs)
    its = []
    for rep in range(N):
        D = PYDAFOAM(options=options(case, adjEqnOption={"gmresRelTol": 1e-12, "printInfo": 0}, jacLowerBounds={"dRdW": 0.0, "dRdWPC": 0.0},
                                     **({"amd": extra_amd} if extra_amd else {})), case=case)
        psi, fail = D.solveAdjoint(rhs)
        info = D.ksp.info()
        e = relerr(psi, ref)
        its.append(info["iters"])
        if fail != 0 or not e <= 1e-8:
            bad += 1
            h = D.ksp.history()
            print(f"BAD dims={dims} rep={rep} fail={fail} err={e:.3e} info={info} nrefine={D.ksp.nRefine() if hasattr(D.ksp, 'nRefine') else '?'}", file=out)
            print("   hist/res0:", " ".join(f"{v / h[0]:.2e}" for v in h[:60]), "..." if len(h) > 60 else "", file=out, flush=True)
    print(f"dims={dims} runs={N} iterations min/max {min(its)}/{max(its)} distinct={sorted(set(its))}", file=out, flush=True)
print(f"TOTAL bad {bad}", file=out, flush=True)
sys.exit(1 if bad else 0)
