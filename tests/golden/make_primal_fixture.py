"""Generates dafoam_amd/data/channel_primal_coarse.npz: a CONVERGED coarse-mesh primal state of the bench channel,
computed with the oracle's SIMPLE restatement (oracle/primal.py).  It is input data: bench.py and the large-size
GPU tests prolong it to the fine mesh (dafoam_amd.meshgen.prolong_channel_state).
Run:  python tests/golden/make_primal_fixture.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dafoam_amd.meshgen import channel_case  # noqa: E402
from oracle.foam_mesh import Geometry  # noqa: E402
from oracle.primal import solve_primal  # noqa: E402

dims = (32, 16, 12)
lengths = (2.0, 0.2, 0.2)
grading = 4.0
case = channel_case(*dims, lengths=lengths, grading_y=grading, perturb=0.0)
g = Geometry(case.mesh)
t = time.time()
W, hist = solve_primal(case, g, max_iters=1500, tol=1e-11, verbose=True)
print("converged in", len(hist) * 10, "iterations,", time.time() - t, "s; final residual RMS (U,p,nuTilda,phi):", hist[-1])
out = os.path.join(ROOT, "dafoam_amd", "data", "channel_primal_coarse.npz")
np.savez_compressed(out, dims=np.array(dims), lengths=np.array(lengths), grading_y=grading, W=W, residual_rms=hist[-1])
print("written", out, os.path.getsize(out), "bytes")
