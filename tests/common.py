import os

import numpy as np

NORM_STATES = {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}  # reference tests/runRegTests_AeroOpt.py:83


NORM_STATES_RHO = {"U": 50.0, "p": 1.0e5, "T": 300.0, "nuTilda": 1e-3, "phi": 1.0}


def norm_states(case):
    if getattr(case, "has_T", False):
        return dict(NORM_STATES, T=300.0)
    return NORM_STATES_RHO if case.solver_name in ("DARhoSimpleFoam", "DATurboFoam") else dict(NORM_STATES, T=1.0)


def blocks(case, g):
    N = g.nC
    if case.solver_name in ("DARhoSimpleFoam", "DATurboFoam") or getattr(case, "has_T", False):
        return (("U", slice(0, 3 * N)), ("p", slice(3 * N, 4 * N)), ("T", slice(4 * N, 5 * N)), ("nuTilda", slice(5 * N, 6 * N)),
                ("phi", slice(6 * N, 6 * N + g.nF)))
    if case.solver_name == "DASimpleFoam":
        return (("U", slice(0, 3 * N)), ("p", slice(3 * N, 4 * N)), ("nuTilda", slice(4 * N, 5 * N)), ("phi", slice(5 * N, 5 * N + g.nF)))
    return (("T", slice(0, N)),)


def relerr(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# The parity tests compare PC residuals / matrices / iteration histories with the oracle's restatement of the REFERENCE's PC semantics
# (div(pc) = upwind, DAResidualSimpleFoam.C:125-132; additive coarse correction as in rounds 2-4): they pin those two options.  The
# library's own defaults (round 5: amd.pcUpwindBlend 0.5, amd.pcCoarseMode "deflated") are what tests/test_gpu_naca.py, the bench-flow
# test and smoke() run with.
REFERENCE_PC = {"pcUpwindBlend": 0.0, "pcCoarseMode": "additive"}


def options(case, **extra):
    o = {"solverName": case.solver_name, "normalizeStates": dict(norm_states(case)), "adjEqnOption": {"printInfo": 0}}
    amd = dict(REFERENCE_PC)
    amd.update(extra.pop("amd", {}) or {})
    # DAS_TEST_AMD="key=value,key=value": run a tier with other amd.* switches (e.g. gradFaceParallel=2: the face-parallel gradient kernel for
    # the dual-number passes as well) without editing the tests
    for kv in filter(None, os.environ.get("DAS_TEST_AMD", "").split(",")):
        k, v = kv.split("=", 1)
        amd.setdefault(k, int(v) if v.lstrip("-").isdigit() else v)
    o.update(extra)
    o["amd"] = amd
    return o


def unrolled_maps(c1, c3):
    """Index map between the states/residual rows of a z-periodic (cyclic front/back) channel `c1` and the MIDDLE copy of its
    3-copy unrolled, non-periodic twin `c3` (dafoam_amd.meshgen.periodic_channel_case(copies=3)).  Returns (idx, sign):
    row r of the cyclic case corresponds to sign[r] * (row idx[r] of the unrolled case).  Cyclic 'front' faces map to the
    interface below the middle copy with flipped orientation, 'back' faces to the interface above it."""
    m1, m3 = c1.mesh, c3.mesh
    N, F1, nIF1 = m1.n_cells, m1.n_faces, m1.n_internal_faces
    N3, nIF3 = m3.n_cells, m3.n_internal_faces
    pair3 = {(int(m3.owner[f]), int(m3.neighbour[f])): f for f in range(nIF3)}
    sl1 = {p.name: slice(p.start, p.start + p.size) for p in m1.patches}
    sl3 = {p.name: slice(p.start, p.start + p.size) for p in m3.patches}
    fidx = np.zeros(F1, dtype=np.int64)
    fsgn = np.ones(F1)
    for f in range(nIF1):
        fidx[f] = pair3[(int(m1.owner[f]) + N, int(m1.neighbour[f]) + N)]
    for nm in ("inlet", "outlet", "bottom", "top"):
        by_cell = {int(m3.owner[f]): f for f in range(sl3[nm].start, sl3[nm].stop)}
        for f in range(sl1[nm].start, sl1[nm].stop):
            fidx[f] = by_cell[int(m1.owner[f]) + N]
    front, back = sl1["front"], sl1["back"]
    for k in range(front.stop - front.start):
        cf, cb = int(m1.owner[front.start + k]), int(m1.owner[back.start + k])  # (i,j,0) and (i,j,nz-1)
        fidx[front.start + k] = pair3[(cb, cf + N)]  # top of copy 0 -> bottom of the middle copy, normal +z
        fsgn[front.start + k] = -1.0
        fidx[back.start + k] = pair3[(cb + N, cf + 2 * N)]
    nsc = (c1.states.size - 3 * N - F1) // N
    cells = np.arange(N) + N
    idx = np.concatenate([np.repeat(3 * cells, 3) + np.tile(np.arange(3), N)] + [(3 + b) * N3 + cells for b in range(nsc)] + [(3 + nsc) * N3 + fidx])
    sign = np.concatenate([np.ones((3 + nsc) * N), fsgn])
    return idx, sign
