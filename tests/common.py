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


def options(case, **extra):
    o = {"solverName": case.solver_name, "normalizeStates": dict(norm_states(case)), "adjEqnOption": {"printInfo": 0}}
    o.update(extra)
    return o
