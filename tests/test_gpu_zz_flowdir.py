"""Forces parallel / normal to the flow (reference DAFunctionForce.C:45-61,92-113): the direction follows the angle of
attack of a patchVelocity input; values and the partial derivatives w.r.t. [UMag, AoA] (boundary value AND direction)
against the oracle.  (Sorted last on purpose: it composes entry points that the other GPU tests cover one by one.)"""
import numpy as np
import pytest

from common import options
from dafoam_amd.meshgen import channel_case
from oracle.foam_mesh import Geometry
from oracle.functions import force

pytestmark = pytest.mark.gpu


def _with_inlet(case, Umag, aoa_deg):
    import copy

    c2 = copy.copy(case)
    c2.bcs = copy.deepcopy(case.bcs)
    a = aoa_deg * np.pi / 180.0
    c2.bcs["inlet"]["U"] = (case.bcs["inlet"]["U"][0], (Umag * np.cos(a), Umag * np.sin(a), 0.0))
    return c2


def test_force_parallel_and_normal_to_flow():
    from dafoam_amd.pyDAFoam import PYDAFOAM

    case = channel_case(6, 5, 4, wall_function=True)
    g = Geometry(case.mesh)
    W = case.states
    walls = ["bottom", "top"]
    fn = {"CD": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "parallelToFlow",
                 "patchVelocityInputName": "patchV", "scale": 2.0},
          "CL": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "normalToFlow",
                 "patchVelocityInputName": "patchV", "scale": 2.0}}
    D = PYDAFOAM(options=options(case, function=fn, inputInfo={"patchV": {"type": "patchVelocity", "patches": ["inlet"], "flowAxis": "x",
                                                                          "normalAxis": "y"}}), case=case)
    x0 = np.array([10.0, 3.0])  # UMag, AoA [deg]
    D.solver.setSolverInput("patchV", "patchVelocity", 2, x0)

    def ref(name, x):
        a = x[1] * np.pi / 180.0
        d = [np.cos(a), np.sin(a), 0.0] if name == "CD" else [-np.sin(a), np.cos(a), 0.0]
        return force(_with_inlet(case, x[0], x[1]), g, W, walls, d, 2.0)

    for name in ("CD", "CL"):
        Fo = ref(name, x0)
        assert abs(D.solver.calcFunction(name) - Fo) <= 1e-11 * abs(Fo), name
        prod = np.zeros(2)
        D.solverAD.calcJacTVecProduct("patchV", "patchVelocity", x0, name, "function", np.ones(1), prod)
        for k, h in ((0, 1e-4), (1, 1e-3)):
            e = np.eye(2)[k]
            fd = (ref(name, x0 + h * e) - ref(name, x0 - h * e)) / (2 * h)
            assert abs(prod[k] - fd) <= 1e-6 * max(abs(fd), 1e-3 * abs(Fo)), (name, k, prod[k], fd)
