"""Adjoint of the drag on an OpenFOAM case directory, end to end (needs an MI355X).

    python examples/adjoint_from_openfoam_case.py <caseDir>     # reads constant/polyMesh, 0/U p nuTilda nut (phi)
    python examples/adjoint_from_openfoam_case.py --demo        # writes the synthetic channel as an OpenFOAM case first

The run script is the reference's, with the import changed: options dict -> PYDAFOAM -> solveAdjoint.
The states should be a CONVERGED primal (run DASimpleFoam/simpleFoam first): the adjoint is linearised about them.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dafoam_amd import foam_io  # noqa: E402
from dafoam_amd.meshgen import channel_case, wall_distance, _InputGeometry  # noqa: E402
from dafoam_amd.pyDAFoam import PYDAFOAM  # noqa: E402


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    case_dir = sys.argv[1]
    if case_dir == "--demo":
        case_dir = "demo_case"
        foam_io.write_case(case_dir, channel_case(20, 12, 8, wall_function=True, perturb=0.0))
    mesh = foam_io.read_polymesh(case_dir)
    g = _InputGeometry(mesh)
    y = wall_distance(mesh, g.C, g.Cf, g.Sf)  # frozen wall distance (the reference: meshWaveFrozen)
    case = foam_io.read_case(case_dir, solver_name="DASimpleFoam", y_wall=y)
    walls = [p.name for p in case.mesh.patches if p.type == "wall"]
    daOptions = {
        "solverName": "DASimpleFoam",
        "function": {"CD": {"type": "force", "source": "patchToFace", "patches": walls, "directionMode": "fixedDirection",
                            "direction": [1.0, 0.0, 0.0], "scale": 1.0}},
        "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRelTol": 1e-6, "gmresMaxIters": 1000, "gmresRestart": 300, "pcFillLevel": 1, "asmOverlap": 1},
        "amdColoringDir": case_dir,  # dRdWColoring_1.bin is cached in the case directory, like the reference does
        "writeAdjointFields": True,
    }
    DASolver = PYDAFOAM(options=daOptions, case=case)
    funcs = {}
    DASolver.evalFunctions(funcs)
    print("functions:", funcs)
    DASolver.calcPrimalResidualStatistics("print")
    W = DASolver.getStates()
    dFdW = np.zeros(W.size)
    DASolver.solverAD.calcJacTVecProduct("states", "stateVar", W, "CD", "function", np.ones(1), dFdW)
    psi, fail = DASolver.solveAdjoint(dFdW)
    info = DASolver.ksp.info()
    print(f"adjoint: fail={fail} iterations={info['iters']} |r|/|r0|={info['res'] / info['res0']:.2e} in {info['seconds']:.2f} s")
    print("wrote", DASolver.writeAdjointFields("CD", 9999, psi, caseDir=case_dir))


if __name__ == "__main__":
    main()
