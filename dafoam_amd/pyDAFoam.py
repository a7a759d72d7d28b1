"""Host-side mirror of the reference's ``dafoam/pyDAFoam.py`` for the adjoint hot path: the ``DAOPTION``
defaults/keys that the path consumes and the ``PYDAFOAM`` driver methods its callers use
(reference dafoam/pyDAFoam.py:39-661 DAOPTION, :664-2295 PYDAFOAM; callers: dafoam/mphys/mphys_dafoam.py:433-612).

Only hot-path behaviour is implemented: option dict with type checking and first-level
sub-dict merging (pyDAFoam.py:1993-2033), getStates/setStates (:2090-2109), array2Vec/vec2Array (:2167-2199),
and ``solveAdjoint`` which performs the exact call sequence of ``DAFoamSolver.solve_linear``
(mphys_dafoam.py:433-574: runColoring -> calcdRdWT(1, dRdWTPC) -> createMLRKSPMatrixFree ->
initializedRdWTMatrixFree -> solveLinearEqn -> destroydRdWTMatrixFree).
"""
from __future__ import annotations

import copy

import numpy as np

from .pyDASolvers import KSP, Mat, Vec, pyDASolvers


class Error(Exception):
    """Error type of the pyDAFoam mirror (same name and role as the reference's pyDAFoam.Error): the message is
    printed inside a ruled box, 80 columns wide, before the exception propagates."""

    WIDTH = 80

    def __init__(self, message):
        import textwrap

        inner = self.WIDTH - 4
        lines = textwrap.wrap("pyDAFoam Error: " + str(message), width=inner) or [""]
        rule = "+" + "-" * (self.WIDTH - 2) + "+"
        print("\n".join([""] + [rule] + ["| " + ln.ljust(inner) + " |" for ln in lines] + [rule, ""]))
        super().__init__(message)


class DAOPTION(object):
    """The reference's option surface: every DAOPTION attribute of dafoam/pyDAFoam.py:59-661 with its name, nesting and
    default value (line numbers in the comments; tests/test_reference_pins.py parses the reference file with `ast` and
    compares).  Options of subsystems outside the adjoint hot path (mesh quality checks, FFD output, regression models,
    ...) are accepted and carried so that existing runScripts construct unchanged; the hot path reads the ones marked *."""

    def __init__(self):
        self.solverName = "DASimpleFoam"  # * :76
        self.primalMinResTol = 1.0e-8  # * :80
        self.primalFuncStdTol = {"stdTol": -1.0, "slopeTol": -1.0, "funcNames": ["CD"], "nStepsFrac": 0.2}  # :91
        self.primalBC = {}  # * :104
        self.primalInitCondition = {}  # :114
        self.normalizeStates = {}  # * :121
        self.function = {}  # * :238
        self.inputInfo = {}  # * :262
        self.outputInfo = {}  # * :270
        self.designSurfaces = ["ALL_OPENFOAM_WALL_PATCHES"]  # :274
        self.fvSource = {}  # :325
        self.prepareCaseOnly = False  # :331
        self.adjEqnSolMethod = "Krylov"  # * :334
        self.dynamicMesh = {"active": False, "mode": "rotation", "center": [0.25, 0.0, 0.0], "axis": "z", "omega": 0.1, "s": 2.0, "t0": 0.35}  # :338
        bounds = {"U": (-1000.0, 1000.0), "p": (20000.0, 500000.0), "p_rgh": (20000.0, 500000.0), "e": (-220000.0, 55000.0), "T": (100.0, 1000.0),
                  "h": (-200000.0, 200000.0), "D": (-1e16, 1e16), "rho": (0.2, 5.0)}
        bounds.update({v: (1e-16, 1e16) for v in ("nuTilda", "k", "omega", "epsilon", "ReThetat", "gammaInt")})
        self.primalVarBounds = {}  # :353
        for v, (lo, hi) in bounds.items():
            self.primalVarBounds[v + "Max"] = hi
            self.primalVarBounds[v + "Min"] = lo
        self.discipline = "aero"  # * :386
        self.adjPartDerivFDStep = {"State": 1.0e-6}  # * :390
        self.transonicPCOption = -1  # * :396
        self.unsteadyAdjoint = {"mode": "None", "PCMatPrecomputeInterval": 100, "PCMatUpdateInterval": 1, "reduceIO": True,
                                "additionalOutput": ["None"], "additionalOldTime": ["None"], "readZeroFields": True}  # :401
        self.adjPCLag = 10000  # * :417
        self.useAD = {"mode": "reverse", "dvName": "None", "seedIndex": -9999}  # * :426
        self.useConstrainHbyA = True  # * :433
        self.forceMeshWaveFrozen = True  # * :437 (the wall distance is always frozen here)
        self.useDdtCorr = False  # :441
        self.regressionModel = {"active": False}  # :450
        self.useMeanStates = False  # :486
        self.solveLinearFunctionName = "None"  # :494
        self.printDAOptions = True  # :497
        self.debug = False  # * :500
        self.writeJacobians = ["None"]  # * :506
        self.printInterval = 100  # :510
        self.printIntervalUnsteady = 1  # :513
        self.primalMinResTolDiff = 1.0e2  # * :517
        self.adjUseColoring = True  # * :521
        self.adjEqnOption = {  # * :526-548
            "globalPCIters": 0,
            "asmOverlap": 1,
            "localPCIters": 1,
            "jacMatReOrdering": "rcm",
            "pcFillLevel": 1,
            "gmresMaxIters": 1000,
            "gmresRestart": 1000,
            "gmresRelTol": 1.0e-6,
            "gmresAbsTol": 1.0e-14,
            "gmresTolDiff": 1.0e2,
            "useNonZeroInitGuess": False,
            "useMGSO": False,
            "printInfo": 1,
            "fpMaxIters": 1000,
            "fpRelTol": 1e-6,
            "fpMinResTolDiff": 1.0e2,
            "fpPCUpwind": False,
            "dynAdjustTol": False,
            "KSPCalcEigen": 0,
            "KSPCalcSingularVal": 0,
            "readPCMat": 0,
        }
        self.normalizeResiduals = [  # * :551-563
            "URes", "pRes", "p_rghRes", "nuTildaRes", "phiRes", "TRes", "DRes", "kRes", "omegaRes", "epsilonRes", "alpha.waterRes",
        ]
        self.maxResConLv4JacPCMat = {  # * :568-582
            "pRes": 2, "phiRes": 1, "URes": 2, "TRes": 2, "nuTildaRes": 2, "kRes": 2, "epsilonRes": 2, "omegaRes": 2,
            "p_rghRes": 2, "DRes": 2, "gammaIntRes": 2, "ReThetatRes": 2, "alpha.waterRes": 2,
        }
        self.jacLowerBounds = {"dRdW": 1.0e-30, "dRdWPC": 1.0e-30}  # * :586-589
        self.maxTractionBCIters = 100  # :592
        self.decomposeParDict = {  # * :597-604 (preservePatches drives the multi-GPU partitioner)
            "method": "scotch",
            "simpleCoeffs": {"n": [2, 2, 1], "delta": 0.001},
            "kahipCoeffs": {"config": "fast", "imbalance": 0.01},
            "preservePatches": ["None"],
            "singleProcessorFaceSets": ["None"],
            "args": ["None"],
        }
        self.adjStateOrdering = "state"  # * :608
        self.checkMeshThreshold = {"maxAspectRatio": 1000.0, "maxNonOrth": 70.0, "maxSkewness": 4.0, "maxIncorrectlyOrientedFaces": 0}  # :611
        self.writeDeformedFFDs = False  # :619
        self.writeDeformedConstraints = False  # :622
        self.writeAdjointFields = False  # * :625
        self.maxCorrectBCCalls = 2  # * :628
        self.writeMinorIterations = False  # :635
        self.primalMinIters = 1  # :639
        self.tensorflow = {"active": False}  # :642
        self.wallDistanceMethod = "default"  # :650
        self.unsteadyCompOutput = {}  # :661
        # MI355X-specific additions (not in the reference)
        self.amd = {"pcType": "bilu", "pcCoarseAggregates": -1, "pcCoarseField": "p", "pcCoarseMode": "deflated", "pcUpwindBlend": 0.5, "pcBlockCells": 1024, "jacMode": 1, "pcJacMode": 0, "pcFactorFP32": 0, "cgsAlwaysRefine": 0, "gmresOrthogonalization": "dcgs2", "setupThreads": 32}
        self.amdDevice = 0
        ## directory of the dRdWColoring_<nProcs>.bin cache (the reference keeps it in the case directory,
        ## DAJacCon.C:1886-2019); "" = do not cache
        self.amdColoringDir = ""
        # solvePrimal: residual norm that primalMinResTol refers to (0 = the norm at the states the first primal of this object starts
        # from); set it when a run restarts from already converged fields, whose own start norm is ~0
        self.amdPrimalResRef = 0.0


class PYDAFOAM(object):
    """Main driver (reference PYDAFOAM, pyDAFoam.py:664).  ``case`` replaces the OpenFOAM case directory."""

    def __init__(self, comm=None, options=None, case=None, initSolver=True):
        assert options is not None, "options must be provided (reference pyDAFoam.py:679-686)"
        self.name = "PYDAFOAM"
        self.dtype = "d"  # pyDAFoam.py:713
        self.comm = comm
        self.defaultOptions = self._getDefOptions()
        self.imOptions = self._getImmutableOptions()
        self.options = copy.deepcopy(self.defaultOptions)
        for name, value in options.items():
            self._initOption(name, value)
        self._case = case
        # patch names of the case (the reference reads constant/polyMesh/boundary, pyDAFoam.py:1553-1563)
        self.boundaries = {p.name: {"type": p.type, "nFaces": p.size, "startFace": p.start} for p in case.mesh.patches} if case is not None else {}
        self._checkOptions()
        self._readMeshInfo()
        self._computeBasicFamilyInfo()
        # the family groups of the reference (pyDAFoam.py:743-756): every patch, the wall-like patches, the design surfaces
        self.allSurfacesGroup = "allSurfaces"
        self.addFamilyGroup(self.allSurfacesGroup, self.basicFamilies)
        self.allWallsGroup = "allWalls"
        self.addFamilyGroup(self.allWallsGroup, self.wallList)
        self.designSurfacesGroup = "designSurfaces"
        ds = self.getOption("designSurfaces")
        self.addFamilyGroup(self.designSurfacesGroup, self.wallList if "ALL_OPENFOAM_WALL_PATCHES" in ds else list(ds))
        self.mesh = None  # the volume-mesh warping object (setMesh)
        self.solver = self.solverAD = None
        if initSolver:  # (False: options + mesh families only - no device needed)
            self._initSolver()
        self.dRdWTPC = None
        self.ksp = None
        self.nSolveAdjoints = 0
        self.runColoring = True

    # ---------------------------------------------------------------- options (pyDAFoam.py:823-844,1892-2033,2201-2208)
    def _getDefOptions(self):
        d = DAOPTION()
        out = {}
        for key in vars(d):
            v = getattr(d, key)
            out[key] = [type(v), v]
        return out

    def _getImmutableOptions(self):
        return ()

    def _initOption(self, name, value):
        if name not in self.defaultOptions:
            raise Error("Option '%-30s' is not a valid %s option." % (name, self.name))
        if name in self.imOptions:
            raise Error("Option '%-35s' cannot be modified after the solver is created." % name)
        if isinstance(value, self.defaultOptions[name][0]):
            if isinstance(value, dict):
                for subKey in value:
                    self.options[name][1][subKey] = value[subKey]
            else:
                self.options[name][1] = value
        else:
            raise Error(
                "Datatype for Option %-35s was not valid \n Expected data type is %-47s \n Received data type is %-47s"
                % (name, self.defaultOptions[name][0], type(value))
            )

    def getOption(self, name):
        if name in self.defaultOptions:
            return self.options[name][1]
        raise Error("%s is not a valid option name." % name)

    def setOption(self, name, value):
        """Merge up to three sub-dict levels (pyDAFoam.py:1892-1991); call updateDAOption() to push to the solver."""
        if name not in self.defaultOptions:
            raise Error("Option '%-30s' is not a valid %s option." % (name, self.name))
        if not isinstance(value, self.defaultOptions[name][0]):
            raise Error("Datatype for Option %-35s was not valid" % name)
        if isinstance(value, dict):
            def merge(dst, src):
                for k, v in src.items():
                    if isinstance(v, dict) and isinstance(dst.get(k), dict):
                        merge(dst[k], v)
                    else:
                        dst[k] = v
            merge(self.options[name][1], value)
        else:
            self.options[name][1] = value

    def updateDAOption(self):
        plain = {k: v[1] for k, v in self.options.items()}
        self.solver.updateDAOption(plain)

    def _checkOptions(self):
        """pyDAFoam.py:846-899: combinations of options that cannot work are rejected before the solver is built."""
        if self.getOption("useAD")["mode"] not in ["reverse", "forward"]:
            raise Error("useAD->mode only supports reverse, or forward!")
        if self.getOption("discipline") not in ["aero", "thermal"]:
            raise Error("discipline: %s not supported. Options are: aero or thermal" % self.getOption("discipline"))
        for key, label in (("primalBC", "primalBC"), ("function", "function")):
            for objKey, entry in self.getOption(key).items():
                patches = entry.get("patches") if isinstance(entry, dict) else None
                for patchName in patches or []:
                    if patchName not in self.boundaries.keys():
                        raise Error("%s-%s-patches-%s is not valid. Please use a patchName from the boundaries list: %s"
                                    % (label, objKey, patchName, self.boundaries.keys()))

    # ---------------------------------------------------------------- solver init (pyDAFoam.py:1417-1452)
    def _initSolver(self):
        solverName = self.getOption("solverName")
        plain = {k: v[1] for k, v in self.options.items()}
        solverArg = (solverName + " -python").encode()
        self.solver = pyDASolvers(solverArg, plain, case=self._case)
        # the reference keeps a second, CoDiPack-typed instance (solverAD); here one GPU object serves both roles
        self.solverAD = self.solver
        self.solver.initSolver()

    # ---------------------------------------------------------------- states / vectors
    def getStates(self):
        n = self.solver.getNLocalAdjointStates()
        states = np.zeros(n, self.dtype)
        self.solver.getOFFields(states)
        return states

    def setStates(self, states):
        self.solver.updateOFFields(states)
        self.solverAD.updateOFFields(states)

    def getNLocalAdjointStates(self):
        return self.solver.getNLocalAdjointStates()

    def getNLocalPoints(self):
        return self.solver.getNLocalPoints()

    def setVolCoords(self, vol_coords):
        """pyDAFoam.py:2111-2119"""
        self.xv = np.asarray(vol_coords, dtype=self.dtype).reshape(-1, 3).copy()
        if self.solver is None:
            return
        self.solver.updateOFMesh(vol_coords)
        if self.solverAD is not self.solver:
            self.solverAD.updateOFMesh(vol_coords)

    # ---------------------------------------------------------------- surface families (pyDAFoam.py:941-1125,1553-1800)
    # What pyGeo / IDWarp and mphys need around the solver: the boundary patches as "families" of surface points.  The
    # reference parses constant/polyMesh; here the FoamCase holds the same arrays.  A family is a list of basic-family ids, a
    # basic family a patch with its unique point ids ("indicesRed", ascending) and its faces in that reduced numbering
    # ("facesRed"); surface arrays of a group are the concatenation over its (sorted) basic families.
    def _readMeshInfo(self):
        """pyDAFoam.py:1553-1563: point coordinates xv0 / xv, faces, owners, neighbours, boundaries."""
        if self._case is None:
            self.xv0 = self.xv = np.zeros((0, 3))
            self.faces, self.owners, self.neighbours = [], np.zeros(0, int), np.zeros(0, int)
            return
        m = self._case.mesh
        self.xv0 = np.array(m.points, dtype=self.dtype).reshape(-1, 3)
        self.xv = self.xv0.copy()
        self.faces = [m.face_pts[m.face_ptr[f]:m.face_ptr[f + 1]].tolist() for f in range(m.n_faces)] if m.n_faces < 200000 else None
        self.owners, self.neighbours = np.asarray(m.owner), np.asarray(m.neighbour)
        for pt in m.patches:
            self.boundaries[pt.name]["faces"] = np.arange(pt.start, pt.start + pt.size)

    def _computeBasicFamilyInfo(self):
        """pyDAFoam.py:1674-1748: per patch the unique point ids and the faces in that reduced numbering; wall-like patches."""
        self.families = {}
        self.basicFamilies = sorted(self.boundaries.keys())
        self.wallList = []
        m = self._case.mesh if self._case is not None else None
        for counter, name in enumerate(self.basicFamilies):
            self.families[name] = [counter]
            bc = self.boundaries[name]
            fids = np.asarray(bc.get("faces", []), dtype=np.int64)
            if fids.size:
                lo, hi = m.face_ptr[fids], m.face_ptr[fids + 1]
                flat = np.concatenate([m.face_pts[a:b] for a, b in zip(lo, hi)]) if np.any(hi - lo != hi[0] - lo[0]) else \
                    m.face_pts[(lo[:, None] + np.arange(hi[0] - lo[0])[None, :]).ravel()]
                indices, inv = np.unique(flat, return_inverse=True)
                sizes = (hi - lo).astype(np.int64)
                cuts = np.concatenate([[0], np.cumsum(sizes)])
                bc["facesRed"] = [inv[cuts[i]:cuts[i + 1]].tolist() for i in range(fids.size)]
                bc["indicesRed"] = indices.tolist()
            else:
                bc["facesRed"], bc["indicesRed"] = [], []
            if bc["type"] in ("wall", "slip", "cyclic"):
                self.wallList.append(name)

    def addFamilyGroup(self, groupName, families):
        """pyDAFoam.py:941-977: a named union of families (groups may be nested)."""
        if groupName in self.families:
            raise Error("The specified groupName '%s' already exists in the mesh file or has already been added." % groupName)
        indices = []
        for fam in families:
            if fam not in self.families:
                raise Error("The specified family '%s' for group '%s', does not exist in the mesh file or has not already been added. "
                            "The current list of families (original and grouped) is: %s" % (fam, groupName, repr(self.families.keys())))
            indices.extend(self.families[fam])
        self.families[groupName] = sorted(np.unique(indices).tolist())

    def printFamilyList(self):
        print(self.families)

    def _getSurfaceSize(self, groupName):
        """pyDAFoam.py:1630-1660: (points, faces) of a group."""
        if groupName is None:
            groupName = self.allSurfacesGroup
        if groupName not in self.families:
            raise Error("'%s' is not a family in the OpenFoam Case or has not been added as a combination of families" % groupName)
        nPts = sum(len(self.boundaries[self.basicFamilies[i]]["indicesRed"]) for i in self.families[groupName])
        nCells = sum(len(self.boundaries[self.basicFamilies[i]]["facesRed"]) for i in self.families[groupName])
        return nPts, nCells

    def getSurfaceCoordinates(self, groupName=None):
        """pyDAFoam.py:1594-1628: the (nPoints, 3) coordinates of the group's surface points (current volume coordinates)."""
        if groupName is None:
            groupName = self.allWallsGroup
        npts, _ = self._getSurfaceSize(groupName)
        ids = [np.asarray(self.boundaries[self.basicFamilies[i]]["indicesRed"], dtype=np.int64) for i in self.families[groupName]]
        xs = np.zeros((npts, 3), self.dtype)
        if npts:
            xs[:] = self.xv[np.concatenate(ids)]
        return xs

    def getSurfaceConnectivity(self, groupName=None):
        """pyDAFoam.py:1000-1047: (conn, faceSizes) of the group's faces in the numbering of getSurfaceCoordinates."""
        if groupName is None:
            groupName = self.allWallsGroup
        conn, faceSizes, offset = [], [], 0
        for i in self.families[groupName]:
            bc = self.boundaries[self.basicFamilies[i]]
            if bc["facesRed"]:
                for face in bc["facesRed"]:
                    conn.extend(v + offset for v in face)
                    faceSizes.append(len(face))
                offset += len(bc["indicesRed"])
        return conn, faceSizes

    def getTriangulatedMeshSurface(self, groupName=None, **kwargs):
        """pyDAFoam.py:1049-1113: [p0, v1, v2] - every face as a fan of triangles about its average point (for DVConstraints)."""
        if groupName is None:
            groupName = self.allWallsGroup
        pts = self.getSurfaceCoordinates(groupName)
        conn, faceSizes = self.getSurfaceConnectivity(groupName)
        p0, v1, v2, c = [], [], [], 0
        for fs in faceSizes:
            nodes = conn[c:c + fs]
            avg = pts[nodes].mean(axis=0)
            for k in range(fs):
                p0.append(avg)
                v1.append(pts[nodes[k]] - avg)
                v2.append(pts[nodes[(k + 1) % fs]] - avg)
            c += fs
        return [p0, v1, v2]

    def mapVector(self, vec1, groupName1, groupName2, vec2=None):
        """pyDAFoam.py:1768-1850: a surface array on group 1 -> the layout of group 2 (entries of families in both groups are
        copied, the others stay what vec2 held / zero)."""
        if groupName1 not in self.families or groupName2 not in self.families:
            raise Error("'%s' or '%s' is not a family in the mesh file or has not been added as a combination of families" % (groupName1, groupName2))
        if vec2 is None:
            vec2 = np.zeros((self._getSurfaceSize(groupName2)[0], 3), self.dtype)
        start1, off = {}, 0
        for i in self.families[groupName1]:
            start1[i] = off
            off += len(self.boundaries[self.basicFamilies[i]]["indicesRed"])
        off = 0
        for i in self.families[groupName2]:
            n = len(self.boundaries[self.basicFamilies[i]]["indicesRed"])
            if i in start1:
                vec2[off:off + n] = vec1[start1[i]:start1[i] + n]
            off += n
        return vec2

    def getSolverMeshIndices(self):
        """pyDAFoam.py:1750-1766: the global indices of this rank's volume coordinates for the warping object (one domain: 0..3P)."""
        return np.arange(self.xv0.size)

    def setMesh(self, mesh):
        """pyDAFoam.py:979-998: attach the volume-mesh warping object (IDWarp USMesh interface: setExternalMeshIndices,
        setSurfaceDefinition, setSurfaceCoordinates)."""
        self.mesh = mesh
        self.mesh.setExternalMeshIndices(self.getSolverMeshIndices())
        conn, faceSizes = self.getSurfaceConnectivity(self.allWallsGroup)
        self.mesh.setSurfaceDefinition(self.getSurfaceCoordinates(self.allWallsGroup), conn, faceSizes)

    def setSurfaceCoordinates(self, coordinates, groupName=None):
        """pyDAFoam.py:1565-1592: new surface points of a group -> the warping object (which returns volume coordinates later)."""
        if self.mesh is None:
            return
        if groupName is None:
            groupName = self.allWallsGroup
        self._updateGeomInfo = True
        surf = self.mapVector(coordinates, groupName, self.allWallsGroup, self.getSurfaceCoordinates(self.allWallsGroup))
        self.mesh.setSurfaceCoordinates(surf)

    def set_solver_input(self, inputs, DVGeo=None):
        """pyDAFoam.py:1350-1374 (called by mphys_dafoam.py solve_nonlinear / linearize / apply_linear): every inputInfo entry
        attached to the solver component is handed to both solver objects.  The forward-mode seeds of the reference's ADF build
        are not mirrored (the GPU path differentiates through calcJacTVecProduct / calcJacVecProduct): useAD mode "forward" with
        a seeded design variable is rejected instead of being ignored."""
        inputDict = self.getOption("inputInfo")
        if self.getOption("useAD")["mode"] == "forward" and self.getOption("useAD").get("dvName", "None") not in ("None", None, ""):
            raise Error("set_solver_input: forward-mode seeds (useAD mode forward + dvName) are not available on this path")
        for inputName in list(inputDict.keys()):
            if "solver" not in inputDict[inputName].get("components", ["solver"]):
                continue
            inputType = inputDict[inputName]["type"]
            value = np.ascontiguousarray(inputs[inputName], dtype=np.float64).ravel()
            if inputType == "volCoord":
                self.xv = value.reshape(-1, 3).copy()
            self.solver.setSolverInput(inputName, inputType, len(value), value, np.zeros(len(value)))
            if self.solverAD is not self.solver:
                self.solverAD.setSolverInput(inputName, inputType, len(value), value, np.zeros(len(value)))

    def setPrimalBoundaryConditions(self, printInfo=1, printInfoAD=0):
        """pyDAFoam.py:1662-1667"""
        self._primalResRef = None  # new boundary values = a new problem: solvePrimal takes its reference norm anew
        self.solver.setPrimalBoundaryConditions(printInfo)
        if self.solverAD is not self.solver:
            self.solverAD.setPrimalBoundaryConditions(printInfoAD)

    def readStateVars(self, timeVal=0.0, deltaT=0.0):
        """pyDAFoam.py:1321-1348: the state fields of a time directory -> both solver objects (steady: one time level)."""
        self.solver.readStateVars(timeVal, 0)
        if self.solverAD is not self.solver:
            self.solverAD.readStateVars(timeVal, 0)

    def getResiduals(self):
        """pyDAFoam.py:2121-2130"""
        residuals = np.zeros(self.solver.getNLocalAdjointStates(), self.dtype)
        self.solver.getResiduals(residuals)
        return residuals

    def evalFunctions(self, funcs):
        """pyDAFoam.py:917-939: funcs[name] = value for every entry of the "function" option (steady solvers: the
        time-operator value is the function value at the current states)."""
        for funcName in list(self.getOption("function").keys()):
            funcs[funcName] = self.solver.calcFunction(funcName)

    def calcPrimalResidualStatistics(self, mode):
        """pyDAFoam.py:901-905"""
        return self.solverAD.calcPrimalResidualStatistics(mode)

    def writeAdjointFields(self, function, writeTime, psi, caseDir="."):
        """pyDAFoam.py:907-915"""
        if self.getOption("writeAdjointFields"):
            if len(self.getOption("function").keys()) > 1:
                raise Error("writeAdjointFields supports only one function, while multiple are defined!")
            return self.solver.writeAdjointFields(function, writeTime, psi, caseDir=caseDir)

    def arrayVal2Vec(self, array1, vec):
        """pyDAFoam.py:2132-2149"""
        Istart, Iend = vec.getOwnershipRange()
        assert Iend - Istart == len(array1), "array1 and vec must have the same size"
        vec.array[Istart:Iend] = array1

    def vecVal2Array(self, vec, array1):
        """pyDAFoam.py:2151-2165"""
        Istart, Iend = vec.getOwnershipRange()
        assert Iend - Istart == len(array1), "array1 and vec must have the same size"
        array1[:] = vec.array[Istart:Iend]

    def vec2Array(self, vec):
        Istart, Iend = vec.getOwnershipRange()
        array1 = np.zeros(Iend - Istart, self.dtype)
        array1[:] = vec.array[Istart:Iend]
        return array1

    def array2Vec(self, array1):
        vec = Vec(len(array1))
        vec.array[:] = array1
        return vec

    # ---------------------------------------------------------------- primal (pyDAFoam.py __call__ / solvePrimal)
    def solvePrimal(self, maxSteps=80):
        """Converge the flow residuals from the current states (reference PYDAFOAM.solvePrimal -> DASimpleFoam::solvePrimal,
        DASimpleFoam.C:123-185).  primalMinResTol is an ABSOLUTE bound on a normalised residual, as in the reference
        (`primalMaxRes < primalMinResTol`, DASolver.C:188: primalMaxRes is the largest of OpenFOAM's normalised initial
        residuals, O(1) for a start from scratch): here the residual 2-norm divided by the norm at the states the FIRST
        primal of this object started from, so that a warm-started call inside an optimisation loop converges to the same
        level instead of 1e-8 below an already converged start.  The reference norm is taken anew after
        setPrimalBoundaryConditions (a different problem), and `amdPrimalResRef` > 0 overrides it - needed when the very first call
        starts from converged fields, whose own norm is ~0 (a deviation from the reference, where primalMaxRes is OpenFOAM's
        normalised initial residual, DASolver.C:188).  The failure flag follows
        DASolver::checkPrimalFailure (DASolver.C:2722-2760): fail when primalMaxRes / primalMinResTol > primalMinResTolDiff."""
        tol = float(self.getOption("primalMinResTol"))
        ref = getattr(self, "_primalResRef", None)
        user_ref = float(self.getOption("amdPrimalResRef") or 0.0)
        if user_ref > 0.0:
            ref = self._primalResRef = user_ref
        if ref is None:
            R = np.zeros(self.getNLocalAdjointStates())
            self.solver.getResiduals(R)
            ref = float(np.linalg.norm(R))
            self._primalResRef = ref if ref > 0.0 else 1.0
            ref = self._primalResRef
        amd = self.getOption("amd") or {}
        if str(amd.get("primalMethod", "newton")) == "simple":
            # the reference's own loop, SIMPLE sweeps on the device (das_simple_iteration): blocks of sweeps until the residual norm meets the
            # tolerance or `maxSteps` blocks of amd.simpleSweepsPerCheck sweeps are done
            per = int(amd.get("simpleSweepsPerCheck", 10))
            R = np.zeros(self.getNLocalAdjointStates())
            self.solver.getResiduals(R)
            hist, inner = [float(np.linalg.norm(R))], 0
            for _ in range(int(maxSteps)):
                if hist[-1] <= tol * ref:
                    break
                it = self.solver.simpleIteration(per, alphaP=float(amd.get("simpleAlphaP", 0.3)), linTol=float(amd.get("simpleLinearTol", 1e-6)),
                                                 maxLinIters=int(amd.get("simpleLinearIters", 2000)))
                inner += per * (it["U"] + it["p"] + it["nuTilda"])
                self.solver.getResiduals(R)
                hist.append(float(np.linalg.norm(R)))
            self.primalInfo = dict(steps=(len(hist) - 1) * per, linearIterations=inner, res0=hist[0], res=hist[-1], history=np.asarray(hist), method="simple")
        else:
            _, self.primalInfo = self.solver.solvePrimal(maxSteps=maxSteps, relTol=0.0, absTol=tol * ref)
        self.primalMaxRes = self.primalInfo["res"] / ref
        self.primalInfo["primalMaxRes"] = self.primalMaxRes
        self.primalFail = int(self.primalMaxRes / tol > float(self.getOption("primalMinResTolDiff")))
        W = self.getStates()
        if self.solverAD is not self.solver:
            self.solverAD.updateOFFields(W)
        return self.primalFail

    def __call__(self):
        """One analysis, like the reference's DASolver(): the primal, then the functions are available through evalFunctions."""
        return self.solvePrimal()

    # ---------------------------------------------------------------- adjoint (mphys_dafoam.py:433-574 sequence)
    def solveAdjoint(self, dFdWArray, psi0=None):
        """psi with D_s (dR/dW)^T psi = dFdW (dFdW already state-scaled like the output of
        calcJacTVecProduct(stateVar -> function), DASolver.C:1820).  Returns (psi, fail).  Follows
        DAFoamSolver.solve_linear (dafoam/mphys/mphys_dafoam.py:433-574): colouring once, dRdWTPC + KSP rebuilt every
        adjPCLag solves (or read from dRdWTPC.bin under adjEqnOption.readPCMat, :469-471), matrix-free operator per
        solve, optional non-zero initial guess `psi0` and dynamically adjusted tolerance (dynAdjustTol, :540-544,576-611)."""
        dFdW = self.array2Vec(np.ascontiguousarray(dFdWArray, dtype=np.float64)) if np.ndim(dFdWArray) == 1 else None
        if self.getOption("adjUseColoring") and self.runColoring:
            self.solver.runColoring(cacheDir=self.getOption("amdColoringDir") or None)
            self.runColoring = False
        adjPCLag = self.getOption("adjPCLag")
        writeJac = self.getOption("writeJacobians")
        adjOpt = self.getOption("adjEqnOption")
        if self.nSolveAdjoints % adjPCLag == 0 or self.dRdWTPC is None:
            if adjOpt.get("readPCMat"):
                from .petsc_io import read_mat

                self.dRdWTPC = Mat.from_scipy(read_mat("dRdWTPC.bin"))
            else:
                self.dRdWTPC = Mat().create()
                self.solver.calcdRdWT(1, self.dRdWTPC)
            # DASolver.C:1080-1085: calcdRdWT writes the matrix it has just built under the "dRdWT" key (matName
            # dRdWTPC when isPC = 1); the explicit "dRdWTPC" key is kept as an alias
            if any(k in writeJac for k in ("dRdWT", "dRdWTPC", "all")):
                from .petsc_io import write_mat

                write_mat("dRdWTPC.bin", self.dRdWTPC.to_scipy())
            self.ksp = KSP().create()
            self.solverAD.createMLRKSPMatrixFree(self.dRdWTPC, self.ksp)
        self.solverAD.initializedRdWTMatrixFree()
        if "dRdWT" in writeJac or "all" in writeJac:
            from .petsc_io import write_mat

            tmp = Mat().create()
            self.solver.calcdRdWT(0, tmp)
            write_mat("dRdWT.bin", tmp.to_scipy())
            tmp.destroy()
        if "dRdWColoring" in writeJac or "all" in writeJac:  # DAJacCon.C:1943,1973-1975: colours stored as doubles
            from .petsc_io import write_vec

            write_vec("dRdWColoring_1.bin", self.solver.getColoring()[0].astype(float))
        if np.ndim(dFdWArray) == 2:
            # several objective functions at once: one block GMRES instead of the reference's loop over the functions
            # (mphys_dafoam.py:478-481); columns = functions
            psiB = np.zeros_like(np.asarray(dFdWArray, dtype=np.float64))
            fail, _, _ = self.solverAD.solveLinearEqnBlock(self.ksp, dFdWArray, psiB)
            self.solverAD.destroydRdWTMatrixFree()
            self.nSolveAdjoints += 1
            return psiB, fail
        psi = Vec(len(dFdWArray))
        psi.set(0)
        if adjOpt.get("useNonZeroInitGuess") and psi0 is not None:
            psi.array[:] = np.asarray(psi0, dtype=np.float64)
        if adjOpt.get("dynAdjustTol"):
            self._updateKSPTolerances(psi, dFdW, self.ksp)
        fail = self.solverAD.solveLinearEqn(self.ksp, dFdW, psi)
        self.solverAD.destroydRdWTMatrixFree()
        self.nSolveAdjoints += 1
        return self.vec2Array(psi), fail

    def _updateKSPTolerances(self, psi, dFdW, ksp):
        """dynAdjustTol (reference DAFoamSolver._updateKSPTolerances, mphys_dafoam.py:576-611): converge gmresRelTol
        orders below the CURRENT residual of the initial guess: atol = max(||A psi0 - dFdW|| * rtol0, atol0), rtol = 0."""
        n = dFdW.array.size
        r = np.zeros(n)
        self.solverAD.calcJacTVecProduct("states", "stateVar", self.getStates(), "residuals", "residual", np.ascontiguousarray(psi.array), r)
        rNorm = float(np.linalg.norm(r - dFdW.array))
        opt = self.getOption("adjEqnOption")
        ksp.setTolerances(rtol=0.0, atol=max(rNorm * opt["gmresRelTol"], opt["gmresAbsTol"]), divtol=None, max_it=None)
