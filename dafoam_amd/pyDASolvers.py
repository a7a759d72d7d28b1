"""Host-side mirror of the reference's Cython boundary class ``pyDASolvers``
(reference src/pyDASolvers/pyDASolvers.pyx:117-482) for the adjoint hot path.

Same method names, argument meaning and error behaviour as the reference for the hot-path subset
(SURVEY.md section 8b); every method forwards to the C-ABI of include/dafoam_amd.h, which runs on the GPU.
PETSc is not part of this stack (north star: "no PETSc"): the ``Mat``/``Vec``/``KSP`` objects the reference's
callers create with petsc4py (reference dafoam/mphys/mphys_dafoam.py:468-475,519-529, dafoam/pyDAFoam.py:2132-2199)
are replaced by the light stand-ins below that expose the handful of methods those callers use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import CaseStruct, check, dptr, lib
from .meshgen import FoamCase


# ----------------------------------------------------------------------------- PETSc stand-ins
class Vec:
    """Stand-in for petsc4py.PETSc.Vec as used by pyDAFoam.array2Vec/vec2Array (pyDAFoam.py:2167-2199)."""

    def __init__(self, n=0):
        self.array = np.zeros(int(n), dtype=np.float64)

    @classmethod
    def createSeq(cls, n, bsize=1, comm=None):
        return cls(n)

    def setSizes(self, size, bsize=1):
        n = size[0] if isinstance(size, (tuple, list)) else size
        self.array = np.zeros(int(n), dtype=np.float64)

    def setFromOptions(self):
        pass

    def getOwnershipRange(self):
        return 0, self.array.size

    def getSize(self):
        return self.array.size

    def set(self, v):
        self.array[:] = v

    def zeroEntries(self):
        self.array[:] = 0.0

    def duplicate(self):
        return Vec(self.array.size)

    def copy(self, other=None):
        if other is None:
            o = Vec(self.array.size)
            o.array[:] = self.array
            return o
        other.array[:] = self.array
        return other

    def axpy(self, a, x):
        self.array += a * x.array

    def scale(self, a):
        self.array *= a

    def norm(self, norm_type=2):
        return float(np.linalg.norm(self.array))

    def assemblyBegin(self):
        pass

    def assemblyEnd(self):
        pass

    def __getitem__(self, i):
        return self.array[i]

    def __setitem__(self, i, v):
        self.array[i] = v

    def destroy(self):
        self.array = np.zeros(0)


class Mat:
    """Stand-in for the PETSc Mat dRdWT/dRdWTPC handle (device-resident CSR owned by the C-ABI)."""

    def __init__(self):
        self.handle = None

    def create(self, comm=None):
        return self

    def _set(self, h):
        self.destroy()
        self.handle = h

    def getSize(self):
        n = lib().das_mat_rows(self.handle)
        return n, n

    def getInfo(self):
        return {"nz_used": float(lib().das_mat_nnz(self.handle))}

    def to_scipy(self):
        import scipy.sparse as sp

        L = lib()
        n, nnz = L.das_mat_rows(self.handle), L.das_mat_nnz(self.handle)
        rp = np.empty(n + 1, np.int64)
        ci = np.empty(nnz, np.int32)
        v = np.empty(nnz, np.float64)
        check(L.das_mat_export(self.handle, rp.ctypes.data_as(_capi.c_ll_p), ci.ctypes.data_as(_capi.c_int_p), dptr(v)))
        return sp.csr_matrix((v, ci, rp), shape=(n, n))

    @staticmethod
    def from_scipy(A):
        """Device handle of a host CSR matrix (e.g. a dRdWTPC.bin read with petsc_io.read_mat: adjEqnOption.readPCMat)."""
        import scipy.sparse as sp

        A = sp.csr_matrix(A)
        A.sort_indices()
        rp, ci, v = A.indptr.astype(np.int64), A.indices.astype(np.int32), np.ascontiguousarray(A.data, dtype=np.float64)
        h = C.c_void_p()
        check(lib().das_mat_create_from_csr(A.shape[0], rp.ctypes.data_as(_capi.c_ll_p), ci.ctypes.data_as(_capi.c_int_p), dptr(v), C.byref(h)))
        m = Mat()
        m._set(h)
        return m

    def mult(self, x: Vec, y: Vec):
        check(lib().das_mat_mult(self.handle, dptr(x.array), dptr(y.array)))

    def destroy(self):
        if self.handle:
            lib().das_mat_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class KSP:
    """Stand-in for PETSc.KSP (created by the caller, configured by createMLRKSPMatrixFree)."""

    def __init__(self):
        self.handle = None
        self._tols = {}

    def create(self, comm=None):
        return self

    def setTolerances(self, rtol=None, atol=None, divtol=None, max_it=None):
        # mphys_dafoam.py:597-611 adjusts tolerances on the KSP; forwarded to adjEqnOption at solve time
        if rtol is not None:
            self._tols["adjEqnOption.gmresRelTol"] = float(rtol)
        if atol is not None:
            self._tols["adjEqnOption.gmresAbsTol"] = float(atol)
        if max_it is not None:
            self._tols["adjEqnOption.gmresMaxIters"] = int(max_it)

    def getIterationNumber(self):
        it = C.c_int(0)
        check(lib().das_ksp_get_info(self.handle, C.byref(it), None, None, None))
        return it.value

    def getResidualNorm(self):
        r = C.c_double(0)
        check(lib().das_ksp_get_info(self.handle, None, None, C.byref(r), None))
        return r.value

    def info(self):
        it, r0, r, sec = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
        check(lib().das_ksp_get_info(self.handle, C.byref(it), C.byref(r0), C.byref(r), C.byref(sec)))
        return dict(iters=it.value, res0=r0.value, res=r.value, seconds=sec.value)

    def status(self):
        """How the last solve ended: dict(reason 0 tolerance | 1 gmresMaxIters | 2 breakdown / stagnation, nBreakdown, nRefine,
        sweepGrid, sweepPerXcd)."""
        a = [C.c_int(0) for _ in range(4)]
        check(lib().das_ksp_get_status(self.handle, *[C.byref(x) for x in a]))
        return dict(reason=a[0].value, nBreakdown=a[1].value, nRefine=lib().das_ksp_get_n_refine(self.handle), sweepGrid=a[2].value,
                    sweepPerXcd=a[3].value)

    def blocks(self):
        L = lib()
        nb = check(L.das_ksp_get_n_blocks(self.handle))
        n = self._n
        perm = np.zeros(n, np.int32)
        off = np.zeros(nb + 1, np.int64)
        check(L.das_ksp_get_blocks(self.handle, perm.ctypes.data_as(_capi.c_int_p), off.ctypes.data_as(_capi.c_ll_p)))
        return perm, off

    def pcStructure(self):
        """Node structure of the default ("bilu") preconditioner: dict(nodeUnk[nNodes, 8], bptr, bcol, lvlPtr, natural)."""
        L = lib()
        nN, nB, nLv = C.c_int(0), C.c_longlong(0), C.c_int(0)
        check(L.das_ksp_get_pc_structure_sizes(self.handle, C.byref(nN), C.byref(nB), C.byref(nLv)))
        nu = np.empty(nN.value * 8, np.int32)
        bptr = np.empty(nN.value + 1, np.int64)
        bcol = np.empty(nB.value, np.int32)
        lvl = np.empty(nLv.value + 1, np.int32)
        nat = np.empty(nN.value, np.int32)
        ip = _capi.c_int_p
        check(L.das_ksp_get_pc_structure(self.handle, nu.ctypes.data_as(ip), bptr.ctypes.data_as(_capi.c_ll_p), bcol.ctypes.data_as(ip),
                                         lvl.ctypes.data_as(ip), nat.ctypes.data_as(ip)))
        out = np.empty(nN.value * 8, np.int32)
        check(L.das_ksp_get_pc_node_out(self.handle, out.ctypes.data_as(ip)))
        return dict(nodeUnk=nu.reshape(-1, 8), bptr=bptr, bcol=bcol, lvlPtr=lvl, natural=nat, nodeOut=out.reshape(-1, 8))

    def coarse(self, n_cells):
        """(nAgg, aggOfCell) of the two-level preconditioner's pressure coarse space (nAgg = 0: none)."""
        agg = np.full(n_cells, -1, np.int32)
        nagg = lib().das_ksp_get_coarse(self.handle, agg.ctypes.data_as(_capi.c_int_p))
        return nagg, agg

    def applyPC(self, solver, x):
        y = np.zeros_like(x)
        check(lib().das_ksp_apply_pc(solver._h, self.handle, dptr(np.ascontiguousarray(x)), dptr(y)))
        return y

    def history(self):
        buf = np.zeros(self.getIterationNumber() + 2)
        m = check(lib().das_ksp_get_history(self.handle, dptr(buf), buf.size))
        return buf[:m]

    def basisInfo(self):
        """Krylov basis of the last solve: dict(fp32 (compressed fp32 storage), split (hi + lo floats: inner products read hi only),
        mappedGB, bytesPerVector) - amd.krylovBasisPrecision."""
        f, mb, bv = C.c_int(0), C.c_double(0), C.c_double(0)
        check(lib().das_ksp_get_basis_info(self.handle, C.byref(f), C.byref(mb), C.byref(bv)))
        return dict(fp32=bool(f.value & 1), split=bool(f.value & 2), mappedGB=mb.value / 2**30, bytesPerVector=bv.value)

    def cycleLengths(self):
        """Columns of every closed Arnoldi cycle of the last solve (all but the last equal gmresRestart, DALinearEqn.C:155)."""
        buf = np.zeros(max(1, self.getIterationNumber() + 2), np.int32)
        m = check(lib().das_ksp_get_cycle_lengths(self.handle, buf.ctypes.data_as(_capi.c_int_p), buf.size))
        return buf[:m].copy()

    def destroy(self):
        if self.handle:
            lib().das_ksp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _flatten(prefix, obj, out):
    for k, v in obj.items():
        key = f"{prefix}.{k}" if prefix else k
        if isinstance(v, dict):
            _flatten(key, v, out)
        else:
            out[key] = v


# ----------------------------------------------------------------------------- pyDASolvers
class pyDASolvers:
    """GPU-backed replacement of the reference's ``pyDASolvers`` extension class.

    ``argsAll`` is the reference's command string (``b"DASimpleFoam -python"``, pyDAFoam.py:1425-1430); its first
    token must agree with ``pyOptions["solverName"]``.  The reference reads the mesh and fields from the OpenFOAM
    case in the working directory; here the same data arrives as a :class:`~dafoam_amd.meshgen.FoamCase`
    (``case=`` keyword or ``pyOptions["amdCase"]``).
    """

    def __init__(self, argsAll, pyOptions, case: FoamCase = None):
        if isinstance(argsAll, bytes):
            argsAll = argsAll.decode()
        self._args = argsAll
        case = case if case is not None else pyOptions.get("amdCase")
        if case is None:
            raise ValueError("pyDASolvers: a FoamCase is required (case= or pyOptions['amdCase'])")
        solver = argsAll.split()[0] if argsAll else case.solver_name
        if solver not in ("DASolvers", case.solver_name):
            raise ValueError(f"argsAll solver {solver} != case solver {case.solver_name}")
        self._case = case
        self._cs = CaseStruct(case)
        L = lib()
        self._h = L.das_create(self._cs.byref())
        if not self._h:
            raise _capi.DASError(L.das_last_error().decode())
        self._inited = False
        self._device = int(pyOptions.get("amdDevice", 0)) if isinstance(pyOptions, dict) else 0
        # adjStateOrdering (reference DAIndex.C:188-397,518-661): the library works in "state" ordering; "cell"
        # ordering is provided at this boundary as a permutation of every state-length array
        self._perm = None
        if isinstance(pyOptions, dict) and pyOptions.get("adjStateOrdering", "state") == "cell":
            self._perm = self._cell_ordering_permutation()
        self.updateDAOption(pyOptions)
        self._inputInfo = dict(pyOptions.get("inputInfo") or {}) if isinstance(pyOptions, dict) else {}
        self._primalBC = dict(pyOptions.get("primalBC") or {}) if isinstance(pyOptions, dict) else {}
        self._checkMeshThreshold = dict(pyOptions.get("checkMeshThreshold") or {}) if isinstance(pyOptions, dict) else {}
        self._patchVelocity = [0.0, 0.0]  # DAGlobalVar::patchVelocity = [UMag, AoA(deg)], set by DAInputPatchVelocity::run
        self._flowdir_fns = {}
        self._define_functions(pyOptions.get("function") if isinstance(pyOptions, dict) else None)
        if case.states is not None:  # FoamCase.states is always in "state" ordering (input data)
            check(lib().das_update_of_fields(self._h, dptr(np.ascontiguousarray(case.states, dtype=np.float64))))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().das_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- adjStateOrdering "cell" ----------------------------------------------------------------------
    def _cell_ordering_permutation(self):
        """perm[k] = index in "state" ordering of the k-th entry of the "cell" ordering: per cell its cell states
        (volVector, volScalar, model) followed by the phi of the faces it owns (DAIndex.C:602-651)."""
        m = self._case.mesh
        N, F = m.n_cells, m.n_faces
        layout = {"DASimpleFoam": (1, 3 if getattr(self._case, "has_T", False) else 2), "DARhoSimpleFoam": (1, 3), "DATurboFoam": (1, 3),
                  "DAScalarTransportFoam": (0, 1)}[self._case.solver_name]
        nvec, nscl = layout
        has_phi = self._case.solver_name != "DAScalarTransportFoam"
        owned = [[] for _ in range(N)]
        if has_phi:
            for f in range(F):
                owned[m.owner[f]].append(f)
        perm = []
        for c in range(N):
            for v in range(nvec):
                perm += [3 * N * v + 3 * c + k for k in range(3)]
            for b in range(nscl):
                perm.append(3 * N * nvec + b * N + c)
            perm += [3 * N * nvec + nscl * N + f for f in owned[c]]
        return np.array(perm, dtype=np.int64)

    def _to_state(self, a):
        if self._perm is None:
            return a
        out = np.empty_like(a)
        out[self._perm] = a
        return out

    def _from_state(self, a_state, out):
        if self._perm is None:
            if out is not a_state:
                out[:] = a_state
        else:
            out[:] = a_state[self._perm]

    # -- lifecycle -------------------------------------------------------------------------
    def initSolver(self):
        check(lib().das_init_solver(self._h, self._device))
        self._inited = True

    def updateDAOption(self, pyOptions):
        """pyDASolvers.pyx:355; nested dict -> flattened "a.b" keys (values may be the reference's
        [type, value] pairs produced by pyDAFoam._getDefOptions, pyDAFoam.py:823-844)."""
        flat = {}
        opts = {k: (v[1] if isinstance(v, list) and len(v) == 2 and isinstance(v[0], type) else v) for k, v in pyOptions.items()}
        _flatten("", {k: v for k, v in opts.items() if not k.startswith("amdCase") and k not in ("function", "inputInfo", "outputInfo", "primalBC")}, flat)
        if flat.get("adjStateOrdering") == "cell" and getattr(self, "_perm", None) is not None:
            flat.pop("adjStateOrdering")  # handled at this boundary (permutation); the library stays in "state" ordering
        L = lib()
        for k, v in flat.items():
            kb = k.encode()
            if isinstance(v, bool):
                check(L.das_set_option_int(self._h, kb, int(v)))
            elif isinstance(v, int):
                check(L.das_set_option_int(self._h, kb, v))
            elif isinstance(v, float):
                check(L.das_set_option_double(self._h, kb, v))
            elif isinstance(v, str):
                check(L.das_set_option_str(self._h, kb, v.encode()))
            elif isinstance(v, (list, tuple)) and all(isinstance(x, str) for x in v):
                check(L.das_set_option_str(self._h, kb, ",".join(v).encode()))
            # other option kinds (lists of numbers, ...) are not forwarded

    def printAllOptions(self):
        print("dafoam_amd options are held by the C-ABI; see DAOPTION for defaults")

    # -- sizes -------------------------------------------------------------------------------
    def getNLocalAdjointStates(self):
        return int(lib().das_get_n_local_adjoint_states(self._h))

    def getNLocalAdjointBoundaryStates(self):
        m = self._case.mesh
        nb = m.n_faces - m.n_internal_faces
        return {"DASimpleFoam": 6 if getattr(self._case, "has_T", False) else 5, "DARhoSimpleFoam": 6, "DATurboFoam": 6}.get(self._case.solver_name, 1) * nb

    def getNLocalCells(self):
        return int(lib().das_get_n_local_cells(self._h))

    def getNGlobalCells(self):
        return int(lib().das_get_n_global_cells(self._h))

    def getNLocalPoints(self):
        return int(lib().das_get_n_local_points(self._h))

    def getInputSize(self, inputName, inputType):
        if inputType == "patchVelocity":  # [UMag, AoA(deg)]  (reference DAInputPatchVelocity.H size() = 2)
            return 2
        if inputType == "patchVar":  # reference DAInputPatchVar.H: 1 (scalar) or 3 (vector)
            return 3 if self._input_entry(inputName, inputType)["varType"] == "vector" else 1
        if inputType == "field":  # reference DAInputField.H size(): the selected cells (here: all) x 1 for a scalar field
            self._field_entry(inputName)
            return self._case.mesh.n_cells
        if inputType == "volCoord":  # reference DAInputVolCoord.H size() = nLocalPoints * 3
            return self.getNLocalPoints() * 3
        return check(lib().das_get_input_size(self._h, inputName.encode(), inputType.encode()))

    # -- boundary-value inputs (reference src/adjoint/DAInput/DAInputPatchVelocity.C, DAInputPatchVar.C) ---------
    def _input_entry(self, inputName, inputType):
        if inputName not in self._inputInfo:
            raise _capi.DASError(f"inputInfo has no entry {inputName}")
        e = self._inputInfo[inputName]
        if e.get("type") != inputType:
            raise _capi.DASError(f"inputInfo[{inputName}].type is {e.get('type')}, not {inputType}")
        if inputType == "patchVar" and e.get("varType") not in ("scalar", "vector"):
            raise _capi.DASError("varType not valid")
        return e

    def _field_entry(self, inputName):
        """inputInfo entry of a `field` input (reference DAInputField.C:33-86): fieldName, fieldType "scalar"; cellSetName /
        vector fields are not implemented here."""
        e = self._input_entry(inputName, "field")
        if e.get("fieldType", "scalar") != "scalar":
            raise _capi.DASError("field input: only fieldType scalar is implemented")
        if "cellSetName" in e:
            raise _capi.DASError("field input: cellSetName is not implemented (the field covers all cells)")
        return e

    def getField(self, fieldName):
        out = np.zeros(self._case.mesh.n_cells)
        check(lib().das_get_field(self._h, fieldName.encode(), dptr(out)))
        return out

    def _patch_ids(self, entry):
        names = [p.name for p in self._case.mesh.patches]
        return np.array([names.index(p) for p in entry["patches"]], dtype=np.int32)

    def _patch_input_tangents(self, inputName, inputType, inputs):
        """(field, value[3], [tangent_i[3] for each input component])"""
        e = self._input_entry(inputName, inputType)
        if inputType == "patchVelocity":
            ax = {"x": 0, "y": 1, "z": 2}
            fi, ni = ax[e["flowAxis"]], ax[e["normalAxis"]]
            ids = self._patch_ids(e)
            cur = np.zeros(3)
            check(lib().das_get_patch_value(self._h, int(ids[0]), b"U", dptr(cur)))
            a = float(inputs[1]) * np.pi / 180.0
            val = cur.copy()
            val[fi], val[ni] = inputs[0] * np.cos(a), inputs[0] * np.sin(a)
            t_mag, t_aoa = np.zeros(3), np.zeros(3)
            t_mag[fi], t_mag[ni] = np.cos(a), np.sin(a)
            t_aoa[fi], t_aoa[ni] = -inputs[0] * np.sin(a) * np.pi / 180.0, inputs[0] * np.cos(a) * np.pi / 180.0
            return "U", val, [t_mag, t_aoa]
        name = e["varName"]
        if e["varType"] == "vector":
            return name, np.asarray(inputs, dtype=np.float64).copy(), [np.eye(3)[k].copy() for k in range(3)]
        return name, np.array([float(inputs[0]), 0.0, 0.0]), [np.array([1.0, 0.0, 0.0])]

    def setSolverInput(self, inputName, inputType, inputSize, inputs, seeds=None):
        """DAInput::run (reference pyDASolvers.pyx:164-182): assign the input to the solver.  (Forward-mode seeds
        belong to the reference's ADF build and are ignored here.)"""
        assert len(inputs) == inputSize, "invalid input array size!"
        if inputType == "stateVar":
            return self.updateOFFields(np.ascontiguousarray(inputs, dtype=np.float64))
        if inputType == "volCoord":
            # DAInputVolCoord::run (reference DAInputVolCoord.C:46-70): the input array is the point field; mesh.movePoints
            return self.updateOFMesh(inputs)
        if inputType == "field":
            # DAInputField::run (reference DAInputField.C:88-151): the input array IS the volScalarField (all cells; scalar fields)
            e = self._field_entry(inputName)
            return check(lib().das_set_field(self._h, e["fieldName"].encode(), dptr(np.ascontiguousarray(inputs, dtype=np.float64))))
        if inputType not in ("patchVelocity", "patchVar"):
            raise _capi.DASError(f"inputType not supported on this path: {inputType}")
        field, val, _ = self._patch_input_tangents(inputName, inputType, inputs)
        ids = self._patch_ids(self._inputInfo[inputName])
        check(lib().das_set_patch_value(self._h, ids.ctypes.data_as(_capi.c_int_p), ids.size, field.encode(), dptr(np.ascontiguousarray(val))))
        if inputType == "patchVelocity":
            # DAGlobalVar::patchVelocity: the forces defined parallel / normal to the flow follow the new AoA
            self._patchVelocity = [float(inputs[0]), float(inputs[1])]
            for fname, meta in self._flowdir_fns.items():
                if meta["pv"] == inputName:
                    self._define_flowdir(fname)

    def getOutputSize(self, outputName, outputType):
        return check(lib().das_get_output_size(self._h, outputName.encode(), outputType.encode()))

    # -- states / residuals --------------------------------------------------------------------
    def updateOFFields(self, states):
        assert len(states) == self.getNLocalAdjointStates(), "invalid array size!"
        check(lib().das_update_of_fields(self._h, dptr(np.ascontiguousarray(self._to_state(states)))))

    def getOFFields(self, states):
        assert len(states) == self.getNLocalAdjointStates(), "invalid array size!"
        tmp = np.zeros(len(states)) if self._perm is not None else states
        check(lib().das_get_of_fields(self._h, dptr(tmp)))
        self._from_state(tmp, states)

    def getResiduals(self, residuals):
        assert len(residuals) == self.getNLocalAdjointStates(), "invalid input array size!"
        tmp = np.zeros(len(residuals)) if self._perm is not None else residuals
        check(lib().das_get_residuals(self._h, dptr(tmp)))
        self._from_state(tmp, residuals)

    def calcResiduals(self, isPC, residuals):
        assert len(residuals) == self.getNLocalAdjointStates(), "invalid input array size!"
        tmp = np.zeros(len(residuals)) if self._perm is not None else residuals
        check(lib().das_calc_residuals(self._h, int(isPC), dptr(tmp)))
        self._from_state(tmp, residuals)

    def updateStateBoundaryConditions(self):
        # boundary values are recomputed inline by every kernel from the state vector: nothing to do
        return None

    def getOFMeshPoints(self, points):
        assert len(points) == self.getNLocalPoints() * 3, "invalid array size!"
        check(lib().das_get_of_mesh_points(self._h, dptr(points)))

    def updateOFMesh(self, vol_coords):
        """pyDASolvers.pyx:297-300 (PYDAFOAM.setVolCoords, pyDAFoam.py:2111-2117): new point coordinates; the library
        recomputes the fvMesh metrics (the wall distance stays frozen).  Jacobians assembled earlier describe the old
        mesh and have to be rebuilt by the caller, as in the reference."""
        assert len(vol_coords) == self.getNLocalPoints() * 3, "invalid array size!"
        check(lib().das_update_of_mesh(self._h, dptr(np.ascontiguousarray(vol_coords, dtype=np.float64))))

    def calcVolCoordDirectionalProduct(self, dX, outputName, outputType, seeds, eps=1e-6):
        """seeds^T (dOutput/dX . dX) for ONE direction dX of the mesh points (e.g. the volume-mesh sensitivity of one
        FFD design variable), by a central difference of the geometry: two metric updates + two residual (or objective)
        evaluations on the device.  The reference returns the full product vector of calcJacTVecProduct(volCoord -> ...)
        from one reverse sweep of its AD tape (DASolver.C:1690-1839); without a tape the per-design-variable form is the
        one that maps to forward evaluations - the design variables of a shape optimisation are few.
        eps is relative to max|dX| (step = eps * characteristic point spacing / max|dX| is the caller's choice: here the
        points move by eps * dX)."""
        n3 = self.getNLocalPoints() * 3
        assert len(dX) == n3, "invalid array size!"
        X0 = np.zeros(n3)
        self.getOFMeshPoints(X0)
        dX = np.asarray(dX, dtype=np.float64)
        vals = []
        try:
            for sgn in (1.0, -1.0):
                self.updateOFMesh(X0 + sgn * eps * dX)
                if outputType == "residual":
                    R = np.zeros(self.getNLocalAdjointStates())
                    check(lib().das_get_residuals(self._h, dptr(R)))
                    vals.append(float(np.dot(self._to_state(np.asarray(seeds, dtype=np.float64)), R)))
                elif outputType == "function":
                    vals.append(float(seeds[0]) * self.calcFunction(outputName))
                else:
                    raise _capi.DASError(f"outputType not supported on this path: {outputType}")
        finally:
            self.updateOFMesh(X0)
        return (vals[0] - vals[1]) / (2.0 * eps)

    def pointInfluence(self):
        """Host-side structure of the volCoord product (no GPU needed): dict(colors[nPoints], nColors, ptr, cells, steps) - the
        cells whose residual rows can feel a point (CSR), a colouring of the points with pairwise disjoint sets, the
        central-difference step of every point."""
        nc, ne = C.c_int(0), C.c_longlong(0)
        check(lib().das_point_influence_build(self._h, C.byref(nc), C.byref(ne)))
        P = self.getNLocalPoints()
        col = np.zeros(P, np.int32)
        ptr = np.zeros(P + 1, np.int64)
        cells = np.zeros(ne.value, np.int32)
        h = np.zeros(P)
        check(lib().das_point_influence_get(self._h, col.ctypes.data_as(_capi.c_int_p), ptr.ctypes.data_as(_capi.c_ll_p),
                                            cells.ctypes.data_as(_capi.c_int_p), dptr(h)))
        return dict(colors=col, nColors=nc.value, ptr=ptr, cells=cells, steps=h)

    def deviceGeometry(self, points):
        """The metrics the DEVICE passes of the volCoord product compute for `points` (test aid): (fg[nF, 12], cg[nC, 5]) in the
        record layout of csrc/das_common.hpp (Sf, magSf, w, nod, corr, Cf | C, V, y)."""
        m = self._case.mesh
        fg = np.zeros(12 * m.n_faces)
        cg = np.zeros(5 * m.n_cells)
        check(lib().das_debug_device_geometry(self._h, dptr(np.ascontiguousarray(points, dtype=np.float64)), dptr(fg), dptr(cg)))
        return fg.reshape(-1, 12), cg.reshape(-1, 5)

    def geometry(self):
        """fvMesh metrics computed by the library (host side)."""
        m = self._case.mesh
        F, Fi, N = m.n_faces, m.n_internal_faces, m.n_cells
        out = dict(Sf=np.zeros(3 * F), Cf=np.zeros(3 * F), C=np.zeros(3 * N), V=np.zeros(N), w=np.zeros(Fi),
                   nonOrthDeltaCoeffs=np.zeros(Fi), nonOrthCorr=np.zeros(3 * Fi), bDeltaCoeffs=np.zeros(F - Fi))
        check(lib().das_get_geometry(self._h, *[dptr(out[k]) for k in
                                                ("Sf", "Cf", "C", "V", "w", "nonOrthDeltaCoeffs", "nonOrthCorr", "bDeltaCoeffs")]))
        return out

    # -- colouring / Jacobians -------------------------------------------------------------------
    def runColoring(self, cacheDir=None, nProcs=1, rank=0):
        """DASolver::runColoring (DASolver.C:708-745).  With cacheDir the colouring is kept like the reference keeps
        it, as a PETSc binary Vec `dRdWColoring_<nProcs>.bin` (`DAJacCon.C:1886-2019`; one file per rank here, suffix
        `_<rank>` when nProcs > 1): read + validated if present and of the right size, computed and written otherwise.
        A file that does not validate (different mesh) is recomputed, not trusted."""
        if cacheDir is None:
            return check(lib().das_run_coloring(self._h))
        import os

        from . import petsc_io

        name = f"dRdWColoring_{nProcs}" + (f"_{rank}" if nProcs > 1 else "") + ".bin"
        path = os.path.join(cacheDir, name)
        n = self.getNLocalAdjointStates()
        if os.path.exists(path):
            try:
                col = petsc_io.read_vec(path)
                if col.size == n and np.all(col == np.round(col)):
                    ci = np.ascontiguousarray(col, dtype=np.int32)
                    check(lib().das_set_coloring(self._h, ci.ctypes.data_as(_capi.c_int_p)))
                    return
            except (_capi.DASError, ValueError, OSError):
                pass  # stale / foreign cache: fall through and recompute
        check(lib().das_run_coloring(self._h))
        col, _ = self.getColoring()
        tmp = path + f".tmp{os.getpid()}"
        petsc_io.write_vec(tmp, col.astype(np.float64))
        os.replace(tmp, path)

    def getColoring(self):
        n = self.getNLocalAdjointStates()
        col = np.zeros(n, np.int32)
        check(lib().das_get_colors(self._h, 0, col.ctypes.data_as(_capi.c_int_p)))
        return col, check(lib().das_get_n_colors(self._h, 0))

    def getConnectivity(self, isPC=0):
        import scipy.sparse as sp

        n = self.getNLocalAdjointStates()
        nnz = check(lib().das_get_con_nnz(self._h, int(isPC)))
        rp = np.zeros(n + 1, np.int64)
        ci = np.zeros(nnz, np.int32)
        check(lib().das_get_con(self._h, int(isPC), rp.ctypes.data_as(_capi.c_ll_p), ci.ctypes.data_as(_capi.c_int_p)))
        return sp.csr_matrix((np.ones(nnz, np.int8), ci, rp), shape=(n, n))

    def pcStructure(self):
        """Node structure the default ("bilu") preconditioner would be factorised on, built on the host (no GPU needed):
        dict(nodeUnk[nNodes, 8], bptr, bcol, lvlPtr, natural, reach)."""
        L = lib()
        nN, nB, nLv, reach = C.c_int(0), C.c_longlong(0), C.c_int(0), C.c_int(0)
        check(L.das_pc_structure_build(self._h, C.byref(nN), C.byref(nB), C.byref(nLv), C.byref(reach)))
        nu = np.empty(nN.value * 8, np.int32)
        bptr = np.empty(nN.value + 1, np.int64)
        bcol = np.empty(nB.value, np.int32)
        lvl = np.empty(nLv.value + 1, np.int32)
        nat = np.empty(nN.value, np.int32)
        ip = _capi.c_int_p
        check(L.das_pc_structure_get(self._h, nu.ctypes.data_as(ip), bptr.ctypes.data_as(_capi.c_ll_p), bcol.ctypes.data_as(ip),
                                     lvl.ctypes.data_as(ip), nat.ctypes.data_as(ip)))
        return dict(nodeUnk=nu.reshape(-1, 8), bptr=bptr, bcol=bcol, lvlPtr=lvl, natural=nat, reach=reach.value)

    def calcdRdWT(self, isPC, dRdWT: Mat, mode=None):
        """pyDASolvers.pyx:237.  mode None: the reference's behaviour for the PC (coloured FD) and exact
        dual-number assembly for isPC=0."""
        if mode is None:
            mode = 0 if isPC else 1
        h = C.c_void_p()
        check(lib().das_calc_drdwt(self._h, int(isPC), int(mode), C.byref(h)))
        dRdWT._set(h)

    def initializedRdWTMatrixFree(self):
        check(lib().das_initialize_drdwt_matrix_free(self._h))

    def destroydRdWTMatrixFree(self):
        check(lib().das_destroy_drdwt_matrix_free(self._h))

    def calcJacTVecProduct(self, inputName, inputType, inputs, outputName, outputType, seeds, product):
        inputSize = self.getInputSize(inputName, inputType)
        outputSize = self.getOutputSize(outputName, outputType)
        assert len(inputs) == inputSize, "invalid input array size!"
        assert len(seeds) == outputSize, "invalid seed array size!"
        assert len(product) == inputSize, "invalid product array size!"
        seeds_s = np.ascontiguousarray(self._to_state(seeds)) if outputType == "residual" else seeds
        if inputType == "volCoord":
            # run(input) = movePoints, then the full product over all points from coloured central differences on the device
            X0 = np.zeros(inputSize)
            self.getOFMeshPoints(X0)
            if not np.array_equal(X0, np.asarray(inputs, dtype=np.float64)):
                self.updateOFMesh(inputs)
            info = np.zeros(4)
            check(lib().das_calc_dvolcoord_product(self._h, outputName.encode(), outputType.encode(),
                                                   dptr(np.ascontiguousarray(seeds_s, dtype=np.float64)), dptr(product), dptr(info)))
            self._volCoordInfo = dict(colors=int(info[0]), passes=int(info[1]), seconds=float(info[2]), build_seconds=float(info[3]))
            return
        if inputType == "field":
            # run(input), then ONE forward-mode pass with a unit tangent on every cell (das_calc_dfield_product)
            self.setSolverInput(inputName, inputType, inputSize, inputs)
            e = self._field_entry(inputName)
            check(lib().das_calc_dfield_product(self._h, e["fieldName"].encode(), outputName.encode(), outputType.encode(),
                                                dptr(np.ascontiguousarray(seeds_s, dtype=np.float64)), dptr(product)))
            return
        if inputType in ("patchVelocity", "patchVar"):
            # run(input), then one forward-mode pass per input component
            self.setSolverInput(inputName, inputType, inputSize, inputs)
            field, _, tangents = self._patch_input_tangents(inputName, inputType, inputs)
            ids = self._patch_ids(self._inputInfo[inputName])
            out = C.c_double(0.0)
            for i, t in enumerate(tangents):
                check(lib().das_calc_dbc_product(self._h, ids.ctypes.data_as(_capi.c_int_p), ids.size, field.encode(), dptr(np.ascontiguousarray(t)),
                                                 outputName.encode(), outputType.encode(), dptr(np.ascontiguousarray(seeds_s, dtype=np.float64)), C.byref(out)))
                product[i] = out.value
            meta = self._flowdir_fns.get(outputName) if outputType == "function" else None
            if meta is not None and inputType == "patchVelocity" and meta["pv"] == inputName:
                # the force direction itself depends on the AoA: F is linear in it, so dF/dAoA|dir = F(d') with the
                # unit vector of d' = dd/dAoA, times |d'| = pi/180
                dprime = self._flow_direction(meta, deriv=True)
                tmp = outputName + "__ddir"
                self._define_flowdir(outputName, name=tmp, direction=dprime * (180.0 / np.pi))
                product[1] += float(seeds[0]) * self.calcFunction(tmp) * (np.pi / 180.0)
            return
        tmp = np.zeros(len(product)) if self._perm is not None else product
        check(lib().das_calc_jac_t_vec_product(
            self._h, inputName.encode(), inputType.encode(), dptr(np.ascontiguousarray(self._to_state(inputs))), outputName.encode(),
            outputType.encode(), dptr(seeds_s), dptr(tmp)))
        self._from_state(tmp, product)

    def calcJacVecProduct(self, v, product):
        """Forward-mode dR/dW (s o v) at the current states: one dual-number residual pass (no colouring, no matrix)."""
        n = self.getNLocalAdjointStates()
        assert len(v) == n, "invalid input array size!"
        assert len(product) == n, "invalid product array size!"
        tmp = np.zeros(n) if self._perm is not None else product
        check(lib().das_calc_jac_vec_product(self._h, dptr(np.ascontiguousarray(self._to_state(v))), dptr(tmp)))
        self._from_state(tmp, product)

    def calcdRdWOldTPsiAD(self, oldTimeLevel, psi, dRdWOldTPsi):
        assert len(psi) == self.getNLocalAdjointStates(), "invalid input array size!"
        assert len(dRdWOldTPsi) == self.getNLocalAdjointStates(), "invalid seed array size!"
        tmp = np.zeros(len(psi)) if self._perm is not None else dRdWOldTPsi
        check(lib().das_calc_drdwold_t_psi(self._h, int(oldTimeLevel), dptr(np.ascontiguousarray(self._to_state(psi))), dptr(tmp)))
        self._from_state(tmp, dRdWOldTPsi)

    def setOldTimeFields(self, phi=None, T_old=None):
        check(lib().das_set_old_time_fields(self._h, dptr(np.ascontiguousarray(phi)) if phi is not None else None,
                                            dptr(np.ascontiguousarray(T_old)) if T_old is not None else None))

    # -- objective functions ------------------------------------------------------------------------
    def _define_functions(self, functions):
        """"function" option dict (reference pyDAFoam.py DAOPTION.function): the patch-integral types force (directionMode
        fixedDirection), moment, massFlowRate, totalPressure and totalTemperatureRatio are on this path."""
        names = [p.name for p in self._case.mesh.patches]
        L = lib()
        for fname, fd in (functions or {}).items():
            ftype = fd.get("type")
            if ftype not in ("force", "moment", "massFlowRate", "totalPressure", "totalTemperatureRatio"):
                raise NotImplementedError(f"function type {ftype} is outside the GPU hot path")
            ids = np.array([names.index(p) for p in fd["patches"]], dtype=np.int32)
            dmode = fd.get("directionMode", "fixedDirection") if ftype == "force" else None
            if dmode in ("parallelToFlow", "normalToFlow"):
                # the direction follows the angle of attack of a patchVelocity input (DAFunctionForce.C:45-61,92-113)
                pv = fd["patchVelocityInputName"]
                e = self._input_entry(pv, "patchVelocity")
                ax = {"x": 0, "y": 1, "z": 2}
                self._flowdir_fns[fname] = dict(mode=dmode, fi=ax[e["flowAxis"]], ni=ax[e["normalAxis"]], ids=ids, scale=float(fd.get("scale", 1.0)), pv=pv)
                self._define_flowdir(fname)
                continue
            if ftype == "force" and dmode != "fixedDirection":
                raise _capi.DASError(f"directionMode for {fname} not valid!Options: fixedDirection, parallelToFlow, normalToFlow.")
            grp = None
            gamma = 0.0
            vecA = vecB = None
            if ftype == "force":
                vecA = np.ascontiguousarray(fd["direction"], dtype=np.float64)
            elif ftype == "moment":
                vecA = np.ascontiguousarray(fd["axis"], dtype=np.float64)
                vecB = np.ascontiguousarray(fd["center"], dtype=np.float64)
            elif ftype == "totalTemperatureRatio":
                inl, out = fd["inletPatches"], fd["outletPatches"]
                for p in fd["patches"]:
                    if p not in inl and p not in out:
                        raise _capi.DASError("inlet/outletPatches names are not in patches")
                grp = np.array([1 if p in out else 0 for p in fd["patches"]], dtype=np.int32)
                gamma = float(self._case.thermo.get("gamma", 1.4))
            check(L.das_define_face_function(
                self._h, fname.encode(), ftype.encode(), ids.ctypes.data_as(_capi.c_int_p),
                grp.ctypes.data_as(_capi.c_int_p) if grp is not None else None, ids.size,
                dptr(vecA) if vecA is not None else None, dptr(vecB) if vecB is not None else None, float(fd.get("scale", 1.0)), gamma))

    def _flow_direction(self, meta, deriv=False):
        """force direction of parallelToFlow / normalToFlow at the current AoA (deriv: d/dAoA[deg])"""
        a = self._patchVelocity[1] * np.pi / 180.0
        d = np.zeros(3)
        ca, sa = (np.cos(a), np.sin(a)) if not deriv else (-np.sin(a) * np.pi / 180.0, np.cos(a) * np.pi / 180.0)
        if meta["mode"] == "parallelToFlow":
            d[meta["fi"]], d[meta["ni"]] = ca, sa
        else:
            d[meta["fi"]], d[meta["ni"]] = -sa, ca
        return d

    def _define_flowdir(self, fname, name=None, direction=None):
        meta = self._flowdir_fns[fname]
        d = np.ascontiguousarray(self._flow_direction(meta) if direction is None else direction, dtype=np.float64)
        ids = meta["ids"]
        check(lib().das_define_face_function(self._h, (name or fname).encode(), b"force", ids.ctypes.data_as(_capi.c_int_p), None, ids.size,
                                             dptr(d), None, meta["scale"], 0.0))

    def calcFunction(self, functionName):
        v = C.c_double(0.0)
        check(lib().das_calc_function(self._h, functionName.encode(), C.byref(v)))
        return v.value

    # -- Krylov ----------------------------------------------------------------------------------
    def createMLRKSPMatrixFree(self, jacPCMat: Mat, myKSP: KSP):
        h = C.c_void_p()
        check(lib().das_create_ml_rksp_matrix_free(self._h, jacPCMat.handle, C.byref(h)))
        myKSP.destroy()
        myKSP.handle = h
        myKSP._pc = jacPCMat  # keep the PC matrix alive as long as the KSP
        myKSP._n = self.getNLocalAdjointStates()

    def updateKSPPCMat(self, PCMat: Mat, myKSP: KSP):
        self.createMLRKSPMatrixFree(PCMat, myKSP)

    def solveLinearEqn(self, myKSP: KSP, rhsVec: Vec, solVec: Vec):
        L = lib()
        for k, v in myKSP._tols.items():
            (L.das_set_option_int if isinstance(v, int) else L.das_set_option_double)(self._h, k.encode(), v)
        if self._perm is None:
            return check(L.das_solve_linear_eqn(self._h, myKSP.handle, dptr(rhsVec.array), dptr(solVec.array)))
        rhs, sol = np.ascontiguousarray(self._to_state(rhsVec.array)), np.ascontiguousarray(self._to_state(solVec.array))
        rc = check(L.das_solve_linear_eqn(self._h, myKSP.handle, dptr(rhs), dptr(sol)))
        self._from_state(sol, solVec.array)
        return rc

    def solvePrimal(self, maxSteps=80, relTol=1e-8, absTol=0.0):
        """solvePrimal (reference pyDASolvers.pyx solvePrimal -> DASimpleFoam::solvePrimal, DASimpleFoam.C:123-185): converge
        the residuals from the current states.  Returns (fail, info) with info = dict(steps, linearIterations, res0, res,
        history); the converged states are read back with getOFFields / getStates."""
        info4 = np.zeros(4)
        hist = np.zeros(int(maxSteps) + 2)
        rc = check(lib().das_solve_primal(self._h, int(maxSteps), float(relTol), float(absTol), dptr(info4), dptr(hist), hist.size))
        nst = int(info4[0])
        return rc, dict(steps=nst, linearIterations=int(info4[1]), res0=float(info4[2]), res=float(info4[3]), history=hist[: nst + 1].copy())

    def simpleIteration(self, nSweeps=1, alphaP=0.3, linTol=1e-12, maxLinIters=20000):
        """nSweeps iterations of the reference's own primal loop, SIMPLE (DASimpleFoam::solvePrimal, DASimpleFoam.C:123-185: UEqnSimple.H,
        pEqnSimple.H, DASpalartAllmaras::correct), on the device from the current states (das_simple_iteration); alphaP = the explicit
        pressure relaxation of fvSolution, linTol / maxLinIters = the inner solvers' settings.  Returns the inner iteration counts of the
        last sweep; the new states are read with getOFFields / getStates."""
        info = np.zeros(3)
        check(lib().das_simple_iteration(self._h, int(nSweeps), float(alphaP), float(linTol), int(maxLinIters), dptr(info)))
        return dict(U=int(info[0]), p=int(info[1]), nuTilda=int(info[2]))

    def solveLinearEqnBlock(self, myKSP: KSP, rhs, sol):
        """Several adjoint systems with the same operator through ONE block GMRES (the reference loops solveLinearEqn over
        the objective functions, mphys_dafoam.py:478-481).  rhs, sol: (n, s) arrays, s <= 8; returns (fail, res0[s], res[s])."""
        L = lib()
        for k, v in myKSP._tols.items():
            (L.das_set_option_int if isinstance(v, int) else L.das_set_option_double)(self._h, k.encode(), v)
        rhs = np.asarray(rhs, dtype=np.float64)
        n, s = rhs.shape
        if self._perm is not None:
            rhs = np.stack([self._to_state(rhs[:, r]) for r in range(s)], axis=1)
        B = np.asfortranarray(rhs)
        X = np.zeros((n, s), order="F")
        r0, r1 = np.zeros(s), np.zeros(s)
        rc = check(L.das_solve_linear_eqn_block(self._h, myKSP.handle, int(s), B.ctypes.data_as(_capi.c_double_p), X.ctypes.data_as(_capi.c_double_p),
                                                dptr(r0), dptr(r1)))
        for r in range(s):
            if self._perm is None:
                sol[:, r] = X[:, r]
            else:
                self._from_state(np.ascontiguousarray(X[:, r]), sol[:, r])
        return rc, r0, r1

    # -- timing ----------------------------------------------------------------------------------
    def _state_blocks(self):
        """[(name, kind, offset, size)] of the DAIndex "state" ordering (reference DAIndex.C:188-258)."""
        m = self._case.mesh
        N, F = m.n_cells, m.n_faces
        names = {"DASimpleFoam": ["U", "p"] + (["T"] if getattr(self._case, "has_T", False) else []) + ["nuTilda", "phi"],
                 "DARhoSimpleFoam": ["U", "p", "T", "nuTilda", "phi"], "DATurboFoam": ["U", "p", "T", "nuTilda", "phi"],
                 "DAScalarTransportFoam": ["T"]}[self._case.solver_name]
        out, off = [], 0
        for nm in names:
            kind = "vec" if nm == "U" else ("face" if nm == "phi" else "scl")
            size = {"vec": 3 * N, "scl": N, "face": F}[kind]
            out.append((nm, kind, off, size))
            off += size
        return out

    def calcPrimalResidualStatistics(self, mode, writeRes=0):
        """DASolver::calcPrimalResidualStatistics (reference DASolver.C:745-946): norm2 / mean / max of every residual
        block at the current states; printed for mode "print", only computed for "calc".  Returns the numbers (the
        reference keeps them internal): {resName: {"norm2", "mean", "max"}, "totalResNorm2": ...}."""
        if mode not in ("print", "calc"):
            raise _capi.DASError("mode not valid")
        n = self.getNLocalAdjointStates()
        R = np.zeros(n)
        check(lib().das_get_residuals(self._h, dptr(R)))
        stats, tot = {}, 0.0
        if mode == "print":
            print("Printing Primal Residual Statistics.")
        for nm, kind, off, size in self._state_blocks():
            blk = R[off : off + size]
            if kind == "vec":
                blk = blk.reshape(-1, 3)
                st = {"norm2": np.sqrt((blk * blk).sum(0)), "mean": np.abs(blk).mean(0), "max": np.abs(blk).max(0)}
            else:
                st = {"norm2": float(np.sqrt((blk * blk).sum())), "mean": float(np.abs(blk).mean()), "max": float(np.abs(blk).max())}
            tot += float((blk * blk).sum())
            stats[nm + "Res"] = st
            if mode == "print":
                print(f"{nm} Residual Norm2: {st['norm2']}\n{nm} Residual Mean: {st['mean']}\n{nm} Residual Max: {st['max']}")
        stats["totalResNorm2"] = float(np.sqrt(tot))
        if mode == "print":
            print(f"Total Residual Norm2: {stats['totalResNorm2']}")
        return stats

    def writeAdjointFields(self, function, writeTime, psi, caseDir="."):
        """DASolver::writeAdjointFields (reference DASolver.C:4055-4160, pyDASolvers.pyx:470): psi as OpenFOAM fields
        adjoint_<function>_<state> under <caseDir>/<writeTime>/ (the reference writes into its case directory)."""
        from . import foam_io

        assert len(psi) == self.getNLocalAdjointStates(), "invalid array size!"
        psi_s = np.ascontiguousarray(self._to_state(np.asarray(psi, dtype=np.float64)))
        return foam_io.write_adjoint_fields(caseDir, self._case, function, writeTime, psi_s, self._state_blocks())

    # -- the rest of the reference's pyDASolvers surface that a runScript / mphys_dafoam.py touches around the hot path ------
    caseDir = "."  # where the IO methods read / write (the reference runs inside its case directory)

    def calcOutput(self, outputName, outputType, output):
        """pyDASolvers.pyx:204-206 -> DAOutput::run: the function value (1 entry) or the residual vector (state ordering of the
        caller: adjStateOrdering) at the current states."""
        assert len(output) == self.getOutputSize(outputName, outputType), "invalid array size!"
        if outputType == "function":
            output[0] = self.calcFunction(outputName)
        elif outputType == "residual":
            R = np.zeros(self.getNLocalAdjointStates())
            self.getResiduals(R)
            output[:] = R
        else:
            raise _capi.DASError(f"outputType not supported on this path: {outputType}")

    def hasVolCoordInput(self):
        """DASolver::hasVolCoordInput (DASolver.C:4149-4164): 1 if an inputInfo entry has type volCoord."""
        return int(any(isinstance(e, dict) and e.get("type") == "volCoord" for e in self._inputInfo.values()))

    def getInputDistributed(self, inputName, inputType):
        """DAInput*::distributed(): 1 for inputs that are partitioned over the ranks (DAInputStateVar.H:57, DAInputVolCoord.H:57; a
        field input says so itself, DAInputField.H:110-114), 0 for the global ones (patchVelocity, patchVar)."""
        if inputType in ("stateVar", "volCoord"):
            return 1
        if inputType == "field":
            return int(self._field_entry(inputName).get("distributed", 1))
        if inputType in ("patchVelocity", "patchVar"):
            return 0
        raise _capi.DASError(f"inputType not supported on this path: {inputType}")

    def getOutputDistributed(self, outputName, outputType):
        """DAOutputResidual.H:59 (1), DAOutputFunction.H:76 (0)."""
        if outputType == "residual":
            return 1
        if outputType == "function":
            return 0
        raise _capi.DASError(f"outputType not supported on this path: {outputType}")

    def getOFField(self, fieldName, fieldType, field):
        """pyDASolvers.pyx:283-289: the internal field `fieldName` (a state of this solver) of the current states, cell by cell
        (vectors interleaved xyz)."""
        N = self.getNLocalCells()
        assert len(field) == (3 * N if fieldType == "vector" else N), "invalid array size!"
        W = np.zeros(self.getNLocalAdjointStates())
        check(lib().das_get_of_fields(self._h, dptr(W)))
        for nm, kind, off, size in self._state_blocks():
            if nm == fieldName and kind != "face":
                if (kind == "vec") != (fieldType == "vector"):
                    raise _capi.DASError(f"fieldType {fieldType} does not match field {fieldName}")
                field[:] = W[off : off + size]
                return
        raise _capi.DASError(f"field {fieldName} is not a cell-centred state of {self._case.solver_name}")

    def getOFFieldGlobal(self, fieldName, fieldType, field):
        """pyDASolvers.pyx:291-295 (the field gathered over all ranks): one domain here - the sharded wrapper owns the gather."""
        if self.getNGlobalCells() != self.getNLocalCells():
            raise _capi.DASError("getOFFieldGlobal on a sharded solver: gather through dafoam_amd.distributed")
        return self.getOFField(fieldName, fieldType, field)

    def getGlobalXvIndex(self, pointI, coordI):
        """DAIndex::getGlobalXvIndex (DAIndex.C:757-775): 3 * (global point) + coordinate; one domain: the local numbering."""
        assert 0 <= pointI < self.getNLocalPoints() and 0 <= coordI < 3
        return 3 * int(pointI) + int(coordI)

    def checkMesh(self):
        """DASolver::checkMesh -> DACheckMesh::run (DACheckMesh.C:49-77: OpenFOAM's checkGeometry with the thresholds of the
        checkMeshThreshold option): 1 = mesh quality passes.  Checked here on the library's metrics: cell volumes > 0, face
        orientation d.Sf > 0 (at most maxIncorrectlyOrientedFaces violations), non-orthogonality angle, skewness (distance of
        the face centre from the point where the line of centres meets the face, over |d|, OpenFOAM primitiveMeshTools::
        faceSkewness) and cell aspect ratio (primitiveMeshTools::cellClosedness) below the thresholds."""
        th = dict(maxAspectRatio=1000.0, maxNonOrth=70.0, maxSkewness=4.0, maxIncorrectlyOrientedFaces=0)
        th.update({k: (v[1] if isinstance(v, list) else v) for k, v in self._checkMeshThreshold.items()})
        g = self.geometry()
        m = self._case.mesh
        nIF, N = m.n_internal_faces, m.n_cells
        own, nei = np.asarray(m.owner), np.asarray(m.neighbour)
        Sf, Cf, Cc, V = g["Sf"].reshape(-1, 3), g["Cf"].reshape(-1, 3), g["C"].reshape(-1, 3), g["V"]
        self.meshQuality = q = {}
        q["minVolume"] = float(V.min())
        d = Cc[nei] - Cc[own[:nIF]]
        dn = np.einsum("ij,ij->i", d, Sf[:nIF])
        magd, magS = np.linalg.norm(d, axis=1), np.linalg.norm(Sf[:nIF], axis=1)
        q["incorrectlyOrientedFaces"] = int(np.count_nonzero(dn <= 0))
        cosang = np.clip(dn / np.maximum(magd * magS, 1e-300), -1.0, 1.0)
        q["maxNonOrth"] = float(np.degrees(np.arccos(cosang)).max()) if nIF else 0.0
        t = np.einsum("ij,ij->i", Cf[:nIF] - Cc[own[:nIF]], Sf[:nIF]) / np.where(dn != 0, dn, 1.0)
        sk = np.linalg.norm(Cf[:nIF] - (Cc[own[:nIF]] + t[:, None] * d), axis=1) / np.maximum(magd, 1e-300)
        q["maxSkewness"] = float(sk.max()) if nIF else 0.0
        sumMag = np.zeros((N, 3))
        absS = np.abs(Sf)
        for k in range(3):
            sumMag[:, k] = np.bincount(own, absS[:, k], N) + np.bincount(nei, absS[:nIF, k], N)
        lo = np.maximum(sumMag.min(axis=1), 1e-300)
        q["maxAspectRatio"] = float(np.maximum(sumMag.max(axis=1) / lo, sumMag.sum(axis=1) / 6.0 / np.maximum(V, 1e-300) ** (2.0 / 3.0)).max())
        ok = (q["minVolume"] > 0 and q["incorrectlyOrientedFaces"] <= th["maxIncorrectlyOrientedFaces"] and q["maxNonOrth"] <= th["maxNonOrth"]
              and q["maxSkewness"] <= th["maxSkewness"] and q["maxAspectRatio"] <= th["maxAspectRatio"])
        return int(ok)

    # time bookkeeping of the steady solvers (pyDASolvers.pyx:358,400-410,464): iterations play the role of time
    def setTime(self, time, timeIndex):
        self._time, self._timeIndex = float(time), int(timeIndex)

    def getDeltaT(self):
        return float(getattr(self._case, "deltaT", 1.0))

    def getEndTime(self):
        return float(getattr(self, "_endTime", getattr(self, "_time", 0.0)))

    def getLatestTime(self):
        return float(getattr(self, "_time", 0.0))

    def getPrevPrimalSolTime(self):
        return float(getattr(self, "_prevPrimalSolTime", getattr(self, "_time", 0.0)))

    def getDdtSchemeOrder(self):
        """1 (Euler) for the unsteady scalar transport residual, the steady solvers have no time derivative (DASolver::getDdtSchemeOrder)."""
        return 1

    def getdFScaling(self, functionName, timeIdx=-1):
        """DASolver::getdFScaling: the weight of a time instance in a time-averaged objective; 1 for steady solvers."""
        return 1.0

    def getTimeOpFuncVal(self, functionName):
        """DASolver::getTimeOpFuncVal: the time-operated (final / averaged) objective; steady solvers: the value at the current states."""
        return self.calcFunction(functionName)

    def updateBoundaryConditions(self, fieldName, fieldType):
        """pyDASolvers.pyx:364: boundary values are re-evaluated inline by every kernel from the states - nothing is stored."""
        return None

    def setPrimalBoundaryConditions(self, printInfo=1):
        """DASolver::setPrimalBoundaryConditions (DASolver.C:3790-4030): the primalBC option {name: {variable, patches, value}}
        written into the patch table (fixedValue / inletOutlet patches)."""
        for name, e in (self._primalBC or {}).items():
            if not isinstance(e, dict) or "variable" not in e:
                continue
            ids = np.ascontiguousarray(self._patch_ids(e), dtype=np.int32)
            val = np.ascontiguousarray(np.atleast_1d(np.asarray(e["value"], dtype=np.float64)))
            check(lib().das_set_patch_value(self._h, ids.ctypes.data_as(_capi.c_int_p), ids.size, str(e["variable"]).encode(), dptr(val)))
            if printInfo:
                print(f"primalBC {name}: {e['variable']} on {e['patches']} = {val.tolist()}")

    # mesh / state files (pyDASolvers.pyx:361,382-395 -> DASolver::readMeshPoints / writeMeshPoints / readStateVars)
    def _points_path(self, timeVal):
        import os

        return os.path.join(self.caseDir, ("%g" % timeVal) if timeVal is not None else "constant", "polyMesh", "points")

    def writeMeshPoints(self, points, timeVal):
        from . import foam_io

        assert len(points) == self.getNLocalPoints() * 3, "invalid array size!"
        foam_io.write_points(self._points_path(timeVal), np.asarray(points, dtype=np.float64).reshape(-1, 3))

    def readMeshPoints(self, timeVal):
        from . import foam_io

        self.updateOFMesh(np.ascontiguousarray(foam_io.read_points(self._points_path(timeVal)).ravel()))

    def writeCurrentMeshPointsToConstant(self):
        X = np.zeros(self.getNLocalPoints() * 3)
        self.getOFMeshPoints(X)
        self.writeMeshPoints(X, None)

    def writeFailedMesh(self):
        """DASolver::writeFailedMesh: the current points under the time directory 9999 for inspection (pyDAFoam.py:1240-1256)."""
        X = np.zeros(self.getNLocalPoints() * 3)
        self.getOFMeshPoints(X)
        self.writeMeshPoints(X, 9999)

    def readStateVars(self, timeVal, timeLevel=0):
        """DASolver::readStateVars (DASolver.C:3478-3600): the state fields of time directory timeVal -> the solver (timeLevel 0)."""
        import os

        from . import foam_io

        if timeLevel != 0:
            raise _capi.DASError("readStateVars: old-time levels are set through setOldTimeFields on this path")
        m = self._case.mesh
        N = m.n_cells
        tdir = os.path.join(self.caseDir, "%g" % timeVal)
        W = np.zeros(self.getNLocalAdjointStates())
        check(lib().das_get_of_fields(self._h, dptr(W)))
        for nm, kind, off, size in self._state_blocks():
            path = os.path.join(tdir, nm)
            if kind == "face" or not os.path.exists(path):
                continue  # phi is a derived field in a time directory written by a primal (kept as it is here)
            vals, _ = foam_io.read_field(path, N, 3 if kind == "vec" else 1)
            W[off : off + size] = np.asarray(vals, dtype=np.float64).ravel()
        check(lib().das_update_of_fields(self._h, dptr(W)))

    def getElapsedClockTime(self):
        return lib().das_get_elapsed_clock_time(self._h)

    def getElapsedCpuTime(self):
        return lib().das_get_elapsed_cpu_time(self._h)

    # -- not on the hot path -----------------------------------------------------------------------
