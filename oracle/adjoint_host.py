"""ORACLE (test infrastructure only - never imported by the product path).

Host (OpenMP C++) assembly of the adjoint matrices dRdW^T and dRdWTPC for DASimpleFoam + Spalart-Allmaras at the sizes of the
parity legs (200 k cells): the CPU side of the psi comparison builds its OWN Jacobians - own connectivity from the reference's
stencil tables, own first-fit colouring, own residual evaluation (face-based scatter form of oracle/residual.py, dual numbers for
dRdW^T, the reference's one-sided differences for dRdWTPC) - instead of receiving the matrices of the GPU run (VERDICT round 4
item 3).  Source: oracle/csrc/oracle_adjoint_host.cpp (reference file:line there); pinned against oracle/residual.py and
oracle/jacobian.py on small meshes in tests/test_oracle_cpu.py.  PARITY UNPINNED like the rest of oracle/ (SURVEY.md 8c)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from dafoam_amd.meshgen import (BC_FIXED_VALUE, BC_INLET_OUTLET, BC_SYMMETRY, BC_ZERO_GRADIENT, NUT_CALCULATED, NUT_LOWRE_WALL,
                                NUT_SPALDING_WALL, NUT_SYMMETRY)

from .linear import available_cpus

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "oracle_adjoint_host.cpp")
_SO = os.path.join(_HERE, "_build", "liboracle_adjoint_host.so")
_lib = None
_dp, _ip, _lp, _up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_ubyte)


def build(force=False):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-o", _SO, _SRC])
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.oah_create.restype = vp
        L.oah_create.argtypes = [C.c_int] * 3 + [_ip, _ip] + [_dp] * 10 + [_ip] * 4 + [_dp] * 3 + [C.c_double, C.c_double, _ip, _ip, C.c_int]
        L.oah_free.argtypes = [vp]
        L.oah_n.restype = C.c_longlong
        L.oah_n.argtypes = [vp]
        L.oah_residual.argtypes = [vp, _dp, _dp, C.c_int, C.c_double]
        L.oah_jvp.argtypes = [vp, _dp, _dp, _dp, C.c_int, C.c_double]
        L.oah_setup.argtypes = [vp]
        L.oah_nnz.restype = C.c_longlong
        L.oah_nnz.argtypes = [vp, C.c_int]
        L.oah_get_pattern.argtypes = [vp, C.c_int, _lp, _ip]
        L.oah_get_colors.argtypes = [vp, _ip]
        L.oah_assemble.argtypes = [vp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _up]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class HostAdjoint:
    """The host Jacobian builder for one case (DASimpleFoam + SA; no MRF / SIMPLEC / T field / wall functions)."""

    def __init__(self, case, g, normalize=("URes", "pRes", "nuTildaRes", "phiRes"), use_constrain_hbya=True, threads=None):
        if case.solver_name != "DASimpleFoam" or getattr(case, "has_T", False) or getattr(case, "mrf", None) or getattr(case, "simple_consistent", False):
            raise ValueError("oracle.adjoint_host covers DASimpleFoam + SA without MRF / SIMPLEC / T field")
        from .residual import BCTable

        L = lib()
        bt = BCTable(case, g, ("U", "p", "nuTilda", "nut"))
        self.N, self.F, self.nIF = g.nC, g.nF, g.nIF
        self.threads = int(threads or available_cpus())
        keep = [_i(g.own), _i(g.nei), _d(g.Sf), _d(g.magSf), _d(g.w), _d(g.nonOrthDeltaCoeffs), _d(g.nonOrthCorr), _d(g.Cf), _d(g.C), _d(g.V),
                _d(case.y_wall), _d(g.bDeltaCoeffs), _i(bt.code["U"]), _i(bt.code["p"]), _i(bt.code["nuTilda"]), _i(bt.code["nut"]),
                _d(bt.val["U"]), _d(bt.val["p"]), _d(bt.val["nuTilda"])]
        norm5 = _i([int("URes" in normalize), int("pRes" in normalize), int("nuTildaRes" in normalize), int("phiRes" in normalize), int(bool(use_constrain_hbya))])
        codes8 = _i([BC_FIXED_VALUE, BC_ZERO_GRADIENT, BC_INLET_OUTLET, BC_SYMMETRY, NUT_CALCULATED, NUT_LOWRE_WALL, NUT_SYMMETRY, NUT_SPALDING_WALL])
        ptrs = [a.ctypes.data_as(_ip if a.dtype == np.int32 else _dp) for a in keep]
        self._h = L.oah_create(g.nC, g.nF, g.nIF, *ptrs, float(case.nu), float(case.relax["U"]), norm5.ctypes.data_as(_ip), codes8.ctypes.data_as(_ip), self.threads)
        if not self._h:
            raise ValueError("oracle.adjoint_host: wall-function patches are outside this port's scope")
        self.n = int(L.oah_n(self._h))
        self.nColors = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oah_free(self._h)
            self._h = None

    def residual(self, W, isPC=False, pc_blend=0.0):
        W = _d(W)
        R = np.empty(self.n)
        lib().oah_residual(self._h, W.ctypes.data_as(_dp), R.ctypes.data_as(_dp), int(isPC), float(pc_blend))
        return R

    def jvp(self, W, v, isPC=False, pc_blend=0.0):
        W, v = _d(W), _d(v)
        out = np.empty(self.n)
        lib().oah_jvp(self._h, W.ctypes.data_as(_dp), v.ctypes.data_as(_dp), out.ctypes.data_as(_dp), int(isPC), float(pc_blend))
        return out

    def setup(self):
        """Connectivity (full + PC) from the stencil tables and a first-fit colouring of the full pattern; returns the colour count."""
        nc = lib().oah_setup(self._h)
        if nc < 0:
            raise RuntimeError("oracle.adjoint_host: more than 2048 colours")
        self.nColors = int(nc)
        return self.nColors

    def pattern(self, isPC=False):
        """Transposed pattern (rows = states j, sorted residual rows i) as (rowptr int64, col int32)."""
        nnz = int(lib().oah_nnz(self._h, int(isPC)))
        rp, ci = np.empty(self.n + 1, np.int64), np.empty(nnz, np.int32)
        lib().oah_get_pattern(self._h, int(isPC), rp.ctypes.data_as(_lp), ci.ctypes.data_as(_ip))
        return rp, ci

    def colors(self):
        c = np.empty(self.n, np.int32)
        lib().oah_get_colors(self._h, c.ctypes.data_as(_ip))
        return c

    def assemble(self, W, scales, isPC=False, mode=None, delta=1e-6, lower_bound=1e-30, pc_blend=0.0):
        """dRdW^T (isPC False: dual numbers) or dRdWTPC (isPC True: one-sided differences, step delta * s_j, DAPartDeriv.C:350-473) as CSR
        (rowptr, col, val) with the entries the reference would insert (|v| > lower_bound or diagonal, DAPartDeriv.C:192-201)."""
        if self.nColors is None:
            self.setup()
        if mode is None:
            mode = "fd" if isPC else "dual"
        rp, ci = self.pattern(isPC)
        W, scales = _d(W), _d(scales)
        val = np.zeros(ci.size)
        keep = np.zeros(ci.size, np.uint8)
        lib().oah_assemble(self._h, W.ctypes.data_as(_dp), scales.ctypes.data_as(_dp), int(isPC), 0 if mode == "fd" else 1, float(delta), float(lower_bound),
                           float(pc_blend), val.ctypes.data_as(_dp), keep.ctypes.data_as(_up))
        if lower_bound < 1e-16:  # (oracle/jacobian.py: a lower bound below the rounding level keeps the whole pattern)
            return rp, ci, val
        k = keep.astype(bool)
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(rp))[k]
        rp2 = np.zeros(self.n + 1, np.int64)
        np.cumsum(np.bincount(rows, minlength=self.n), out=rp2[1:])
        return rp2, ci[k], val[k]
