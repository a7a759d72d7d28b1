"""ctypes binding of include/dafoam_amd.h (thin; no compute here).

The library is built in-tree by __graft_entry__.build() (hipcc --offload-arch=gfx950).  Import fails
loudly if the shared object is missing: there is no Python/CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .meshgen import FoamCase

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAFOAM_AMD_LIB") or os.path.join(_HERE, "lib", "libdafoam_amd.so")  # (override: tuning builds)

SOLVER_IDS = {"DASimpleFoam": 0, "DAScalarTransportFoam": 1, "DARhoSimpleFoam": 2, "DATurboFoam": 3}
PATCH_TYPES = {"patch": 0, "wall": 1, "symmetry": 2, "cyclic": 3}

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)


class das_case_t(C.Structure):
    _fields_ = [
        ("solver", C.c_int),
        ("n_points", C.c_int),
        ("n_faces", C.c_int),
        ("n_internal_faces", C.c_int),
        ("n_cells", C.c_int),
        ("n_patches", C.c_int),
        ("points", c_double_p),
        ("face_ptr", c_int_p),
        ("face_pts", c_int_p),
        ("owner", c_int_p),
        ("neighbour", c_int_p),
        ("patch_start", c_int_p),
        ("patch_size", c_int_p),
        ("patch_type", c_int_p),
        ("bc_U_code", c_int_p),
        ("bc_U_val", c_double_p),
        ("bc_p_code", c_int_p),
        ("bc_p_val", c_double_p),
        ("bc_nuTilda_code", c_int_p),
        ("bc_nuTilda_val", c_double_p),
        ("bc_nut_code", c_int_p),
        ("bc_T_code", c_int_p),
        ("bc_T_val", c_double_p),
        ("nu", C.c_double),
        ("relax_U", C.c_double),
        ("relax_nuTilda", C.c_double),
        ("relax_T", C.c_double),
        ("DT", C.c_double),
        ("deltaT", C.c_double),
        ("y_wall", c_double_p),
        ("phi_frozen", c_double_p),
        ("T_old", c_double_p),
        ("Cp", C.c_double),
        ("molWeight", C.c_double),
        ("mu", C.c_double),
        ("Pr", C.c_double),
        ("Prt", C.c_double),
        ("mrf_active", C.c_int),
        ("mrf_omega", C.c_double * 3),
        ("mrf_origin", C.c_double * 3),
        ("patch_mrf_rotating", c_int_p),
        ("transonic", C.c_int),
        ("transonic_pc_option", C.c_int),
        ("simple_has_T", C.c_int),
        ("patch_neighbour", c_int_p),
        ("patch_rotation", c_double_p),
        ("transport_sutherland", C.c_int),
        ("sutherland_As", C.c_double),
        ("sutherland_Ts", C.c_double),
        ("beta_fi_nuTilda", c_double_p),
    ]


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_int_p) if a is not None else None


class CaseStruct:
    """Keeps the numpy buffers alive next to the ctypes struct."""

    def __init__(self, case: FoamCase):
        m = case.mesh
        k = self.keep = {}
        k["points"] = np.ascontiguousarray(m.points, dtype=np.float64)
        k["face_ptr"] = np.ascontiguousarray(m.face_ptr, dtype=np.int32)
        k["face_pts"] = np.ascontiguousarray(m.face_pts, dtype=np.int32)
        k["owner"] = np.ascontiguousarray(m.owner, dtype=np.int32)
        k["neighbour"] = np.ascontiguousarray(m.neighbour, dtype=np.int32)
        P = len(m.patches)
        k["patch_start"] = np.array([p.start for p in m.patches], dtype=np.int32)
        k["patch_size"] = np.array([p.size for p in m.patches], dtype=np.int32)
        k["patch_type"] = np.array([PATCH_TYPES[p.type] for p in m.patches], dtype=np.int32)

        def table(field, vec=False):
            code = np.zeros(P, dtype=np.int32)
            val = np.zeros((P, 3) if vec else P, dtype=np.float64)
            present = False
            for i, p in enumerate(m.patches):
                ent = case.bcs.get(p.name, {}).get(field)
                if ent is None:
                    code[i] = 1  # zeroGradient
                    continue
                present = True
                code[i] = ent[0]
                val[i] = ent[1]
            return (code, np.ascontiguousarray(val)) if present else (None, None)

        k["bc_U_code"], k["bc_U_val"] = table("U", True)
        k["bc_p_code"], k["bc_p_val"] = table("p")
        k["bc_nuTilda_code"], k["bc_nuTilda_val"] = table("nuTilda")
        k["bc_nut_code"], _ = table("nut")
        k["bc_T_code"], k["bc_T_val"] = table("T")
        k["y_wall"] = None if case.y_wall is None else np.ascontiguousarray(case.y_wall, dtype=np.float64)
        k["phi_frozen"] = None if case.phi is None else np.ascontiguousarray(case.phi, dtype=np.float64)
        k["T_old"] = None if case.T_old is None else np.ascontiguousarray(case.T_old, dtype=np.float64)
        bfi = getattr(case, "beta_fi", None)
        k["beta_fi"] = None if bfi is None else np.ascontiguousarray(bfi, dtype=np.float64)
        s = self.struct = das_case_t()
        s.solver = SOLVER_IDS[case.solver_name]
        s.n_points, s.n_faces, s.n_internal_faces, s.n_cells, s.n_patches = (
            m.n_points,
            m.n_faces,
            m.n_internal_faces,
            m.n_cells,
            P,
        )
        for name in ("points", "bc_U_val", "bc_p_val", "bc_nuTilda_val", "bc_T_val", "y_wall", "phi_frozen", "T_old"):
            setattr(s, name, _dp(k[name]))
        for name in (
            "face_ptr", "face_pts", "owner", "neighbour", "patch_start", "patch_size", "patch_type",
            "bc_U_code", "bc_p_code", "bc_nuTilda_code", "bc_nut_code", "bc_T_code",
        ):
            setattr(s, name, _ip(k[name]))
        s.nu = case.nu
        s.relax_U = case.relax.get("U", 0.7)
        s.relax_nuTilda = case.relax.get("nuTilda", 0.7)
        s.relax_T = case.relax.get("T", 1.0)
        s.DT = case.DT
        s.deltaT = case.deltaT
        th = getattr(case, "thermo", None) or {}
        s.Cp, s.molWeight, s.mu, s.Pr, s.Prt = (th.get("Cp", 1005.0), th.get("molWeight", 28.96), th.get("mu", 1.8e-5), th.get("Pr", 0.7), th.get("Prt", 1.0))
        s.beta_fi_nuTilda = _dp(k["beta_fi"])
        s.transport_sutherland = 1 if th.get("transport", "const") == "sutherland" else 0
        s.sutherland_As, s.sutherland_Ts = th.get("As", 1.4792e-06), th.get("Ts", 116.0)
        mrf = getattr(case, "mrf", None)
        s.mrf_active = 1 if mrf else 0
        if mrf:
            for i in range(3):
                s.mrf_omega[i] = float(mrf["omega"][i])
                s.mrf_origin[i] = float(mrf.get("origin", (0.0, 0.0, 0.0))[i])
            k["patch_mrf_rotating"] = np.array([0 if p.name in mrf.get("nonRotatingPatches", ()) else 1 for p in m.patches], dtype=np.int32)
            s.patch_mrf_rotating = _ip(k["patch_mrf_rotating"])
        s.transonic = 1 if getattr(case, "transonic", False) else 0
        s.transonic_pc_option = int(getattr(case, "transonic_pc_option", 1))
        if any(p.type == "cyclic" for p in m.patches):
            pnames = [p.name for p in m.patches]
            k["patch_neighbour"] = np.array([pnames.index(p.neighbour) if p.type == "cyclic" else -1 for p in m.patches], dtype=np.int32)
            s.patch_neighbour = _ip(k["patch_neighbour"])
            if any(getattr(p, "rotation", None) is not None for p in m.patches):
                k["patch_rotation"] = np.ascontiguousarray(
                    [np.asarray(p.rotation, dtype=np.float64).reshape(9) if getattr(p, "rotation", None) is not None else np.eye(3).reshape(9)
                     for p in m.patches], dtype=np.float64)
                s.patch_rotation = _dp(k["patch_rotation"])
        s.simple_has_T = 1 if (case.solver_name == "DASimpleFoam" and getattr(case, "has_T", False)) else 0

    def byref(self):
        return C.byref(self.struct)


# every symbol include/dafoam_amd.h declares, with its signature
_VP = C.c_void_p
_SIGS = {
    "das_last_error": (C.c_char_p, []),
    "das_version": (C.c_int, []),
    "das_device_count": (C.c_int, []),
    "das_create": (_VP, [C.POINTER(das_case_t)]),
    "das_destroy": (None, [_VP]),
    "das_set_option_double": (C.c_int, [_VP, C.c_char_p, C.c_double]),
    "das_set_option_int": (C.c_int, [_VP, C.c_char_p, C.c_longlong]),
    "das_set_option_str": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "das_get_option_double": (C.c_int, [_VP, C.c_char_p, c_double_p]),
    "das_init_solver": (C.c_int, [_VP, C.c_int]),
    "das_get_n_local_adjoint_states": (C.c_longlong, [_VP]),
    "das_get_n_local_cells": (C.c_longlong, [_VP]),
    "das_get_n_global_cells": (C.c_longlong, [_VP]),
    "das_set_n_global_cells": (C.c_int, [_VP, C.c_longlong]),
    "das_get_n_local_points": (C.c_longlong, [_VP]),
    "das_get_n_local_faces": (C.c_longlong, [_VP]),
    "das_get_geometry": (C.c_int, [_VP] + [c_double_p] * 8),
    "das_update_of_fields": (C.c_int, [_VP, c_double_p]),
    "das_get_of_fields": (C.c_int, [_VP, c_double_p]),
    "das_get_residuals": (C.c_int, [_VP, c_double_p]),
    "das_calc_residuals": (C.c_int, [_VP, C.c_int, c_double_p]),
    "das_solve_primal": (C.c_int, [_VP, C.c_int, C.c_double, C.c_double, c_double_p, c_double_p, C.c_int]),
    "das_simple_iteration": (C.c_int, [_VP, C.c_int, C.c_double, C.c_double, C.c_int, c_double_p]),
    "das_run_coloring": (C.c_int, [_VP]),
    "das_update_of_mesh": (C.c_int, [_VP, c_double_p]),
    "das_get_of_mesh_points": (C.c_int, [_VP, c_double_p]),
    "das_set_coloring": (C.c_int, [_VP, c_int_p]),
    "das_debug_factor_block": (C.c_int, [C.c_int, c_ll_p, c_int_p, c_double_p, C.c_int, c_double_p, c_ll_p, c_int_p, c_int_p]),
    "das_get_n_colors": (C.c_int, [_VP, C.c_int]),
    "das_get_con_nnz": (C.c_longlong, [_VP, C.c_int]),
    "das_get_con": (C.c_int, [_VP, C.c_int, c_ll_p, c_int_p]),
    "das_get_colors": (C.c_int, [_VP, C.c_int, c_int_p]),
    "das_calc_drdwt": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "das_mat_rows": (C.c_longlong, [_VP]),
    "das_mat_nnz": (C.c_longlong, [_VP]),
    "das_mat_export": (C.c_int, [_VP, c_ll_p, c_int_p, c_double_p]),
    "das_mat_mult": (C.c_int, [_VP, c_double_p, c_double_p]),
    "das_mat_create_from_csr": (C.c_int, [C.c_longlong, c_ll_p, c_int_p, c_double_p, C.POINTER(_VP)]),
    "das_mat_destroy": (None, [_VP]),
    "das_initialize_drdwt_matrix_free": (C.c_int, [_VP]),
    "das_destroy_drdwt_matrix_free": (C.c_int, [_VP]),
    "das_op_nnz": (C.c_longlong, [_VP]),
    "das_op_format_bytes": (C.c_longlong, [_VP]),
    "das_op_export": (C.c_int, [_VP, c_ll_p, c_int_p, c_double_p]),
    "das_calc_drdwold_t_psi": (C.c_int, [_VP, C.c_int, c_double_p, c_double_p]),
    "das_set_old_time_fields": (C.c_int, [_VP, c_double_p, c_double_p]),
    "das_calc_jac_vec_product": (C.c_int, [_VP, c_double_p, c_double_p]),
    "das_set_patch_value": (C.c_int, [_VP, c_int_p, C.c_int, C.c_char_p, c_double_p]),
    "das_get_patch_value": (C.c_int, [_VP, C.c_int, C.c_char_p, c_double_p]),
    "das_set_field": (C.c_int, [_VP, C.c_char_p, c_double_p]),
    "das_get_field": (C.c_int, [_VP, C.c_char_p, c_double_p]),
    "das_calc_dfield_product": (C.c_int, [_VP, C.c_char_p, C.c_char_p, C.c_char_p, c_double_p, c_double_p]),
    "das_calc_dvolcoord_product": (C.c_int, [_VP, C.c_char_p, C.c_char_p, c_double_p, c_double_p, c_double_p]),
    "das_point_influence_build": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "das_point_influence_get": (C.c_int, [_VP, c_int_p, c_ll_p, c_int_p, c_double_p]),
    "das_debug_device_geometry": (C.c_int, [_VP, c_double_p, c_double_p, c_double_p]),
    "das_debug_strength_aggregates": (C.c_int, [_VP, C.c_int, c_int_p, C.POINTER(C.c_int)]),
    "das_calc_dbc_product": (C.c_int, [_VP, c_int_p, C.c_int, C.c_char_p, c_double_p, C.c_char_p, C.c_char_p, c_double_p, c_double_p]),
    "das_define_force_function": (C.c_int, [_VP, C.c_char_p, c_int_p, C.c_int, c_double_p, C.c_double]),
    "das_define_face_function": (C.c_int, [_VP, C.c_char_p, C.c_char_p, c_int_p, c_int_p, C.c_int, c_double_p, c_double_p, C.c_double, C.c_double]),
    "das_calc_function": (C.c_int, [_VP, C.c_char_p, c_double_p]),
    "das_get_input_size": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "das_get_output_size": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "das_calc_jac_t_vec_product": (
        C.c_int,
        [_VP, C.c_char_p, C.c_char_p, c_double_p, C.c_char_p, C.c_char_p, c_double_p, c_double_p],
    ),
    "das_drdwt_mult_device": (C.c_int, [_VP, _VP, _VP]),
    "das_create_ml_rksp_matrix_free": (C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    "das_solve_linear_eqn": (C.c_int, [_VP, _VP, c_double_p, c_double_p]),
    "das_solve_linear_eqn_block": (C.c_int, [_VP, _VP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "das_ksp_apply_pc": (C.c_int, [_VP, _VP, c_double_p, c_double_p]),
    "das_ksp_get_n_blocks": (C.c_int, [_VP]),
    "das_ksp_get_factor_nnz": (C.c_longlong, [_VP]),
    "das_ksp_get_n_ext": (C.c_longlong, [_VP]),
    "das_ksp_get_blocks": (C.c_int, [_VP, c_int_p, c_ll_p]),
    "das_pc_structure_build": (C.c_int, [_VP, c_int_p, c_ll_p, c_int_p, c_int_p]),
    "das_pc_structure_get": (C.c_int, [_VP, c_int_p, c_ll_p, c_int_p, c_int_p, c_int_p]),
    "das_ksp_get_pc_structure_sizes": (C.c_int, [_VP, c_int_p, c_ll_p, c_int_p]),
    "das_ksp_get_pc_structure": (C.c_int, [_VP, c_int_p, c_ll_p, c_int_p, c_int_p, c_int_p]),
    "das_ksp_get_info": (C.c_int, [_VP, c_int_p, c_double_p, c_double_p, c_double_p]),
    "das_ksp_get_history": (C.c_int, [_VP, c_double_p, C.c_int]),
    "das_ksp_get_cycle_lengths": (C.c_int, [_VP, c_int_p, C.c_int]),
    "das_ksp_get_basis_info": (C.c_int, [_VP, c_int_p, c_double_p, c_double_p]),
    "das_ksp_get_n_refine": (C.c_int, [_VP]),
    "das_set_dense_eig_callback": (C.c_int, [_VP]),
    "das_debug_gmres_dr_host": (C.c_int, [C.c_longlong, _VP, _VP, _VP, c_double_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_longlong, c_double_p, C.c_int,
                                          c_double_p, c_double_p]),
    "das_debug_gmres_dr_restart": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "das_ksp_get_status": (C.c_int, [_VP, c_int_p, c_int_p, c_int_p, c_int_p]),
    "das_ksp_get_pc_stability": (C.c_int, [_VP, C.POINTER(C.c_double), c_int_p]),
    "das_ksp_get_pc_subdomains": (C.c_int, [_VP, c_int_p, c_double_p]),
    "das_ksp_get_pc_node_out": (C.c_int, [_VP, c_int_p]),
    "das_mesh_metrics": (C.c_int, [C.c_int, c_double_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "das_ksp_get_coarse": (C.c_int, [_VP, c_int_p]),
    "das_ksp_coarse_sparse_az_active": (C.c_int, [_VP]),
    "das_ksp_set_global_coarse": (C.c_int, [_VP, _VP, C.c_int, C.c_int, c_int_p]),
    "das_ksp_run_fixed_device": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int]),
    "das_ksp_begin_device": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int]),
    "das_ksp_advance": (C.c_int, [_VP, _VP, C.c_int]),
    "das_ksp_end": (C.c_int, [_VP, _VP]),
    "das_ksp_destroy": (None, [_VP]),
    "das_set_owned_mask": (C.c_int, [_VP, C.POINTER(C.c_ubyte)]),
    "das_comm_load_rccl": (C.c_int, []),
    "das_comm_unique_id": (C.c_int, [C.c_char_p]),
    "das_comm_init_rccl": (C.c_int, [_VP, C.c_int, C.c_int, C.c_char_p]),
    "das_comm_set_halo": (C.c_int, [_VP, C.c_int, c_int_p, c_ll_p, c_int_p, c_ll_p, c_int_p, C.c_longlong, c_int_p]),
    "das_set_exchange_cb": (C.c_int, [_VP, _VP, _VP]),
    "das_set_pc_overlap": (C.c_int, [_VP, C.POINTER(C.c_ubyte), C.c_int, c_ll_p, c_int_p, c_ll_p, c_int_p]),
    "das_set_gather_cb": (C.c_int, [_VP, _VP, _VP]),
    "das_comm_is_native": (C.c_int, [_VP]),
    "das_comm_reset": (C.c_int, [_VP]),
    "das_set_comm": (C.c_int, [_VP, _VP, _VP, _VP]),
    "das_set_stream": (C.c_int, [_VP, _VP]),
    "das_get_elapsed_clock_time": (C.c_double, [_VP]),
    "das_get_elapsed_cpu_time": (C.c_double, [_VP]),
    "das_timer_avg_ms": (C.c_double, [_VP, C.c_char_p]),
    "das_debug_orth_bench": (C.c_int, [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "das_timer_count": (C.c_longlong, [_VP, C.c_char_p]),
    "das_timer_reset": (None, [_VP]),
    "das_timer_enable": (None, [_VP, C.c_int]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load libdafoam_amd.so (built in-tree); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "dafoam_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        _install_dense_eig(L)
    return _lib


_DENSE_EIG_FN = C.CFUNCTYPE(C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p)
_dense_eig_keepalive = []


def _install_dense_eig(L):
    """The dense nonsymmetric eigen-solver of the deflated restart (amd.gmresDeflation): numpy.linalg.eig behind the callback the
    C-ABI asks for (the library itself carries no LAPACK; a C host would pass LAPACK's dgeev the same way)."""

    def eig(m, A, wr, wi, vr, vi):
        try:
            G = np.ctypeslib.as_array(A, shape=(m * m,)).reshape(m, m)
            w, v = np.linalg.eig(G)
            np.ctypeslib.as_array(wr, shape=(m,))[:] = w.real
            np.ctypeslib.as_array(wi, shape=(m,))[:] = w.imag
            np.ctypeslib.as_array(vr, shape=(m * m,))[:] = np.ascontiguousarray(v.T.real).ravel()
            np.ctypeslib.as_array(vi, shape=(m * m,))[:] = np.ascontiguousarray(v.T.imag).ravel()
            return 0
        except Exception:  # noqa: BLE001 - never raise through the C boundary
            return 1

    cb = _DENSE_EIG_FN(eig)
    _dense_eig_keepalive.append(cb)
    L.das_set_dense_eig_callback(C.cast(cb, C.c_void_p))


class DASError(RuntimeError):
    pass


def check(rc):
    if rc < 0:
        raise DASError(f"dafoam_amd error {rc}: {lib().das_last_error().decode()}")
    return rc


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)
