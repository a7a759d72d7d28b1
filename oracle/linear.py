"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's Krylov solve (reference src/adjoint/DALinearEqn/DALinearEqn.C:28-339
createMLRKSP: restarted GMRES, right preconditioning, classical Gram-Schmidt with refinement, ILU(k)
sub-solves, unpreconditioned residual norm; :341-437 solveLinearEqn with the gmresTolDiff failure rule
:422-434; defaults reference dafoam/pyDAFoam.py:526-548).  PETSc itself is un-vendored: PARITY UNPINNED;
the solver is pinned by scipy's sparse direct solve in tests/ (psi is preconditioner independent).

The C kernels live in oracle/csrc/oracle_linalg.c (built on demand with gcc into oracle/_build/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "oracle_linalg.c")
_SO = os.path.join(_HERE, "_build", "liboracle_linalg.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_longlong)


def build(force=False):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-shared", "-o", _SO, _SRC, "-lm"])
    build_omp(force)
    return _SO


_SRC_OMP = os.path.join(_HERE, "csrc", "oracle_krylov_omp.c")
_SO_OMP = os.path.join(_HERE, "_build", "liboracle_krylov_omp.so")
_lib_omp = None


def build_omp(force=False):
    os.makedirs(os.path.dirname(_SO_OMP), exist_ok=True)
    if force or not os.path.exists(_SO_OMP) or os.path.getmtime(_SO_OMP) < os.path.getmtime(_SRC_OMP):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", _SO_OMP, _SRC_OMP, "-lm"])
    return _SO_OMP


def lib_omp():
    global _lib_omp
    if _lib_omp is None:
        L = C.CDLL(build_omp())
        vp = C.c_void_p
        L.okry_create.restype = vp
        L.okry_create.argtypes = [C.c_int]
        L.okry_threads.argtypes = [vp]
        L.okry_free.argtypes = [vp]
        L.okry_stream_GBps.restype = C.c_double
        L.okry_stream_GBps.argtypes = [vp, C.c_longlong, C.c_int]
        L.okry_set_operator.argtypes = [vp, C.c_longlong, _lp, _ip, _dp]
        L.okry_spmv.argtypes = [vp, _dp, _dp]
        L.okry_set_pc_ilu0.argtypes = [vp, C.c_longlong, _lp, _ip, _dp, _ip, C.c_double]
        L.okry_set_pc_bilu.argtypes = [vp, C.c_longlong, _lp, _ip, _dp, C.c_int, _ip, _lp, _ip, C.c_int, _ip, C.c_double]
        L.okry_set_pc_bilu2.argtypes = [vp, C.c_longlong, _lp, _ip, _dp, C.c_int, _ip, _ip, _lp, _ip, C.c_int, _ip, C.c_double]
        L.okry_bilu_blocks.restype = C.c_longlong
        L.okry_bilu_blocks.argtypes = [vp]
        L.okry_pc_levels.argtypes = [vp, _ip, _ip]
        L.okry_pc_nnz.restype = C.c_longlong
        L.okry_pc_nnz.argtypes = [vp]
        L.okry_set_coarse.argtypes = [vp, C.c_longlong, C.c_longlong, _ip, C.c_int, _dp]
        L.okry_coarse_operator.argtypes = [C.c_longlong, C.c_longlong, _ip, C.c_int, _lp, _ip, _dp, _dp]
        L.okry_pc.argtypes = [vp, _dp, _dp]
        L.okry_set_max_seconds.argtypes = [C.c_double]
        L.okry_gmres.argtypes = [vp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _dp, C.c_int, _dp]
        _lib_omp = L
    return _lib_omp


def available_cpus():
    """CPUs this process may really use: the smaller of the affinity mask and the container's CFS quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us`).  Round 4: the bench host shows 256 CPUs to a container whose quota is 16 - 128 OpenMP threads then run
    throttled (host STREAM 68 GB/s instead of 490, every barrier a scheduler time slice); the CPU legs use this count."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, p = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


class OmpKrylov:
    """All-core CPU solver of the adjoint system (oracle/csrc/oracle_krylov_omp.c): CSR operator with first-touch placement,
    ONE global level-scheduled scalar ILU(0) of the PC matrix in a caller-given unknown order, optional additive coarse
    correction on one scalar cell field, GMRES(CGS2).  Matrices are passed as raw CSR arrays (rowptr int64, col int32, values
    fp64) so that the bench-size systems (10^9 entries) never go through scipy."""

    def __init__(self, threads=0):
        self.L = lib_omp()
        self.h = C.c_void_p(self.L.okry_create(int(threads) if int(threads) > 0 else available_cpus()))
        self.threads = int(self.L.okry_threads(self.h))
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.okry_free(self.h)
            self.h = None

    @staticmethod
    def _csr(A):
        if isinstance(A, tuple):
            rp, ci, v = A
        else:
            A = sp.csr_matrix(A)
            rp, ci, v = A.indptr, A.indices, A.data
        return (np.ascontiguousarray(rp, dtype=np.int64), np.ascontiguousarray(ci, dtype=np.int32), np.ascontiguousarray(v, dtype=np.float64))

    def stream_GBps(self, n_doubles=1 << 27, reps=5):
        return float(self.L.okry_stream_GBps(self.h, int(n_doubles), int(reps)))

    def set_operator(self, A):
        rp, ci, v = self._csr(A)
        self.n = rp.size - 1
        assert self.L.okry_set_operator(self.h, self.n, _p(rp, _lp), _p(ci, _ip), _p(v, _dp)) == 0

    def set_pc(self, P, perm=None, shift=1e-12):
        rp, ci, v = self._csr(P)
        self.n = rp.size - 1
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        rc = self.L.okry_set_pc_ilu0(self.h, self.n, _p(rp, _lp), _p(ci, _ip), _p(v, _dp), _p(pm, _ip) if pm is not None else None, float(shift))
        assert rc >= 0, rc
        a, b = C.c_int(0), C.c_int(0)
        self.L.okry_pc_levels(self.h, C.byref(a), C.byref(b))
        self.levels = (a.value, b.value)
        self.nshift = rc
        return rc

    def set_pc_bilu(self, P, structure, shift=1e-12):
        """The product's default preconditioner restated for the host: node-block ILU(0) with dense 8 x 8 blocks on the node
        pattern of `structure` (the dict of KSP.pcStructure(): nodeUnk, bptr, bcol, lvlPtr in processing order), factorised and
        applied level by level with the nodes of a level in parallel (csrc/das_bilu.hpp; reference role: ILU of the one
        sub-domain of a rank, DALinearEqn.C:199-299).  ~600-1300 levels instead of the ~10^4 of the scalar ILU(0): this is what
        keeps a level-scheduled host solve off the barrier floor on a 256-thread machine."""
        rp, ci, v = self._csr(P)
        self.n = rp.size - 1
        nu = np.ascontiguousarray(structure["nodeUnk"], dtype=np.int32).reshape(-1, 8)
        bptr = np.ascontiguousarray(structure["bptr"], dtype=np.int64)
        bcol = np.ascontiguousarray(structure["bcol"], dtype=np.int32)
        lvl = np.ascontiguousarray(structure["lvlPtr"], dtype=np.int32)
        nout = structure.get("nodeOut") if hasattr(structure, "get") else None
        if nout is not None and not np.array_equal(np.asarray(nout).reshape(-1, 8), nu):
            # multi-block structure (amd.pcSubdomains of the product): overlap unknowns sit in one node per block, only the owner's copy writes
            nout = np.ascontiguousarray(nout, dtype=np.int32).reshape(-1, 8)
            rc = self.L.okry_set_pc_bilu2(self.h, self.n, _p(rp, _lp), _p(ci, _ip), _p(v, _dp), nu.shape[0], _p(nu, _ip), _p(nout, _ip), _p(bptr, _lp), _p(bcol, _ip),
                                          lvl.size - 1, _p(lvl, _ip), float(shift))
        else:
            rc = self.L.okry_set_pc_bilu(self.h, self.n, _p(rp, _lp), _p(ci, _ip), _p(v, _dp), nu.shape[0], _p(nu, _ip), _p(bptr, _lp), _p(bcol, _ip),
                                         lvl.size - 1, _p(lvl, _ip), float(shift))
        assert rc >= 0, rc
        self.levels = (lvl.size - 1, lvl.size - 1)
        self.nshift = rc
        return rc

    def set_coarse(self, P, offset, ncells, agg):
        """Additive correction Z (Z^T P_ff Z)^-1 Z^T on the scalar cell field at `offset` (aggregate id per cell)."""
        rp, ci, v = self._csr(P)
        agg = np.ascontiguousarray(agg, dtype=np.int32)
        nagg = int(agg.max()) + 1
        E = np.zeros((nagg, nagg))
        self.L.okry_coarse_operator(int(offset), int(ncells), _p(agg, _ip), nagg, _p(rp, _lp), _p(ci, _ip), _p(v, _dp), _p(E, _dp))
        Einv = np.ascontiguousarray(np.linalg.inv(E))
        self.L.okry_set_coarse(self.h, int(offset), int(ncells), _p(agg, _ip), nagg, _p(Einv, _dp))
        return nagg

    def matvec(self, x):
        y = np.empty(self.n)
        self.L.okry_spmv(self.h, _p(np.ascontiguousarray(x, dtype=np.float64), _dp), _p(y, _dp))
        return y

    def pc_solve(self, b):
        x = np.empty(self.n)
        self.L.okry_pc(self.h, _p(np.ascontiguousarray(b, dtype=np.float64), _dp), _p(x, _dp))
        return x

    def gmres(self, rhs, restart=1000, max_iters=1000, rel_tol=1e-6, abs_tol=1e-14, tol_diff=1e2, fixed_iters=0, max_seconds=0.0):
        """max_seconds > 0: a wall-clock bound (the solve stops at the iterate it has reached; info["fail"] then reports the state)."""
        self.L.okry_set_max_seconds(float(max_seconds))
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.empty(self.n)
        cap = int(max(max_iters, fixed_iters)) + 8
        hist = np.zeros(cap)
        info = np.zeros(8)
        fail = self.L.okry_gmres(self.h, _p(rhs, _dp), _p(x, _dp), int(restart), int(max_iters), float(rel_tol), float(abs_tol), float(tol_diff), int(fixed_iters),
                                 _p(hist, _dp), cap, _p(info, _dp))
        assert fail >= 0, "allocation failure in okry_gmres"
        return x, dict(iters=int(info[0]), res0=float(info[1]), res=float(info[2]), seconds=float(info[3]), seconds_spmv=float(info[4]), seconds_pc=float(info[5]),
                       seconds_orth=float(info[6]), hist=hist[: int(info[7])].copy(), fail=int(fail))


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.csr_spmv.argtypes = [C.c_longlong, _lp, _ip, _dp, _dp, _dp]
        L.ilu_symbolic.restype = C.c_void_p
        L.ilu_symbolic.argtypes = [C.c_longlong, _lp, _ip, C.c_int]
        L.ilu_nnz.restype = C.c_longlong
        L.ilu_nnz.argtypes = [C.c_void_p]
        L.ilu_get.argtypes = [C.c_void_p, _lp, _ip, _lp]
        L.ilu_free.argtypes = [C.c_void_p]
        L.ilu_numeric.restype = C.c_int
        L.ilu_numeric.argtypes = [C.c_longlong, _lp, _ip, _dp, _lp, _ip, _lp, _dp, C.c_double]
        L.ilu_solve.argtypes = [C.c_longlong, _lp, _ip, _lp, _dp, _dp, _dp]
        L.multi_dot.argtypes = [C.c_longlong, C.c_int, _dp, _dp, _dp]
        L.multi_axpy.argtypes = [C.c_longlong, C.c_int, _dp, _dp, _dp]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


class CSR:
    def __init__(self, A):
        A = sp.csr_matrix(A)
        A.sort_indices()
        self.n = A.shape[0]
        self.rp = A.indptr.astype(np.int64)
        self.ci = A.indices.astype(np.int32)
        self.v = np.ascontiguousarray(A.data, dtype=np.float64)

    def matvec(self, x):
        y = np.empty(self.n)
        lib().csr_spmv(self.n, _p(self.rp, _lp), _p(self.ci, _ip), _p(self.v, _dp), _p(np.ascontiguousarray(x), _dp), _p(y, _dp))
        return y


class ILU:
    """ILU(k) of a CSR matrix (natural ordering), optional block-Jacobi restriction:
    blocks = array of block id per row -> entries coupling different blocks are dropped before
    factorisation (additive Schwarz with zero overlap, one block per 'subdomain')."""

    def __init__(self, A, fill=0, blocks=None, shift=1e-12):
        A = sp.csr_matrix(A)
        if blocks is not None:
            A = A.tocoo()
            keep = blocks[A.row] == blocks[A.col]
            A = sp.csr_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape)
        A.sort_indices()
        a = CSR(A)
        L = lib()
        h = L.ilu_symbolic(a.n, _p(a.rp, _lp), _p(a.ci, _ip), fill)
        nnz = L.ilu_nnz(h)
        self.n = a.n
        self.rp = np.empty(a.n + 1, np.int64)
        self.ci = np.empty(nnz, np.int32)
        self.diag = np.empty(a.n, np.int64)
        L.ilu_get(h, _p(self.rp, _lp), _p(self.ci, _ip), _p(self.diag, _lp))
        L.ilu_free(h)
        self.v = np.empty(nnz)
        self.nshift = L.ilu_numeric(
            a.n, _p(a.rp, _lp), _p(a.ci, _ip), _p(a.v, _dp), _p(self.rp, _lp), _p(self.ci, _ip), _p(self.diag, _lp), _p(self.v, _dp), shift
        )

    def solve(self, b):
        x = np.empty(self.n)
        lib().ilu_solve(self.n, _p(self.rp, _lp), _p(self.ci, _ip), _p(self.diag, _lp), _p(self.v, _dp), _p(np.ascontiguousarray(b), _dp), _p(x, _dp))
        return x


def gmres(matvec, rhs, pc_solve=None, x0=None, restart=1000, max_iters=1000, rel_tol=1e-6, abs_tol=1e-14, tol_diff=1e2,
          fixed_iters=None):
    """Right-preconditioned restarted GMRES, CGS with one refinement pass (CGS2), Givens rotations,
    unpreconditioned residual norm.  Returns (x, info) with info = dict(iters, res0, res, hist, fail)
    where fail follows the reference rule DALinearEqn.C:422-434."""
    n = rhs.size
    x = np.zeros(n) if x0 is None else x0.copy()
    M = pc_solve if pc_solve is not None else (lambda v: v)
    L = lib()
    r = rhs - matvec(x) if x0 is not None else rhs.copy()
    beta = np.linalg.norm(r)
    res0 = beta
    hist = [beta]
    its = 0
    target = max(rel_tol * res0, abs_tol)
    done = beta <= target and fixed_iters is None
    while not done:
        m = min(restart, (fixed_iters if fixed_iters is not None else max_iters) - its)
        if m <= 0:
            break
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 1, m))
        cs = np.zeros(m)
        sn = np.zeros(m)
        gvec = np.zeros(m + 1)
        V[0] = r / beta
        gvec[0] = beta
        j = 0
        while j < m:
            w = matvec(M(V[j]))
            h = np.empty(j + 1)
            L.multi_dot(n, j + 1, _p(V, _dp), _p(w, _dp), _p(h, _dp))
            L.multi_axpy(n, j + 1, _p(V, _dp), _p(h, _dp), _p(w, _dp))
            h2 = np.empty(j + 1)
            L.multi_dot(n, j + 1, _p(V, _dp), _p(w, _dp), _p(h2, _dp))
            L.multi_axpy(n, j + 1, _p(V, _dp), _p(h2, _dp), _p(w, _dp))
            h += h2
            hn = np.linalg.norm(w)
            H[: j + 1, j] = h
            H[j + 1, j] = hn
            if hn > 0:
                V[j + 1] = w / hn
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            d = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j] = d
            H[j + 1, j] = 0.0
            gvec[j + 1] = -sn[j] * gvec[j]
            gvec[j] = cs[j] * gvec[j]
            its += 1
            j += 1
            res = abs(gvec[j])
            hist.append(res)
            if fixed_iters is None and (res <= target or its >= max_iters):
                break
        y = np.linalg.solve(np.triu(H[:j, :j]), gvec[:j])
        x = x + M(V[:j].T @ y)
        r = rhs - matvec(x)
        beta = np.linalg.norm(r)
        hist[-1] = beta
        if fixed_iters is not None:
            done = its >= fixed_iters
        else:
            done = beta <= target or its >= max_iters
    res = hist[-1]
    fail = int((res / res0 / rel_tol > tol_diff) and (res / abs_tol > tol_diff)) if res0 > 0 else 0
    return x, dict(iters=its, res0=res0, res=res, hist=np.array(hist), fail=fail)


BREAKDOWN_TOL = 1e-13  # csrc/das_device.hip GMRES_BREAKDOWN_TOL


def gmres_dcgs2(matvec, rhs, pc_solve=None, restart=1000, max_iters=1000, rel_tol=1e-6, abs_tol=1e-14, tol_diff=1e2, noise=0.0, rng=None):
    """Restatement of the GPU engine's default orthogonalisation (csrc/das_device.hip gmres_iter_dcgs2 + gmres_advance):
    right-preconditioned GMRES with classical Gram-Schmidt and DELAYED re-orthogonalisation (Bielich et al. 2022) - per step
    ONE product [Q u]^T [u v] and ONE update that finishes q_j and produces the once-projected next vector.  Same interface
    and the same iterates as `gmres` (CGS2); the Hessenberg column of a step is known one step later.

    Breakdown rule (as in the engine): what is left of B q_{j-1} after the projections is compared with |B q_{j-1}|
    (= sqrt(|h1|^2 + u.u)); below BREAKDOWN_TOL it is rounding noise: the column is closed with a zero sub-diagonal (happy
    breakdown) and the cycle ends.  A cycle that ended early (breakdown, lost orthogonality) or on a recurrence residual
    below the target must leave a true residual below the target; two such cycles in a row that do not halve it end the
    solve (info["reason"] = 2).  `noise` > 0 perturbs every inner product and operator image relatively (test hook: the
    run-to-run rounding differences of a parallel machine)."""
    n = rhs.size
    M = pc_solve if pc_solve is not None else (lambda v: v)
    rng = rng if rng is not None else np.random.default_rng(0)
    jit = (lambda a: a * (1.0 + noise * rng.standard_normal(np.shape(a)))) if noise > 0 else (lambda a: a)
    x = np.zeros(n)
    r = rhs.copy()
    beta = np.linalg.norm(r)
    res0 = beta
    hist = [beta]
    its = 0
    target = max(rel_tol * res0, abs_tol)
    done = beta <= target
    non_improving, n_breakdown, stalled, safe = 0, 0, False, False
    while not done:
        m = min(restart, max_iters - its)
        if m <= 0:
            break
        beta_start = beta
        early = False
        if safe:  # after lost orthogonality the engine runs the two-pass scheme for the following cycles
            x1, i1 = gmres(lambda v: jit(matvec(v)), r, pc_solve, restart=m, max_iters=m, rel_tol=target / beta, abs_tol=0.0)
            x = x + x1
            its += i1["iters"]
            hist.extend(list(i1["hist"][1:]))
            res = hist[-1]
        else:
            Q = np.zeros((m + 2, n))
            H = np.zeros((m + 1, m))
            Q[0] = r / beta          # pending vector u, not yet normalised "finally"
            h1 = np.zeros(0)
            ncol = 0
            y = np.zeros(0)
            res = beta
            for j in range(m + 1):   # step j makes q_j final and completes Hessenberg column j - 1
                u = Q[j].copy()
                v = jit(matvec(M(u)))
                QU = Q[: j + 1]
                su, tv = jit(QU @ u), jit(QU @ v)  # the one fused pass: [Q u]^T u, [Q u]^T v
                s, uu, t, uv = su[:j], su[j], tv[:j], tv[j]
                al2, num = uu - s @ s, uv - s @ t
                norm_bq2 = uu + h1 @ h1
                explicit = j > 0 and not (s @ s <= 1e-2 * uu)
                if explicit:  # u is (nearly) noise in span(Q): Pythagoras cancels - project explicitly (same rule as the GPU engine)
                    u = u - s @ Q[:j]
                    al2, num = u @ u, u @ v
                breakdown = not (al2 > BREAKDOWN_TOL ** 2 * norm_bq2)
                al = 0.0 if breakdown else np.sqrt(al2)
                if j == 0:
                    g0 = beta * al
                else:
                    H[:j, j - 1] = h1 + (0.0 if breakdown else s)
                    H[j, j - 1] = al
                    ncol = j
                    its += 1
                    e = np.zeros(j + 1)
                    e[0] = g0
                    y, *_ = np.linalg.lstsq(H[: j + 1, :j], e, rcond=None)
                    res = np.linalg.norm(H[: j + 1, :j] @ y - e)
                    hist.append(res)
                    if breakdown or explicit:
                        early = True
                        n_breakdown += int(breakdown)
                        safe = safe or not breakdown
                        break
                    if res <= target or its >= max_iters or j == m:
                        break
                gam = num / al2
                Q[j] = (u - s @ Q[:j]) / al        # the one fused update
                Q[j + 1] = (v - gam * u - (t - gam * s) @ Q[:j]) / al
                h1 = np.concatenate([(t - H[:j, :j] @ s) / al, [gam - (s[j - 1] if j > 0 else 0.0)]])
            x = x + M(Q[:ncol].T @ y)
        r = rhs - matvec(x)
        beta = np.linalg.norm(r)
        hist[-1] = beta
        judged = early or (res <= target and its < max_iters)
        if judged and beta > target:
            if beta < 0.5 * beta_start:
                non_improving = 0
            else:
                non_improving += 1
                stalled = non_improving >= 2
        done = beta <= target or its >= max_iters or stalled
    res = hist[-1]
    fail = int((res / res0 / rel_tol > tol_diff) and (res / abs_tol > tol_diff)) if res0 > 0 else 0
    return x, dict(iters=its, res0=res0, res=res, hist=np.array(hist), fail=fail, reason=0 if res <= target else (2 if stalled else 1),
                   n_breakdown=n_breakdown)


class ThreadedOperators:
    """Multi-core variant of the CPU baseline (bench.py `cpu_baseline_mt`): the rows are cut into `threads` contiguous
    chunks of equal nnz; the mat-vec runs one chunk per thread (the C kernels release the GIL), the preconditioner is
    block-Jacobi ILU(fill) with one block per chunk - the reference's layout of one ASM sub-domain per MPI rank
    (DALinearEqn.C:199-299) without the overlap."""

    def __init__(self, A, P, threads, fill=0):
        from concurrent.futures import ThreadPoolExecutor

        A = sp.csr_matrix(A)
        P = sp.csr_matrix(P)
        self.n = A.shape[0]
        threads = max(1, min(int(threads), self.n))
        cuts = np.searchsorted(A.indptr, np.linspace(0, A.nnz, threads + 1))
        cuts[0], cuts[-1] = 0, self.n
        self.bounds = [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        self.pool = ThreadPoolExecutor(max_workers=len(self.bounds))
        self.rows = [CSR(A[a:b]) for a, b in self.bounds]
        self.ilus = list(self.pool.map(lambda ab: ILU(P[ab[0]:ab[1]][:, ab[0]:ab[1]], fill=fill), self.bounds))
        self.threads = len(self.bounds)

    def matvec(self, x):
        x = np.ascontiguousarray(x)
        return np.concatenate(list(self.pool.map(lambda c: c.matvec(x), self.rows)))

    def pc_solve(self, b):
        b = np.ascontiguousarray(b)
        return np.concatenate(list(self.pool.map(lambda t: t[0].solve(b[t[1][0]:t[1][1]]), zip(self.ilus, self.bounds))))


class NodeBlockILU:
    """Oracle of the product's default preconditioner (amd.pcType "bilu"): block ILU(0) of P on a given grouping of the
    unknowns into nodes of <= 8 slots and a given node pattern - the reference's PC stack ASM + ILU
    (DALinearEqn.C:199-299) with ONE sub-domain, restated as a dense-block incomplete factorisation.

    node_unk[nNodes, 8] = unknown index or -1 (empty slot = identity row), bptr/bcol = block CSR over node positions.
    Two independent formulations are provided and compared in tests/:
      * `solve`        - dense 8x8 block IKJ factorisation in the GIVEN node order (numpy);
      * `scalar_twin`  - the scalar ILU(0) (oracle C kernel) of P with the node pattern filled in explicitly, in any
                         node order that keeps the relative order of coupled nodes (e.g. the natural one): the two are
                         the same incomplete factorisation, because both drop exactly the updates outside the pattern.
    """

    def __init__(self, P, node_unk, bptr, bcol, shift=1e-12):
        P = sp.csr_matrix(P)
        self.n = P.shape[0]
        nu = np.asarray(node_unk).reshape(-1, 8)
        self.nu = nu
        nN = nu.shape[0]
        self.bptr = np.asarray(bptr, dtype=np.int64)
        self.bcol = np.asarray(bcol, dtype=np.int64)
        pos = np.full(self.n, -1, dtype=np.int64)
        slot = np.zeros(self.n, dtype=np.int64)
        for k in range(8):
            m = nu[:, k] >= 0
            pos[nu[m, k]] = np.nonzero(m)[0]
            slot[nu[m, k]] = k
        self.pos, self.slot = pos, slot
        nB = self.bcol.size
        val = np.zeros((nB, 8, 8))
        # scatter the scalar entries (rows/cols outside the grouping are dropped: block-Jacobi across ranks)
        C = P.tocoo()
        keep = (pos[C.row] >= 0) & (pos[C.col] >= 0)
        I, J = pos[C.row[keep]], pos[C.col[keep]]
        key = I * nN + J
        bkey = np.repeat(np.arange(nN), np.diff(self.bptr)) * nN + self.bcol
        order = np.argsort(bkey)
        e = order[np.minimum(np.searchsorted(bkey[order], key), nB - 1)]
        inpat = bkey[e] == key
        # entries outside the node pattern are dropped (the product drops the couplings between two "late" nodes); the
        # caller checks `dropped_pairs` against what it expects
        self.dropped_pairs = np.unique(np.stack([I[~inpat], J[~inpat]], axis=1), axis=0) if not np.all(inpat) else np.zeros((0, 2), np.int64)
        rr, cc, dd = C.row[keep][inpat], C.col[keep][inpat], C.data[keep][inpat]
        np.add.at(val, (e[inpat], slot[rr], slot[cc]), dd)
        diag = np.empty(nN, dtype=np.int64)
        for p in range(nN):
            b0, b1 = self.bptr[p], self.bptr[p + 1]
            diag[p] = b0 + np.searchsorted(self.bcol[b0:b1], p)
            for k in range(8):
                if nu[p, k] < 0:
                    val[diag[p], k, k] = 1.0
        invD = np.zeros((nN, 8, 8))
        self.nshift = 0
        for p in range(nN):
            b0, b1 = self.bptr[p], self.bptr[p + 1]
            cols = self.bcol[b0:b1]
            for e in range(b0, diag[p]):
                J = self.bcol[e]
                L = val[e] @ invD[J]
                val[e] = L
                for f in range(diag[J] + 1, self.bptr[J + 1]):
                    M = self.bcol[f]
                    q = np.searchsorted(cols, M)
                    if q < cols.size and cols[q] == M:
                        val[b0 + q] -= L @ val[f]
            D = val[diag[p]]
            try:
                invD[p] = np.linalg.inv(D)
            except np.linalg.LinAlgError:
                self.nshift += 1
                invD[p] = np.linalg.inv(D + shift * np.eye(8))
        self.val, self.diag, self.invD = val, diag, invD

    def solve(self, b):
        nu, nN = self.nu, self.nu.shape[0]
        y = np.zeros((nN, 8))
        m = nu >= 0
        y[m] = np.asarray(b)[nu[m]]
        for p in range(nN):
            for e in range(self.bptr[p], self.diag[p]):
                y[p] -= self.val[e] @ y[self.bcol[e]]
        for p in range(nN - 1, -1, -1):
            t = y[p].copy()
            for e in range(self.diag[p] + 1, self.bptr[p + 1]):
                t -= self.val[e] @ y[self.bcol[e]]
            y[p] = self.invD[p] @ t
        x = np.zeros(self.n)
        x[nu[m]] = y[m]
        return x

    def scalar_twin(self, P, node_order=None):
        """ILU(0) (scalar oracle kernel) of P on the explicitly filled node pattern; unknowns ordered node by node in
        `node_order` (default: node positions sorted by their first unknown = the natural cell order)."""
        nu, nN = self.nu, self.nu.shape[0]
        if node_order is None:
            first = np.where(nu >= 0, nu, np.iinfo(np.int64).max).min(axis=1)
            node_order = np.argsort(first, kind="stable")
        unk = np.concatenate([nu[p][nu[p] >= 0] for p in node_order])
        n_loc = unk.size
        loc = np.full(self.n, -1, dtype=np.int64)
        loc[unk] = np.arange(n_loc)
        Pl = sp.csr_matrix(P)[unk][:, unk].tocsr()
        # pattern: all unknown pairs of coupled nodes
        rows, cols = [], []
        for p in range(nN):
            up = loc[nu[p][nu[p] >= 0]]
            for e in range(self.bptr[p], self.bptr[p + 1]):
                uq = loc[nu[self.bcol[e]][nu[self.bcol[e]] >= 0]]
                rr, cc = np.meshgrid(up, uq, indexing="ij")
                rows.append(rr.ravel())
                cols.append(cc.ravel())
        E = sp.csr_matrix((np.full(sum(r.size for r in rows), 1e-300), (np.concatenate(rows), np.concatenate(cols))), shape=(n_loc, n_loc))
        Pl = Pl.multiply(E != 0).tocsr()  # entries outside the node pattern are dropped
        ilu = ILU((Pl + E).tocsr(), fill=0)

        def solve(b):
            x = np.zeros(self.n)
            x[unk] = ilu.solve(np.asarray(b)[unk])
            return x

        return solve
