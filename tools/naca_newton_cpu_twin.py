"""Dev tool (CPU only): a twin of the GPU Newton-Krylov primal (das_solve_primal) built from the HOST-EMULATED kernel bodies
(tests/hostemu: the same templated bodies the HIP kernels wrap) - residual in milliseconds, exact J.v per colour through Dual<1> -
with scipy's sparse direct solve in place of GMRES.  Separates the nonlinear pseudo-time iteration from the linear solver.
Round 4: `--mass all` explodes beyond tau ~ 3 on the NACA0012 O-grid, `--mass Unu` (pseudo-time term on the
transport rows only) converges; logs: profiles/r05_cpu_twin_*.log.
   python tools/naca_newton_cpu_twin.py --n 100 32 --fc 1.6e-4 --steps 30 --grow 1.1 --taumax 20 --mass all|Unu|Unuphi|Unup
   python tools/naca_newton_cpu_twin.py --n 100 32 --fc 1.6e-4 --steps 120 --grow 1.5 --ls 1 --ser 1.0 --mass Unu"""
import sys, time, ctypes as C, os, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dafoam_amd.meshgen import naca0012_case, prolong_naca_state
from dafoam_amd import _capi
from dafoam_amd._capi import das_case_t, CaseStruct
from oracle.foam_mesh import Geometry
from oracle import jacobian as J
from common import norm_states
dptr = lambda a: a.ctypes.data_as(_capi.c_double_p)
L = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
L.emu_residual.argtypes = [C.POINTER(das_case_t), _capi.c_double_p, C.c_longlong, C.c_int, _capi.c_double_p, _capi.c_double_p, _capi.c_double_p]

class Twin:
    def __init__(self, case):
        self.case = case; self.cs = CaseStruct(case); self.g = Geometry(case.mesh)
        self.n = case.states.size; self.N = self.g.nC
        self.sc = J.state_scales(case, self.g, norm_states(case))
        t = time.time()
        self.con = J.connectivity(case, self.g).tocsr(); self.con.sort_indices()
        self.col, _ = J.greedy_coloring(self.con)
        self.nc = int(self.col.max()) + 1
        self.rows = np.repeat(np.arange(self.n), np.diff(self.con.indptr))
        print(f"twin: n {self.n} nnz {self.con.nnz} colours {self.nc} setup {time.time()-t:.1f}s", flush=True)
    def res(self, W, isPC=0):
        Rv, Rd = np.zeros(self.n), np.zeros(self.n)
        assert L.emu_residual(self.cs.byref(), dptr(W), self.n, isPC, None, dptr(Rv), dptr(Rd)) == 0
        return Rv
    def jac(self, W, isPC=0):
        """J S as CSR (rows residuals, cols states)."""
        RD = np.zeros((self.nc, self.n)); Rv = np.zeros(self.n)
        for c in range(self.nc):
            d = np.where(self.col == c, self.sc, 0.0)
            assert L.emu_residual(self.cs.byref(), dptr(W), self.n, isPC, dptr(d), dptr(Rv), dptr(RD[c])) == 0
        vals = RD[self.col[self.con.indices], self.rows]
        return sp.csr_matrix((vals, self.con.indices, self.con.indptr), shape=(self.n, self.n))



if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs=2, default=[100, 32]); ap.add_argument("--fc", type=float, default=1.6e-4)
    ap.add_argument("--steps", type=int, default=200); ap.add_argument("--tau0", type=float, default=1.0); ap.add_argument("--grow", type=float, default=1.2)
    ap.add_argument("--taumax", type=float, default=1e12); ap.add_argument("--ls", type=int, default=0, help="1: backtracking on |R| < 1.5 |R|")
    ap.add_argument("--jlag", type=int, default=1); ap.add_argument("--init", default=None); ap.add_argument("--save", default=None)
    ap.add_argument("--pcjac", type=int, default=0, help="1: first-order (isPC) Jacobian in the Newton matrix")
    ap.add_argument("--ser", type=float, default=0.0)
    ap.add_argument("--mass", default="all", help="all | Unu (pseudo-time term on U and nuTilda only) | Unuphi | Up")
    a = ap.parse_args()
    case = naca0012_case(a.n[0], a.n[1], 1, first_cell=a.fc, perturb=0.0)
    T = Twin(case); N = T.N
    W = np.load(a.init) if a.init else case.states.copy()
    R = T.res(W); r0 = rn = np.linalg.norm(R); tau = a.tau0
    blk = lambda R: " ".join(f"{np.linalg.norm(R[s]):.2e}" for s in (slice(0,3*N), slice(3*N,4*N), slice(4*N,5*N), slice(5*N,None)))
    print("R0", rn, blk(R))
    t0 = time.time()
    for k in range(a.steps):
        if k % a.jlag == 0:
            A = T.jac(W, a.pcjac); d = A.diagonal()
        msk = np.ones(T.n)
        if a.mass == 'Unu': msk[3*N:4*N] = 0; msk[5*N:] = 0
        if a.mass == 'Unuphi': msk[3*N:4*N] = 0
        if a.mass == 'Unup': msk[5*N:] = 0
        M = (A + sp.diags(msk * d / tau)).tocsc()
        dw = spla.splu(M).solve(-R)
        om = 1.0
        while True:
            Wn = W + om * T.sc * dw
            Wn[4*N:5*N] = np.maximum(Wn[4*N:5*N], 1e-14)
            Rn = T.res(Wn); rnew = np.linalg.norm(Rn)
            if not a.ls or (np.isfinite(rnew) and rnew < 1.5 * rn) or om < 1e-3: break
            om *= 0.5
        if not np.isfinite(rnew): print("NaN"); break
        adu = np.abs(T.sc * dw)[:3*N]; dU = adu.max(); cU = int(adu.argmax())//3; cR = int(np.abs(Rn[:3*N]).argmax())//3
        loc = f'dUcell(i,j)=({cU%a.n[0]},{cU//a.n[0]}) Rcell=({cR%a.n[0]},{cR//a.n[0]})'
        W, R = Wn, Rn
        g = a.grow if om == 1.0 else om
        if a.ser > 0 and om == 1.0: g = max(a.grow, min(10.0, (rn / rnew) ** a.ser))
        tau = min(a.taumax, tau * g)
        rn = rnew
        print(f"{k+1:4d} |R| {rn:.3e} [{blk(R)}] om {om:.3f} tau {tau:.2e} max|dU| {dU:.2e} {loc} t {time.time()-t0:.0f}s", flush=True)
        if rn < 1e-9 * r0: break
    if a.save: np.save(a.save, W)
