"""Dev tool (CPU only): what limits the adjoint's iteration count on the NACA0012 O-grid?  Exact Jacobians (operator A and the
first-order PC matrix P) of the host-emulated kernel bodies at a CONVERGED section (a naca_primal_*.npz written by
tools/naca_primal_study.py), then GMRES with (a) an exact LU of P, (b) scalar ILU(0) / ILU(1) of P in four cell orders, (c) blends
P + beta (A|pattern(P) - P) with exact and incomplete factorisations, (d) diagonally boosted blends, (e) flow-aligned orders.
Round 4: profiles/r05_cpu_pc_floor_study.log.   python tools/naca_pc_floor_study.py <naca_primal_100x32.npz>"""
import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tools"))
from oracle import linear as OL
from dafoam_amd.meshgen import naca0012_case
from naca_newton_cpu_twin import Twin
d=np.load(sys.argv[1]); nx,ny=[int(v) for v in d["dims"]]; fc=float(d["first_cell"])
case=naca0012_case(nx,ny,1,first_cell=fc,perturb=0.0); W=d["states"]; case.states=W
T=Twin(case); N=T.N; n=T.n
print("|R|",np.linalg.norm(T.res(W)))
A=T.jac(W).T.tocsr(); P=T.jac(W,1).T.tocsr()
rhs=np.zeros(n); rhs[0:3*N:3]=1.0/N
def run(pc,label):
    x,info=OL.gmres(lambda v:A@v,rhs,pc,restart=1500,max_iters=1500,rel_tol=1e-6)
    h=info["hist"]; print(f"{label:60s} iterations {info['iters']} plateau {int(np.argmax(h<0.5*h[0]))}",flush=True)
luP=spla.splu(P.tocsc()); run(lambda v:luP.solve(v),"exact LU of the first-order PC Jacobian (full stencil)")
# cell-by-cell ordering, ring-major (mesh order) and rcm-like via scipy
perm=np.concatenate([np.array([3*c,3*c+1,3*c+2,3*N+c,4*N+c]) for c in range(N)]+[np.arange(5*N,n)])
# faces to owner cell: put each face unknown after its owner cell's unknowns
own=case.mesh.owner
cells_f=[[] for _ in range(N)]
for f in range(case.mesh.n_faces): cells_f[int(own[f])].append(5*N+f)
def cellperm(order): return np.concatenate([np.array([3*c,3*c+1,3*c+2,3*N+c,4*N+c]+cells_f[c]) for c in order])
def ilu_run(order,label,M=P,fill=0):
    pm=cellperm(order); Mp=sp.csr_matrix(M[pm][:,pm]); Mp.sort_indices(); ilu=OL.ILU(Mp,fill=fill)
    def pc(v):
        y=np.empty(n); y[pm]=ilu.solve(v[pm]); return y
    run(pc,label)
ring=np.arange(N)
ilu_run(ring,"scalar ILU(0) of P, ring-major cell order (mesh numbering)")
ray=np.array([i+nx*j for i in range(nx) for j in range(ny)])
ilu_run(ray,"scalar ILU(0) of P, ray-major (wall-normal fastest)")
from scipy.sparse.csgraph import reverse_cuthill_mckee
g=T.g; C=g.cellCells.tocsr(); rc=reverse_cuthill_mckee(sp.csr_matrix(C),symmetric_mode=True)
ilu_run(rc,"scalar ILU(0) of P, RCM of the cell graph")
ilu_run(ring,"scalar ILU(1) of P, ring-major",fill=1)
ilu_run(rc,"scalar ILU(1) of P, RCM",fill=1)
ilu_run(ring,"scalar ILU(0) of A (exact 2nd-order matrix), ring-major",M=A)
print("--- blends of the PC matrix towards the operator (on the PC pattern)")
mask=(P!=0).astype(float)
Ar=A.multiply(mask).tocsr()
for beta in (0.25,0.5,0.75,1.0):
    Pb=(P+beta*(Ar-P)).tocsr()
    lu=spla.splu(Pb.tocsc()); run(lambda v:lu.solve(v),f"exact LU of P + {beta} (A|pattern - P)")
    ilu_run(rc,f"scalar ILU(0), RCM, of P + {beta} (A|pattern - P)",M=Pb)
    ilu_run(rc,f"scalar ILU(1), RCM, of P + {beta} (A|pattern - P)",M=Pb,fill=1)
print("--- diagonally boosted blends")
for beta,sig in ((0.5,0.1),(0.5,0.3),(1.0,0.3),(1.0,1.0),(0.75,0.5),(0.35,0.0),(0.35,0.1)):
    Pb=(P+beta*(Ar-P)); Pb=(Pb+sig*sp.diags(Pb.diagonal())).tocsr()
    ilu_run(rc,f"scalar ILU(0), RCM, of [P + {beta} (A|pat - P)] + {sig} diag",M=Pb)
    ilu_run(rc,f"scalar ILU(1), RCM, of [P + {beta} (A|pat - P)] + {sig} diag",M=Pb,fill=1)
print("--- flow-aligned orderings for blended PC matrices")
xc=T.g.C[:,0]; yc=T.g.C[:,1]
ox=np.argsort(xc,kind="stable"); oxr=ox[::-1]
# streamline-like: order by potential phi ~ x (far) ; near airfoil rings
for beta in (0.0,0.5,1.0):
    Pb=(P+beta*(Ar-P)).tocsr()
    for nm,o in (("x ascending",ox),("x descending",oxr),("ring-major",ring),("ray-major",ray)):
        ilu_run(o,f"ILU(0) beta {beta} order {nm}",M=Pb)
        if beta>0: ilu_run(o,f"ILU(1) beta {beta} order {nm}",M=Pb,fill=1)
