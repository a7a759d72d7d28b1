"""Dev tool (CPU only, oracle matrices): variants of the two-level preconditioner behind the scalar twin of the GPU's node-block
ILU(0) - Schur-complement coarse operators, smoothed prolongators, the multiplicative form, coarse operators of the
PRECONDITIONED operator (E = Z^T A M^-1 Z).  usage: coarse_ideas_study.py channel | naca.  Results of round 3: DESIGN.md 6b."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import linear as OL, jacobian as J
from oracle.foam_mesh import Geometry
from dafoam_amd.meshgen import bench_channel_case, naca0012_case
from dafoam_amd.pyDASolvers import pyDASolvers
from common import options, norm_states
which=sys.argv[1]
if which=="channel":
    nx,ny,nz=[int(v) for v in sys.argv[2:5]] if len(sys.argv) >= 5 else (40,12,10); case=bench_channel_case(nx,ny,nz)
else:
    nx,ny,nz=96,32,1; case=naca0012_case(nx,ny,nz,span=0.1)
g=Geometry(case.mesh); N=g.nC
sc=J.state_scales(case,g,norm_states(case)); con=J.connectivity(case,g); col,_=J.greedy_coloring(con)
A=J.jacobian_colored(case,g,case.states,con,col,sc,mode="cs",lower_bound=0).tocsr(); n=A.shape[0]
rhs=np.zeros(n); rhs[0:3*N:3]=g.V; rhs*=sc
s=pyDASolvers(b"DASimpleFoam -python", options(case), case=case); S=s.pcStructure()
B=OL.NodeBlockILU.__new__(OL.NodeBlockILU); B.n,B.nu,B.bptr,B.bcol=n,S["nodeUnk"],S["bptr"].astype(np.int64),S["bcol"].astype(np.int64)
ilu=B.scalar_twin(A,node_order=np.argsort(S["natural"]))
sl={"U":slice(0,3*N),"p":slice(3*N,4*N),"nuT":slice(4*N,5*N),"phi":slice(5*N,n)}
App=A[sl["p"],sl["p"]].tocsr()
kk,jj,ii=np.meshgrid(np.arange(nz),np.arange(ny),np.arange(nx),indexing="ij")
def blocks(bi,bj,bk):
    nbi,nbj=(nx+bi-1)//bi,(ny+bj-1)//bj
    return ((kk//bk)*(nbi*nbj)+(jj//bj)*nbi+(ii//bi)).ravel()
agg=blocks(4,4,5) if which=="channel" else blocks(3,8,1)  # (channel: 4 cells along x per aggregate at every size)
nagg=agg.max()+1
Zp=sp.csr_matrix((np.ones(N),(np.arange(N),agg)),shape=(N,nagg))
def run(label, pc):
    x,info=OL.gmres(lambda v:A@v,rhs,pc,restart=1000,max_iters=600,rel_tol=1e-6)
    print(f"{which}: {label:70s} iterations {info['iters']:4d} rel {info['res']/info['res0']:.1e}", flush=True)
run("block ILU(0) twin only", ilu)
def additive(Z,E):
    Einv=np.linalg.inv(E)
    def pc(v):
        y=ilu(v).copy(); y[sl["p"]]+=Z@(Einv@(Z.T@v[sl["p"]])); return y
    return pc
run(f"+ additive p coarse space, E = Z^T A_pp Z ({nagg} aggregates)", additive(Zp,(Zp.T@App@Zp).toarray()))
# idea: Schur complement coarse operator, eliminating U and phi with their diagonals
def schur(fields):
    Sm=App.copy()
    for f in fields:
        Aff=A[sl[f],sl[f]]; d=Aff.diagonal(); d[d==0]=1.0
        Sm=Sm-A[sl["p"],sl[f]]@sp.diags(1.0/d)@A[sl[f],sl["p"]]
    return Sm.tocsr()
for fields in (("U",),("phi",),("U","phi")):
    Sm=schur(fields); run(f"+ additive p coarse space, E = Z^T S Z, S eliminates {fields} by their diagonals", additive(Zp,(Zp.T@Sm@Zp).toarray()))
# idea: smoothed aggregation (one damped-Jacobi step on the prolongator)
Dp=App.diagonal(); Zs=(Zp-0.67*sp.diags(1.0/Dp)@App@Zp).tocsr()
run("+ additive, smoothed prolongator (1 Jacobi step, omega 0.67)", additive(Zs,(Zs.T@App@Zs).toarray()))
# idea: multiplicative (coarse first, then ILU on the updated residual; one extra operator product)
E0inv=np.linalg.inv((Zp.T@App@Zp).toarray())
def mult(v):
    c=np.zeros(n); c[sl["p"]]=Zp@(E0inv@(Zp.T@v[sl["p"]])); return c+ilu(v-A@c)
run("multiplicative: coarse, then ILU of the updated residual", mult)
# idea: coarse space through the FULL operator: E = Z^T A M^-1 Z (deflation of the preconditioned operator), Z on p only
Zfull=sp.lil_matrix((n,nagg)); Zfull[3*N:4*N,:]=Zp; Zfull=Zfull.tocsc()
MZ=np.column_stack([ilu(Zfull[:,k].toarray().ravel()) for k in range(nagg)])
AMZ=A@MZ; E=Zfull.T@AMZ; Einv=np.linalg.inv(np.asarray(E))
def defl(v):
    # right-preconditioned operator B = A M^-1; correct with the coarse solve of B on span(Z): x = M^-1 (v - Z..)
    y=ilu(v); c=Einv@(Zfull.T@v); return y+MZ@c-ilu(AMZ@c)*0  # additive in the B-sense: M^-1 v + M^-1 Z E^-1 Z^T v
def defl2(v):
    c=Einv@(Zfull.T@v); return ilu(v)+MZ@c
run("additive with E = Z^T (A M^-1) Z, correction M^-1 Z E^-1 Z^T", defl2)
def adef1(v):
    c=Einv@(Zfull.T@v); w=v-AMZ@c; return ilu(w)+MZ@c
run("A-DEF1 on the preconditioned operator: M^-1 (I - B Z E^-1 Z^T) + M^-1 Z E^-1 Z^T", adef1)
