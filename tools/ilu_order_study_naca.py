"""Dev tool (CPU only, oracle matrix): GMRES iterations behind the scalar twin of the node-block ILU(0) for different
ELIMINATION ORDERS of the nodes - NACA0012 O-grid (96 x 32 x nz, span): usage ilu_order_study_naca.py nz span.  Results of round 3: profiles/r04l_ilu_ordering_study_cpu.log, DESIGN.md 6b."""
import sys, time, numpy as np, scipy.sparse as sp
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import linear as OL, jacobian as J
from oracle.foam_mesh import Geometry
from dafoam_amd.meshgen import naca0012_case
from dafoam_amd.pyDASolvers import pyDASolvers
from common import options, norm_states
na,nn=96,32; nz=int(sys.argv[1]); span=float(sys.argv[2])
case=naca0012_case(na,nn,nz,span=span); g=Geometry(case.mesh); N=g.nC
sc=J.state_scales(case,g,norm_states(case)); con=J.connectivity(case,g); col,_=J.greedy_coloring(con)
A=J.jacobian_colored(case,g,case.states,con,col,sc,mode="cs",lower_bound=0).tocsr(); n=A.shape[0]
rhs=np.zeros(n); rhs[0:3*N:3]=g.V; rhs*=sc
s=pyDASolvers(b"DASimpleFoam -python", options(case), case=case); S=s.pcStructure()
B=OL.NodeBlockILU.__new__(OL.NodeBlockILU); B.n,B.nu,B.bptr,B.bcol=n,S["nodeUnk"],S["bptr"].astype(np.int64),S["bcol"].astype(np.int64)
nu=S["nodeUnk"]
# cell of a node: from its U unknown if it has one, else "late"
firstU=np.where((nu>=0)&(nu<3*N), nu, 10**9).min(axis=1)
cell=np.where(firstU<10**9, firstU//3, -1)
i=cell%na; j=(cell//na)%nn; k=cell//(na*nn)
late=cell<0
def order(keys):
    key=np.lexsort(keys[::-1])  # first key slowest
    return np.concatenate([key[~late[key]], key[late[key]]])
orders={"library default (rcm of the cell graph)": np.argsort(S["natural"]),
        "k fastest, then i, then j": order((j,i,k)),
        "k fastest, then j, then i": order((i,j,k)),
        "j fastest, then i, then k": order((k,i,j)),
        "j fastest, then k, then i": order((i,k,j))}
for nm,o in orders.items():
    twin=B.scalar_twin(A,node_order=o)
    x,info=OL.gmres(lambda v:A@v,rhs,twin,restart=1000,max_iters=600,rel_tol=1e-6)
    print(f"NACA 96x32x{nz} span {span}: ILU order {nm:40s} iterations {info['iters']} rel {info['res']/info['res0']:.1e}", flush=True)
